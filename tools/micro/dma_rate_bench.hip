// Developer micro-benchmark (GPU box): L2-resident LDS-DMA rate per CU for the access shapes gemm.hip uses.
// One wave-instruction = 64 lanes x 16 B = 8 rows x 128 B.  Variants: lane order within a row straight or XOR-swizzled
// (gemm.hip applies its LDS swizzle on the SOURCE address), rows at a pitch or packed, slab size, ring depth, waves/block.
// build+run: hipcc --offload-arch=gfx950 -O3 tools/micro/dma_rate_bench.hip -o /tmp/drb && /tmp/drb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

// block streams a panel of ROWS rows x (slabs*128 B), `reps` times; NW waves; ring of DEPTH slabs
template <int ROWS, int NW, int DEPTH>
__global__ __launch_bounds__(64 * NW) void dma_kernel(const char* w, size_t pitch, int slabs, int reps, int mode, unsigned* sink, int npanels) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PIECES = ROWS / 8;           // 1 KiB pieces per slab
    constexpr int PER_WAVE = PIECES / NW;
    constexpr int SLAB = ROWS * 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row_in = lane >> 3;
    const int chunk = mode == 1 ? ((lane & 7) ^ row_in) : (lane & 7);
    const char* base = w + (size_t)(blockIdx.x % npanels) * ROWS * pitch; // npanels small => every block re-reads L2-resident data
    unsigned acc = 0;
    const int total = slabs * reps;
    auto issue = [&](int t) {
        const int s = t % slabs;
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int piece = j * NW + wave;
            const char* g = mode == 2 ? base + ((size_t)s * PIECES + piece) * 1024 + lane * 16
                                      : base + (size_t)(piece * 8 + row_in) * pitch + (size_t)s * 128 + chunk * 16;
            __builtin_amdgcn_global_load_lds((glb_ptr)g, (lds_ptr)(smem + (t % DEPTH) * SLAB + piece * 1024), 16, 0, 0);
        }
    };
    for (int t = 0; t < DEPTH - 1 && t < total; ++t) issue(t);
    for (int t = 0; t < total; ++t) {
        if (t + DEPTH - 1 < total) {
            issue(t + DEPTH - 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE * (DEPTH - 1)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        acc ^= *reinterpret_cast<unsigned*>(smem + (t % DEPTH) * SLAB + threadIdx.x * 4);
    }
    if (acc == 0x12345678u) *sink = acc;
}

template <int ROWS, int NW, int DEPTH>
void run(const char* w, size_t pitch, int slabs, unsigned* sink, int blocks, int npanels) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 64;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<ROWS, NW, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * ROWS * 128));
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((dma_kernel<ROWS, NW, DEPTH>), dim3(blocks), dim3(64 * NW), DEPTH * ROWS * 128, 0, w, pitch, slabs, reps, mode, sink, npanels);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double bytes = (double)blocks * ROWS * 128.0 * slabs * reps;
        printf("panels %4d rows %3d waves %d depth %d blocks %4d %-9s : %8.1f us  %6.1f GB/s per block  %6.2f TB/s chip\n", npanels, ROWS, NW, DEPTH, blocks,
               mode == 0 ? "straight" : mode == 1 ? "swizzled" : "packed", best * 1e3, bytes / blocks / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e12);
    }
}

int main() {
    const int slabs = 16;                       // 2 KiB of K per row
    const size_t pitch = 2880 * 2;              // a conv 320 weight row
    const size_t bytes = (size_t)512 * 256 * pitch;
    char* w; unsigned* sink;
    CK(hipMalloc(&w, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(w, 1, bytes));
    for (int npanels : {4, 100000})
        for (int blocks : {192, 256, 512}) {
            run<128, 4, 3>(w, pitch, slabs, sink, blocks, npanels);
            run<256, 8, 2>(w, pitch, slabs, sink, blocks, npanels);
            run<256, 8, 4>(w, pitch, slabs, sink, blocks, npanels);
        }
    return 0;
}

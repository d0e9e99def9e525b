// Developer micro-benchmark (GPU box): how fast can 64-row weight panels be streamed from HBM when every 1 KiB DMA piece is
// (a) 8 rows x 128 B at the row pitch of a [N][K] matrix (what gemm.hip's B operand does today), or
// (b) one contiguous 1 KiB piece (a slab-packed weight layout)?
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/weight_stream_bench.hip -o gpurun_out/wsb && gpurun_out/wsb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;

// one block = 64 rows x (slabs * 64) halves of K; 4 waves; each wave moves 2 pieces (16 rows) per slab
template <int DEPTH, bool DMA>
__global__ __launch_bounds__(256) void stream_kernel(const char* w, size_t pitch, int slabs_per_block, int kblocks, int packed, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt = blockIdx.x / kblocks, kb = blockIdx.x % kblocks;
    const int row0 = nt * 64 + wave * 16;
    unsigned acc = 0;
    uint4 regs[DEPTH][2];
    auto addr = [&](int slab, int j) -> const char* {
        const int s = kb * slabs_per_block + slab;
        if (packed) {
            // [k-slab][n/8][8 rows][128 B]
            const size_t nrows8 = (size_t)gridDim.x / kblocks * 8;
            return w + ((size_t)s * nrows8 + (size_t)(row0 / 8 + j)) * 1024 + lane * 16;
        }
        return w + (size_t)(row0 + j * 8 + (lane >> 3)) * pitch + (size_t)s * 128 + (lane & 7) * 16;
    };
    if constexpr (DMA) {
        // ring of DEPTH slabs (8 KiB each); wait with vmcnt so DEPTH-1 slabs stay in flight
        for (int s = 0; s < DEPTH - 1 && s < slabs_per_block; ++s)
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_global_load_lds((glb_ptr)addr(s, j), (lds_ptr)(smem + (s % DEPTH) * 8192 + (wave * 2 + j) * 1024), 16, 0, 0);
        for (int s = 0; s < slabs_per_block; ++s) {
            if (s + DEPTH - 1 < slabs_per_block) {
                for (int j = 0; j < 2; ++j)
                    __builtin_amdgcn_global_load_lds((glb_ptr)addr(s + DEPTH - 1, j), (lds_ptr)(smem + ((s + DEPTH - 1) % DEPTH) * 8192 + (wave * 2 + j) * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (DEPTH - 1)) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            acc ^= *reinterpret_cast<unsigned*>(smem + (s % DEPTH) * 8192 + wave * 2048 + lane * 4);
        }
    } else {
        for (int s = 0; s < DEPTH - 1 && s < slabs_per_block; ++s)
            for (int j = 0; j < 2; ++j) regs[s % DEPTH][j] = *reinterpret_cast<const uint4*>(addr(s, j));
#pragma unroll 1
        for (int s0 = 0; s0 < slabs_per_block; s0 += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int s = s0 + u;
                if (s + DEPTH - 1 < slabs_per_block)
                    for (int j = 0; j < 2; ++j) regs[(u + DEPTH - 1) % DEPTH][j] = *reinterpret_cast<const uint4*>(addr(s + DEPTH - 1, j));
                if (s < slabs_per_block) acc ^= regs[u][0].x ^ regs[u][1].y;
            }
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const int K = 11520, rows = 1280 * 20;          // 20 conv 1280->1280 weight matrices: 590 MB, > Infinity Cache
    const size_t pitch = (size_t)K * 2, bytes = (size_t)rows * pitch;
    char* w; unsigned* sink;
    CK(hipMalloc(&w, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(w, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int KT = K / 64;                          // 180 slabs
    printf("%-8s %-7s %-6s %8s %8s %9s\n", "path", "layout", "kblks", "blocks", "us", "TB/s");
    for (int dma = 0; dma < 2; ++dma)
        for (int packed = 0; packed < 2; ++packed)
            for (int kblocks : {1, 4, 12, 36}) {
                const int spb = KT / kblocks, blocks = rows / 64 * kblocks;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(e0, 0));
                    if (dma) hipLaunchKernelGGL((stream_kernel<4, true>), dim3(blocks), dim3(256), 4 * 8192, 0, w, pitch, spb, kblocks, packed, sink);
                    else hipLaunchKernelGGL((stream_kernel<4, false>), dim3(blocks), dim3(256), 0, 0, w, pitch, spb, kblocks, packed, sink);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("%-8s %-7s %-6d %8d %8.1f %9.2f\n", dma ? "lds-dma" : "regs", packed ? "packed" : "pitched", kblocks, blocks, best * 1e3, bytes / (best * 1e-3) / 1e12);
            }
    return 0;
}

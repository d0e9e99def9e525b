// gemm.hip -- fp16 MFMA GEMM with an implicit-im2col A operand (gfx950 / MI355X).
//
//   out[M][N] = act(alpha * A[M][K] . W[N][K]^T + bias + row_bias) + residual
//
// This is the arithmetic behind every Linear / 1x1 conv / 3x3 conv of the UNet, VAE decoder and CLIP
// graphs that the reference executes as opaque QNN blobs (qnn_context.cpp:711-713; op inventory in
// analyze_results.py:20-93).  Design (MI355X-first, not a CUDA tiling):
//   * one workgroup = 256 threads = 4 wave64; each wave owns a (BM/WM)x(BN/WN) output tile built from
//     v_mfma_f32_16x16x32_f16 (fp32 accumulate in the unified VGPR/AGPR file);
//   * K is walked in BK=64 slabs; both operands are K-contiguous, so a slab row is one 128-byte line.
//     For the 3x3 conv the slab of row m=(img,oy,ox) is the 128-byte channel run of ONE input pixel
//     (tap (r,s), channels c..c+63) -- im2col never exists in HBM; nearest-2x upsampling, stride 2, a
//     two-tensor channel concat and the 1x1 case (ksize 1) are folded into the same address computation;
//   * slabs are staged global -> registers -> LDS (double-buffered, one barrier per slab; the loads for
//     slab t+1 are issued before the MFMAs of slab t), LDS rows are XOR-swizzled in 16-byte chunks so
//     the ds_read_b128 fragment reads are bank-conflict free;
//   * W is the MFMA "A" operand and the activations the "B" operand, so each lane ends up holding four
//     CONSECUTIVE output columns of one row: the epilogue packs them to fp16, stages the tile through
//     LDS and stores full 16-byte row segments (coalesced), with bias / per-image bias (time embedding)
//     / activation / residual fused;
//   * workgroup ids are remapped so that the n-tiles of one m-tile share an XCD (its L2 holds the A rows);
//   * small-M layers (8x8 / 16x16 feature maps) are weight-bandwidth bound: split-K spreads the weight
//     stream over all 256 CUs, partial slabs are reduced (with the fused epilogue) by a second kernel.
#include "common.h"
#include "sdod_hip.h"
#include "host_util.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {

struct GemmP {
    const f16* a0;
    const f16* a1;
    const f16* w;
    const float* bias;
    const f16* row_bias;
    const f16* residual;
    f16* out;
    float* partial;
    int M, N, K;
    int lda, ldw, ldo, ldr;
    int mode;
    int h_in, w_in, h_out, w_out, c0, c1, stride, ups, ksize;
    int sa0, sa1; // pixel strides (elements) of the two A sources: c0/c1 for NHWC images, lda for plain rows
    int geglu;    // epilogue pairs 16-column blocks: out = a * gelu(gate)
    int k_tail;   // K columns >= k_tail come from the 1x1-gathered tail sources t0|t1 (0 = none)
    const f16* t0;
    const f16* t1;
    int tc0, tc1;
    const float* bias2;
    unsigned mg_hw, sh_hw, mg_w, sh_w, mg_tn, sh_tn, mg_cin, sh_cin; // magic multipliers: m/(h_out*w_out), rem/w_out, lid/tiles_n, k0/cin
    int ln;            // LayerNorm of the A rows folded into this GEMM (row statistics gathered from the LDS slabs)
    const float* ln_s;
    float ln_eps;
    int rows_per_img, ldrb;
    int act;
    float alpha;
    int bias_on_m;
    int splits, kt_per_split;
    int tiles_m, tiles_n;
    // int8 weight streaming (config 5, "mirrors the QNN quant path"): W holds the affine-uint8 codes q of the reference's
    // encoding real = (q + offset) * scale (qnn_context.cpp:1018-1033), one byte per element, [N][ldw bytes]; per output
    // column n: w_scale[n] = scale, w_off[n] = offset + 128 (as float; offset an INTEGER in [-1024, 0]).  out = scale * sum_k A (q + offset).
    int wq;
    const float* w_scale;
    const float* w_off;
    int fixup;               // split-K without a reduce launch: the last slice to arrive at a tile's counter reduces (conv_halo_kernel)
    unsigned* fix_counters;  // [tiles_m * tiles_n], zero when idle (in the caller's zeroed workspace, behind the partial slabs)
    // per-image weights (rows mode, gemm_glds_kernel): the rows of image i = m / rows_per_img multiply w + i * w_img_stride (halves) and
    // take their per-column vectors (bias, ln_s) from + i * vec_img_stride (floats); tiles never straddle images (BM | rows_per_img)
    int w_img_stride, vec_img_stride;
    unsigned mg_rpi, sh_rpi; // m / rows_per_img
    int softmax_g;           // 80: the epilogue replaces every run of 80 columns (one wave's columns) by its row softmax, exp2 domain
    int no_respf; // developer switch (SDOD_GEMM_RESPF=0): no early residual prefetch
    int lean; // plain row-major operands whose byte offsets fit 32 bits: the loaders take the short issue path
    // halo-patch convolution (conv_halo_kernel, tiles 37..): geometry of one workgroup's output tile and of the input patch
    // it keeps in LDS, all host-computed (halo_geometry)
    int h_tw, h_th;          // tile = h_tw consecutive pixels of h_th rows per part (h_tw == BM: a row segment; else whole rows)
    int h_pw, h_ppix;        // patch row pitch h_tw + 2, pixels of one part's patch (h_th + 2) * h_pw
    int h_npix, h_nr;        // patch pixels over all parts (a tile taller than the image spans `parts` whole images), DMA rounds
    int h_nimg;              // images in the batch (parts past the last one read zeros)
    int h_patch_halves;      // LDS halves of one patch buffer
    int h_colv_off;          // byte offset of the per-column epilogue vectors in LDS
    int h_main_splits;       // splits that walk the 3x3 taps (split h_main_splits, when k_tail != 0, walks the 1x1 tail)
    int h_chain;             // 1: ONE split; its workgroups run the 1x1 tail behind the tap walk themselves (no partial slabs)
    unsigned mg_tw, sh_tw, mg_th, sh_th, mg_pw, sh_pw, mg_pp, sh_pp;
    // XCD-aware tile order (tile_of): the n-tiles are cut into xg panels, the first xg_r of them one tile wider (xg_w + 1);
    // logical ids walk panel after panel, m-major inside a panel.  xg = 1 is the plain m-major order.
    // A-panel kernel (gemm_apanel_kernel): workgroup = (row panel, group of ap_tpg consecutive n-tiles); ap_groups groups per panel
    int ap_groups, ap_tpg, ap_panels, ap_nmajor;
    int xg_w, xg_big;        // narrow panel width; ids below xg_big belong to the wide panels
    int xg_s1, xg_s0;        // tiles per wide / narrow panel (tiles_m * width)
    int xg_nbig;             // n-tiles covered by the wide panels
    unsigned mg_s1, sh_s1, mg_s0, sh_s0, mg_w1, sh_w1, mg_w0, sh_w0;
};

constexpr int BK = 64;

// n / d for n < 2^31 through a host-computed magic multiplier (the kernels' prologues are on the critical path of
// every launch: a hardware-less integer division costs ~35 instructions)
SDOD_DEVICE int fast_div(int n, unsigned magic, unsigned shift) {
    return (int)(((unsigned)__umulhi((unsigned)n, magic) + (unsigned)n) >> shift);
}
inline void make_magic(unsigned d, unsigned* magic, unsigned* shift) {
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    *shift = s;
    *magic = (unsigned)((((1ull << 32) * ((1ull << s) - d)) / d) + 1);
}

// Logical tile id -> (tile_m, tile_n).  Workgroups b and b + 8 share an XCD (round-robin dispatch; speed only) and xcd_remap
// hands each XCD a contiguous run of logical ids, so the ORDER of the ids decides which operand bytes each of the eight L2s
// has to fetch: m-major ids give every XCD a band of m-tiles and ALL of W (fabric traffic A + 8 W), n-major ids all of A and
// an eighth of W (8 A + W), and xg panels of n-tiles (8 / xg bands of m-tiles inside each) anything in between:
// xg A + (8 / xg) W.  The host picks xg per shape (xcd_panels()).
SDOD_DEVICE void tile_of(const GemmP& p, int lid, int& tile_m, int& tile_n) {
    if (lid < p.xg_big) {
        const int j = fast_div(lid, p.mg_s1, p.sh_s1);
        const int o = lid - j * p.xg_s1;
        tile_m = fast_div(o, p.mg_w1, p.sh_w1);
        tile_n = j * (p.xg_w + 1) + (o - tile_m * (p.xg_w + 1));
    } else {
        const int l2 = lid - p.xg_big;
        const int j = fast_div(l2, p.mg_s0, p.sh_s0);
        const int o = l2 - j * p.xg_s0;
        tile_m = fast_div(o, p.mg_w0, p.sh_w0);
        tile_n = p.xg_nbig + j * p.xg_w + (o - tile_m * p.xg_w);
    }
}

SDOD_DEVICE int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ (row & 7)) << 3); }

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmP p) {
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_IT = BM / 32;
    constexpr int B_IT = (BN + 31) / 32;
    constexpr int STAGE = (BM + BN) * 64; // halves per stage
    constexpr int SC = BN + 8;            // epilogue tile row stride (halves)
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(TM >= 1 && TN >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f16* smem = reinterpret_cast<f16*>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile assignment: consecutive logical ids (same XCD) walk the n-tiles of one m-tile
    const int nwg = p.tiles_m * p.tiles_n;
    const int lid = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    tile_of(p, lid, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.z;

    const int KT = p.K / BK;
    int kt_begin = 0, kt_end = KT;
    if (p.splits > 1) {
        kt_begin = split * p.kt_per_split;
        kt_end = min(KT, kt_begin + p.kt_per_split);
    }

    // ---- per-thread staging geometry: thread owns chunk (tid&7) of rows (tid>>3)+32*i
    const int ld_chunk = tid & 7;
    const int ld_row = tid >> 3;

    // A rows: precompute the pixel decomposition for the conv gather
    int a_img[A_IT], a_oy[A_IT], a_ox[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + ld_row + 32 * i;
        a_ok[i] = m < p.M;
        if (p.mode == SDOD_A_CONV3X3) {
            const int hw = p.h_out * p.w_out;
            const int mm = a_ok[i] ? m : 0;
            const int img = mm / hw;
            const int rem = mm - img * hw;
            const int oy = rem / p.w_out;
            a_img[i] = img;
            a_oy[i] = oy * p.stride;
            a_ox[i] = (rem - oy * p.w_out) * p.stride;
        } else {
            a_img[i] = m;
            a_oy[i] = 0;
            a_ox[i] = 0;
        }
    }
    const int cin = p.c0 + p.c1;
    const int hup = p.h_in << p.ups, wup = p.w_in << p.ups;

    f16x8 ra[A_IT], rb[B_IT];

    auto load_tile = [&](int kt) {
        const int k0 = kt * BK;
        if (p.mode == SDOD_A_CONV3X3) {
            const int tap = k0 / cin;
            const int cc = k0 - tap * cin;
            const int r = tap / p.ksize, s = tap - r * p.ksize;
            const int pad = p.ksize >> 1;
            const f16* src = p.a0;
            int csrc = p.c0, ccs = cc;
            if (cc >= p.c0) {
                src = p.a1;
                csrc = p.c1;
                ccs = cc - p.c0;
            }
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                int yy = a_oy[i] + r - pad, xx = a_ox[i] + s - pad;
                const bool ok = a_ok[i] && yy >= 0 && yy < hup && xx >= 0 && xx < wup;
                yy >>= p.ups;
                xx >>= p.ups;
                const size_t off = ((size_t)(a_img[i] * p.h_in + yy) * p.w_in + xx) * csrc + ccs + ld_chunk * 8;
                ra[i] = ok ? ldg8(src + off) : zero8();
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const size_t off = (size_t)a_img[i] * p.lda + k0 + ld_chunk * 8;
                ra[i] = a_ok[i] ? ldg8(p.a0 + off) : zero8();
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int rloc = ld_row + 32 * i;
            const int n = n0 + rloc;
            const bool ok = (rloc < BN) && (n < p.N);
            rb[i] = ok ? ldg8(p.w + (size_t)n * p.ldw + k0 + ld_chunk * 8) : zero8();
        }
    };

    auto store_tile = [&](int stage) {
        f16* sA = smem + stage * STAGE;
        f16* sB = sA + BM * 64;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int row = ld_row + 32 * i;
            *reinterpret_cast<f16x8*>(sA + lds_off(row, ld_chunk)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int row = ld_row + 32 * i;
            if (row < BN) *reinterpret_cast<f16x8*>(sB + lds_off(row, ld_chunk)) = rb[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frag_row = lane & 15;
    const int frag_chunk = lane >> 4;

    if (kt_begin < kt_end) {
        load_tile(kt_begin);
        store_tile(0);
    }
    __syncthreads();

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const bool has_next = kt + 1 < kt_end;
        if (has_next) load_tile(kt + 1);

        const f16* sA = smem + cur * STAGE;
        const f16* sB = sA + BM * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 xa[TM], wb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                xa[i] = *reinterpret_cast<const f16x8*>(sA + lds_off(wm * WTM + i * 16 + frag_row, ks * 4 + frag_chunk));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                wb[j] = *reinterpret_cast<const f16x8*>(sB + lds_off(wn * WTN + j * 16 + frag_row, ks * 4 + frag_chunk));
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = mfma16(wb[j], xa[i], acc[i][j]);
        }
        if (has_next) store_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  lane holds out[m = .. + (lane&15)][n = .. + 4*(lane>>4) + r], r = 0..3
    const int e_m = lane & 15;
    const int e_n = (lane >> 4) * 4;

    if (p.splits > 1) {
        float* slab = p.partial + (size_t)split * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + e_m;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + e_n;
                if (n + 3 < p.N) {
                    *reinterpret_cast<f32x4*>(slab + (size_t)m * p.N + n) = acc[i][j];
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N) slab[(size_t)m * p.N + n + r] = acc[i][j][r];
                }
            }
        }
        return;
    }

    f16* sC = smem; // aliases the staging buffers; all waves are past the final barrier of the K loop
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm * WTM + i * 16 + e_m;
        const int m = m0 + ml;
        const f16* rbias = nullptr;
        if (p.row_bias != nullptr && m < p.M) rbias = p.row_bias + (size_t)(m / p.rows_per_img) * p.ldrb;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * WTN + j * 16 + e_n;
            const int n = n0 + nl;
            f16x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[i][j][r] * p.alpha;
                if (p.bias != nullptr) {
                    if (p.bias_on_m) {
                        if (m < p.M) v += p.bias[m];
                    } else if (n + r < p.N) {
                        v += p.bias[n + r];
                    }
                }
                if (rbias != nullptr && n + r < p.N) v += (float)rbias[n + r];
                v = apply_act(v, p.act);
                h[r] = (f16)v;
            }
            *reinterpret_cast<f16x4*>(sC + ml * SC + nl) = h;
        }
    }
    __syncthreads();

    constexpr int CPR = BN / 8; // 16-byte chunks per tile row
    const bool vec_ok = (p.N % 8 == 0) && (p.ldo % 8 == 0) && (p.residual == nullptr || p.ldr % 8 == 0);
    for (int idx = tid; idx < BM * CPR; idx += 256) {
        const int row = idx / CPR;
        const int ch = idx - row * CPR;
        const int m = m0 + row, n = n0 + ch * 8;
        if (m >= p.M || n >= p.N) continue;
        f16x8 v = *reinterpret_cast<const f16x8*>(sC + row * SC + ch * 8);
        if (vec_ok) {
            if (p.residual != nullptr) {
                const f16x8 rr = ldg8(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] + (float)rr[e]);
            }
            stg8(p.out + (size_t)m * p.ldo + n, v);
        } else {
            for (int e = 0; e < 8; ++e) {
                if (n + e < p.N) {
                    float f = (float)v[e];
                    if (p.residual != nullptr) f += (float)p.residual[(size_t)m * p.ldr + n + e];
                    p.out[(size_t)m * p.ldo + n + e] = (f16)f;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// v2 main loop: LDS-DMA staging (global_load_lds_dwordx4: 1 KiB per wave-instruction straight into LDS, no VGPR
// round trip, no ds_write) into a ring of STAGES slabs with the prefetch running STAGES-1 slabs ahead.  One raw
// s_barrier per slab; loads stay in flight across it behind a counted s_waitcnt vmcnt(N) (the compiler-inserted
// vmcnt(0) of __syncthreads() would drain them).  The LDS image is lane-linear per wave-instruction (8 rows x 8
// chunks), so the XOR swizzle is applied to the per-lane SOURCE address and again on the fragment reads; rows that
// do not exist (M/N tails, conv halo) read a zero line instead of being predicated, so every lane issues exactly the
// same number of DMA ops and the vmcnt arithmetic stays exact.
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

template <int N>
SDOD_DEVICE void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wait until at most `y` slabs (y <= Y, wave-uniform) of LOADS DMA instructions each are still in flight
template <int LOADS, int Y>
SDOD_DEVICE void wait_younger(int y) {
    if (y >= Y) wait_vmcnt<LOADS * Y>();
    else if constexpr (Y > 0) wait_younger<LOADS, Y - 1>(y);
}

// Keep a wave-uniform value in an SGPR for good.  Kernel arguments are otherwise re-read from the kernarg segment (s_load)
// wherever the register allocator finds that cheaper -- also inside main loops, where a pending scalar load forces every
// following LDS wait to lgkmcnt(0) (SMEM returns out of order), i.e. kills the counted waits of a software pipeline.
template <class T>
SDOD_DEVICE T sgpr_pin(T v) {
    if constexpr (sizeof(T) == 8) {
        unsigned long long u = (unsigned long long)v;
        asm volatile("" : "+s"(u));
        return (T)u;
    } else {
        asm volatile("" : "+s"(v));
        return v;
    }
}

// conv_halo_kernel's loaders: patch DMA rounds (rpt in front of every slab at taps >= ahead, nrmax in all) that are issued
// after the slab of tap ti and before the slab ahead slabs later, i.e. in front of taps ti+1 .. ti+ahead-1 of the same chunk
constexpr int halo_rounds_between(int ti, int ahead, int nrmax, int rpt) {
    int n = 0;
    for (int u = ti + 1; u <= ti + ahead - 1 && u <= 8; ++u) {
        if (u < ahead) continue;
        int r = nrmax - rpt * (u - ahead);
        r = r < 0 ? 0 : r > rpt ? rpt : r;
        n += r;
    }
    return n;
}

// 16-byte-per-lane LDS-DMA in the SADDR form: address = wave-uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset.
// (Inline asm also keeps the DMA out of the compiler's wait-count model, which treats global_load_lds as a FLAT access that may
// touch LDS and, while one is pending, turns every LDS wait of the wave into lgkmcnt(0); the waits for these DMAs are the
// hand-counted s_waitcnt vmcnt(N) of the loaders.)
// Three instructions per DMA (m0, the hazard nop, the load) and no vector arithmetic; the compiler's own selection of the
// builtin spends two 64-bit vector adds per DMA on the same address.  base and lds_dst must be wave-uniform.
SDOD_DEVICE void lds_dma16_saddr(const void* base, unsigned off, f16* lds_dst) {
    const unsigned a = (unsigned)(uintptr_t)(lds_void_ptr)lds_dst;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(a) : "memory", "m0");
}
// (Round 3 measured the non-temporal cache policy for the weight slabs -- `global_load_lds_dwordx4 ... nt` for matrices of
// >= 4 / 16 MiB, MI355X_MICROARCH.md nt-weights -- at +1.3 % per evaluation (profiles/r03_nt_weights.txt); the switch is gone
// again because even switched off its wave-uniform branch sat between the barrier and every weight DMA of the loaders.)
// The same with the LDS destination given as a byte address (running pointers of the lean loader loop).
SDOD_DEVICE void lds_dma16_saddr_raw(const void* base, unsigned off, unsigned lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds_byte_addr) : "memory", "m0");
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <int N, int I = 0, class F>
SDOD_DEVICE void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate): n >= 32 waits for everything
SDOD_DEVICE void wait_vmcnt_dyn(int n) {
    switch (n) {
#define SDOD_W(k) case k: wait_vmcnt<k>(); break;
        SDOD_W(0) SDOD_W(1) SDOD_W(2) SDOD_W(3) SDOD_W(4) SDOD_W(5) SDOD_W(6) SDOD_W(7) SDOD_W(8) SDOD_W(9) SDOD_W(10) SDOD_W(11)
        SDOD_W(12) SDOD_W(13) SDOD_W(14) SDOD_W(15) SDOD_W(16) SDOD_W(17) SDOD_W(18) SDOD_W(19) SDOD_W(20) SDOD_W(21) SDOD_W(22)
        SDOD_W(23) SDOD_W(24) SDOD_W(25) SDOD_W(26) SDOD_W(27) SDOD_W(28) SDOD_W(29) SDOD_W(30) SDOD_W(31)
#undef SDOD_W
    default: wait_vmcnt<0>(); break;
    }
}

// developer builds only (make lib/libsdod_stamp.so): wall-clock stamps (s_memrealtime, 100 MHz) of the phases of every
// workgroup -- 0 entry, 1 prologue issued, 2 main loop drained, 3 epilogue tile staged, 4 stores retired -- read back by
// sdod_gemm_stamps() (tools/gemm_phases.py).  In the product build the macro is empty.
#ifdef SDOD_GEMM_STAMP
__device__ unsigned long long g_stamp[8 * 8192];
#define STAMP(i)                                                                                            \
    do {                                                                                                    \
        const unsigned wg_ = blockIdx.x + gridDim.x * blockIdx.z;                                           \
        if (threadIdx.x == 0 && wg_ < 8192) g_stamp[wg_ * 8 + (i)] = __builtin_amdgcn_s_memrealtime();     \
    } while (0)
// (the A-panel kernel stamps from the first lane of one consumer and one loader wave)
#define STAMPW(i, w)                                                                                        \
    do {                                                                                                    \
        const unsigned wg_ = blockIdx.x;                                                                    \
        if (threadIdx.x == 64u * (w) && wg_ < 8192) g_stamp[wg_ * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define STAMP(i)
#define STAMPW(i, w)
#endif

// 8 affine-uint8 weight codes (two dwords) -> f16x8 of (q - 128), exactly: byte b next to 0x64 is the fp16 number 1024 + b
// (v_perm_b32 builds two of them per instruction), and one packed subtraction of z = 1024 - offset turns it into the integer
// q + offset of the affine encoding, exactly (|q + offset| <= 2048): the zero point never reaches the accumulators, so no
// row sums of A are needed.  4 + 4 VALU per 8 weights.
SDOD_DEVICE f16x8 u8x8_to_f16(u32x2 d, f16 z) {
    const uint32_t c64 = 0x64646464u;
    typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 nz = {(_Float16)-z, (_Float16)-z};
    uint32_t w[4];
    w[0] = __builtin_amdgcn_perm(c64, d[0], 0x04010400u);
    w[1] = __builtin_amdgcn_perm(c64, d[0], 0x04030402u);
    w[2] = __builtin_amdgcn_perm(c64, d[1], 0x04010400u);
    w[3] = __builtin_amdgcn_perm(c64, d[1], 0x04030402u);
    u32x4v o;
#pragma unroll
    for (int i = 0; i < 4; ++i) // two-wide vector adds: v_pk_add_f16 (the 8-wide subtraction of a splat is scalarised when z is a register)
        o[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(h2, w[i]) + nz);
    return __builtin_bit_cast(f16x8, o);
}
// fragment of weight row `row` (tile-local), K half `ks`, lane chunk `fc` (0..3), from the uint8 slab image (8-row groups at a
// 1 KiB pitch, 64-byte rows, 16-byte chunks swizzled with (row >> 2) & 3)
SDOD_DEVICE f16x8 wq_frag(const f16* sB, int row, int ks, int fc, f16 z) {
    const unsigned char* base = reinterpret_cast<const unsigned char*>(sB) + (row >> 3) * 1024 + (row & 7) * 64;
    const int chunk = (ks * 2 + (fc >> 1)) ^ ((row >> 2) & 3);
    return u8x8_to_f16(*reinterpret_cast<const u32x2*>(base + chunk * 16 + (fc & 1) * 8), z);
}

// SPEC = wave specialisation: the workgroup is WM*WN CONSUMER waves (one per SIMD: fragment reads + MFMA, each owning a
// (BM/WM)x(BN/WN) output tile) plus as many LOADER waves (their SIMD partners: nothing but the LDS-DMA issue of the slab
// STAGES-1 ahead, and the LayerNorm-fold row statistics).  In the unspecialised form every wave does reads -> DMA issue ->
// MFMA in series and the one barrier per slab keeps all waves in lockstep, so per slab the LDS read burst (~0.18 us at
// 128x128), the DMA issue (~100-180 cycles per instruction, ~0.2 us) and the MFMA cluster (~0.24 us) ADD UP (0.70 us
// measured, tools/gemm_phases.py); with the roles split they overlap between the same barriers.
// WQ = the weight operand is affine uint8 (GemmP::wq): a compile-time variant, so that the fp16 kernels carry none of its code
// or registers (several sit exactly at the 128-register step that lets two workgroups share a CU).
// KSUB = slabs per barrier: the ring holds STAGES groups of KSUB slabs and the workgroup synchronises once per GROUP (the
// counted wait + barrier + loop bookkeeping of a 64-deep slab cost ~0.15 us, a third of a 64x64 tile's slab time).
template <int BM, int BN, int WM, int WN, int STAGES, bool SPEC = false, bool WQ = false, int KSUB = 1>
__global__ __launch_bounds__(64 * WM * WN * (SPEC ? 2 : 1)) void gemm_glds_kernel(const GemmP p, const f16* __restrict__ zeros) {
    constexpr int NC = WM * WN;                   // waves that own output tiles
    constexpr int NW = SPEC ? 2 * NC : NC;        // waves per workgroup: 4 (one per SIMD) or 8 (two per SIMD)
    constexpr int NL = SPEC ? NC : NW;            // waves that issue the DMA
    constexpr int NT = 64 * NW;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_LD = BM / (8 * NL), B_LD = BN / (8 * NL); // DMA instructions per loading wave per slab (8 rows each)
#if defined(SDOD_GEMM_ABLATE) && (SDOD_GEMM_ABLATE & 8)
    constexpr int LOADS = B_LD;                   // developer build: the A operand is never fetched (what would a free A cost?)
#else
    constexpr int LOADS = A_LD + B_LD;
#endif
    constexpr int STAGE = (BM + BN) * 64;         // halves per slab
    constexpr int SC = BN + 8;
    static_assert((NW == 4 || NW == 8) && BM % (8 * NL) == 0 && BN % (8 * NL) == 0 && TM >= 1 && TN >= 1, "tile shape");
    static_assert(LOADS * (STAGES - 1) * KSUB < 64, "vmcnt is a 6-bit counter");
    constexpr int NSLOT = STAGES * KSUB;          // slabs in the ring
    constexpr int AHEAD = (STAGES - 1) * KSUB;    // prefetch distance in slabs

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f16* smem = reinterpret_cast<f16*>(smem_raw);
#ifdef SDOD_GEMM_ABLATE
    constexpr int dbg = SDOD_GEMM_ABLATE; // developer builds only (Makefile: lib/libsdod_abl<mask>.so)
#else
    constexpr int dbg = 0;
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_consumer = !SPEC || wave < NC;                                  // wave-uniform role
    const int lw = SPEC ? (wave >= NC ? wave - NC : wave) : wave;                 // index among the loading waves
    const int cw = SPEC ? (wave >= NC ? wave - NC : wave) : wave;                 // index among the consuming waves
    const int wm = cw / WN, wn = cw % WN;
    STAMP(0);

    const int nwg = p.tiles_m * p.tiles_n;
    const int lid = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    tile_of(p, lid, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.z;
    // per-image operands (the folded cross-attention GEMMs: one weight matrix per prompt of the batch)
    const f16* w_img = p.w;
    const float* bias_img = p.bias;
    const float* lns_img = p.ln_s;
    if (p.w_img_stride != 0) {
        const int img = fast_div(m0, p.mg_rpi, p.sh_rpi);
        w_img += (size_t)img * p.w_img_stride;
        if (bias_img != nullptr) bias_img += (size_t)img * p.vec_img_stride;
        if (lns_img != nullptr) lns_img += (size_t)img * p.vec_img_stride;
    }

    const int KT = p.K / BK;
    int kt_begin = 0, kt_end = KT;
    if (p.splits > 1) {
        kt_begin = split * p.kt_per_split;
        kt_end = min(KT, kt_begin + p.kt_per_split);
    }
    const int nkt = kt_end - kt_begin;

    // lane covers row (lane>>3) of an 8-row group and PHYSICAL chunk (lane&7); it must fetch the logical chunk that
    // the swizzle maps there: c = phys ^ (row & 7), and row & 7 == lane>>3 because groups start at multiples of 8
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;

    // Per-row gather state, computed once.  Rows mode is the same code with n_img = M, 1x1 "image", pixel stride lda.
    //   a_ro0/1 : element offset of tap (0,0) of this row in source 0 / 1 (+ this lane's chunk), 32-bit
    //   a_mask  : bit t set <=> tap t of this row exists (row < M and the pixel is inside the image)
    // Every slab then costs one scalar delta + a handful of VALU per DMA instruction (no divisions, no branches).
    int a_ro0[A_LD], a_ro1[A_LD], a_py[A_LD], a_px[A_LD], a_im[A_LD];
    unsigned a_mask[A_LD];
    const f16* b_row[B_LD];
    const int pad = p.ksize >> 1;
    const int hup = p.h_in << p.ups, wup = p.w_in << p.ups;
    auto setup_rows = [&]() {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int m = m0 + (i * NL + lw) * 8 + lrow;
            const bool row_ok = m < p.M;
            const int hw = p.h_out * p.w_out;
            const int mm = row_ok ? m : 0;
            const int img = fast_div(mm, p.mg_hw, p.sh_hw);
            const int rem = mm - img * hw;
            const int oy = fast_div(rem, p.mg_w, p.sh_w);
            const int py = oy * p.stride - pad, px = (rem - oy * p.w_out) * p.stride - pad;
            a_py[i] = py; a_px[i] = px; a_im[i] = img * p.h_in;
            const int pix = (img * p.h_in + py) * p.w_in + px;
            a_ro0[i] = pix * p.sa0 + lchunk * 8;
            a_ro1[i] = pix * p.sa1 + lchunk * 8;
            unsigned mask = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    const bool ok = row_ok & (r < p.ksize) & (sx < p.ksize) & ((unsigned)(py + r) < (unsigned)hup) &
                                    ((unsigned)(px + sx) < (unsigned)wup);
                    mask |= (ok ? 1u : 0u) << (r * p.ksize + sx);
                }
            a_mask[i] = mask;
        }
    #pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            if (!WQ) {
                const int n = n0 + (i * NL + lw) * 8 + lrow;
                b_row[i] = n < p.N ? w_img + (size_t)n * p.ldw + lchunk * 8 : zeros;
            } else {
                // uint8 weights: a slab row is 64 bytes = 4 chunks of 16; lanes 0..31 of a DMA instruction cover the same 8 rows
                // (lanes 32..63 idle, so the instruction count -- and the vmcnt arithmetic -- equals the fp16 form); chunks
                // are XOR-swizzled with (row >> 2) & 3 so that the 8-byte fragment reads of 16 rows hit 16 different bank pairs
                const int r = (i * NL + lw) * 8 + ((lane & 31) >> 2);
                const int n = n0 + r;
                const int lc = (lane & 3) ^ ((r >> 2) & 3);
                b_row[i] = n < p.N ? reinterpret_cast<const f16*>(reinterpret_cast<const unsigned char*>(w_img) + (size_t)n * p.ldw + lc * 16) : zeros;
            }
        }
    };
    const int cin = p.c0 + p.c1;
    // LEAN issue path for plain row-major operands (every Linear of the graphs): the DMA address is a wave-uniform 64-bit base
    // (operand + k, advanced in SGPRs once per slab) plus a per-lane 32-bit byte offset fixed for the whole kernel -- two or three
    // instructions per DMA instead of the ~10 of the gather below.  (The loaders are instruction-bound, not bandwidth-bound:
    // conv_halo_kernel's loaders went from 42 to 87-102 GB/s per CU when their per-DMA arithmetic was hoisted,
    // profiles/r02_halo_phases.txt.)  Rows past M / N are CLAMPED to the last row instead of zero-filled: they only feed
    // accumulators whose rows / columns the epilogue never stores.
    const bool lean = p.lean != 0;
    unsigned a_off[A_LD], b_off[B_LD];
    auto setup_lean = [&]() {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int m = min(m0 + (i * NL + lw) * 8 + lrow, p.M - 1);
            a_off[i] = ((unsigned)m * (unsigned)p.lda + (unsigned)lchunk * 8u) * 2u;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            if (!WQ) {
                const int n = min(n0 + (i * NL + lw) * 8 + lrow, p.N - 1);
                b_off[i] = ((unsigned)n * (unsigned)p.ldw + (unsigned)lchunk * 8u) * 2u;
            } else { // uint8 weights: 64-byte slab rows, lanes 0..31 cover the same 8 rows (see setup_rows)
                const int r = (i * NL + lw) * 8 + ((lane & 31) >> 2);
                const int n = min(n0 + r, p.N - 1);
                const int lc = (lane & 3) ^ ((r >> 2) & 3);
                b_off[i] = (unsigned)n * (unsigned)p.ldw + (unsigned)lc * 16u; // ldw in bytes
            }
        }
    };

    auto issue_tile = [&](int kt, int stage) {
        const int k0 = kt * BK;
        f16* sA = smem + stage * STAGE;
        f16* sB = sA + BM * 64;
        if (lean) {
            const f16* ab = p.a0 + k0;
            const f16* wb = WQ ? reinterpret_cast<const f16*>(reinterpret_cast<const unsigned char*>(w_img) + k0) : w_img + k0;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) lds_dma16_saddr(ab, a_off[i], sA + (i * NL + lw) * 8 * 64);
#pragma unroll
            for (int i = 0; i < B_LD; ++i)
                if (!WQ || lane < 32) lds_dma16_saddr(wb, b_off[i], sB + (i * NL + lw) * 8 * 64);
            return;
        }
        if (p.k_tail && k0 >= p.k_tail) {
            // tail segment: the skip connection's 1x1 conv reads the block input at the OUTPUT pixel (centre tap)
            const int kk = k0 - p.k_tail;
            const bool sec = kk >= p.tc0;
            const f16* src = sec ? p.t1 : p.t0;
            const int sa = sec ? p.tc1 : p.tc0;
            const int ccs = sec ? kk - p.tc0 : kk;
            const int centre = pad * p.ksize + pad;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const int ro = ((a_im[i] + a_py[i] + pad) * p.w_in + a_px[i] + pad) * sa + ccs + lchunk * 8;
                const bool ok = (a_mask[i] >> centre) & 1u;
                const f16* g = ok ? src + ro : zeros;
                __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(sA + (i * NL + lw) * 8 * 64), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < B_LD; ++i) {
                const f16* g = b_row[i] + (b_row[i] == zeros ? 0 : (WQ ? k0 >> 1 : k0));
                if (!WQ || lane < 32) __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(sB + (i * NL + lw) * 8 * 64), 16, 0, 0);
            }
            return;
        }
        const int tap = fast_div(k0, p.mg_cin, p.sh_cin); // wave-uniform scalar arithmetic
        const int cc = k0 - tap * cin;
        const int r = p.ksize == 3 ? tap / 3 : 0, sx = tap - r * p.ksize;
        const bool second = cc >= p.c0;
        const f16* src = second ? p.a1 : p.a0;
        const int sa = second ? p.sa1 : p.sa0;
        const int ccs = second ? cc - p.c0 : cc;
        if (dbg & 8) {
        } else if (!p.ups) {
            const int sdelta = (r * p.w_in + sx) * sa + ccs; // wave-uniform
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const int ro = (second ? a_ro1[i] : a_ro0[i]) + sdelta;
                const bool ok = (a_mask[i] >> tap) & 1u;
                const f16* g = ok ? src + ro : zeros;
                __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(sA + (i * NL + lw) * 8 * 64), 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const int yy = (a_py[i] + r) >> 1, xx = (a_px[i] + sx) >> 1;
                const int ro = ((a_im[i] + yy) * p.w_in + xx) * sa + ccs + lchunk * 8;
                const bool ok = (a_mask[i] >> tap) & 1u;
                const f16* g = ok ? src + ro : zeros;
                __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(sA + (i * NL + lw) * 8 * 64), 16, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
            const f16* g = b_row[i] + (b_row[i] == zeros ? 0 : (WQ ? k0 >> 1 : k0));
            if (!WQ || lane < 32) __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(sB + (i * NL + lw) * 8 * 64), 16, 0, 0);
        }
    };

    const int frag_row = lane & 15;
    const int frag_chunk = lane >> 4;
    float rs1[A_LD], rs2[A_LD]; // LayerNorm fold: this lane's share of sum(x), sum(x^2) of the rows it DMA-ed
    auto ln_accumulate = [&](const f16* sA) {
            // every lane re-reads the 16 bytes it DMA-ed into this slab (lane-linear image: conflict-free) -- all of
            // the row's K columns pass through here, so the row statistics cost one extra LDS read per slab
#pragma unroll
            for (int i = 0; i < A_LD; ++i) {
                const f16x8 v = *reinterpret_cast<const f16x8*>(sA + (i * NL + lw) * 8 * 64 + lane * 8);
                const f16x2 one2 = {(f16)1.0f, (f16)1.0f};
#pragma unroll
                for (int e = 0; e < 8; e += 2) { // v_dot2_f32_f16: two products + fp32 accumulate per instruction
                    const f16x2 pr = {v[e], v[e + 1]};
                    rs1[i] = __builtin_amdgcn_fdot2(pr, one2, rs1[i], false);
                    rs2[i] = __builtin_amdgcn_fdot2(pr, pr, rs2[i], false);
                }
            }
    };
    constexpr int SC_ = BN + 8;
    float* ln_stats = reinterpret_cast<float*>(smem_raw + (size_t)BM * SC_ * sizeof(f16)); // [BM][2] mean, rstd
    auto ln_finalize = [&]() {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            float a1 = rs1[i], a2 = rs2[i];
            a1 = oct_sum(a1); // (the eight lanes of a row's 128-byte slab line: DPP moves, no ds_bpermute on the kernel's tail)
            a2 = oct_sum(a2);
            if ((lane & 7) == 0) {
                const float mean = a1 / (float)p.K;
                float var = a2 / (float)p.K - mean * mean;
                var = var < 0.f ? 0.f : var;
                const int r = (i * NL + lw) * 8 + lrow;
                ln_stats[2 * r] = mean;
                ln_stats[2 * r + 1] = rsqrt_fast(var + p.ln_eps);
            }
        }
    };

    // per-column epilogue vectors (bias, bias2, LayerNorm-fold s) -> LDS by LDS-DMA (4 bytes per lane, no VGPR round trip, no
    // wait): issued BEFORE the first slab, so they are older than every slab and have landed when the first counted wait of
    // the main loop returns; the epilogue then has no dependent global loads.  Absent vectors / columns past N read the zero line.
    constexpr size_t RING_BYTES = (size_t)NSLOT * STAGE * sizeof(f16);
    constexpr size_t CTILE_BYTES = (size_t)BM * SC * sizeof(f16) + (size_t)BM * 2 * sizeof(float);
    float* colv = reinterpret_cast<float*>(smem_raw + (RING_BYTES > CTILE_BYTES ? RING_BYTES : CTILE_BYTES)); // [4][BN]
    {
        constexpr int CHUNKS = (BN + 63) / 64;
        const float* zf = reinterpret_cast<const float*>(zeros) + lane;
        for (int job = wave; job < 4 * CHUNKS; job += NW) { // wave-uniform
            const int vec = job / CHUNKS, q = job - vec * CHUNKS;
            const int c = q * 64 + lane, n = n0 + c;
            const float* base = vec == 0 ? ((bias_img != nullptr && !p.bias_on_m) ? bias_img : nullptr) : vec == 1 ? p.bias2
                              : vec == 2 ? (p.ln ? lns_img : nullptr) : (WQ ? p.w_scale : nullptr);
            const float* g = (base != nullptr && n < p.N) ? base + n : zf;
            if (c < BN) __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(colv + vec * BN + q * 64), 4, 0, 0);
        }
    }
    const int e_m = lane & 15;
    const int e_n = (lane >> 4) * 4;
    f16* sC = smem;
    // The residual pieces a thread will add in the store phase, requested EARLY (wave-specialised kernels, behind the main
    // loop's last barrier): the loads land while the consumers run the epilogue arithmetic instead of exposing a full load
    // latency in front of every store.
    constexpr int RES_K = SPEC ? (BM * (BN / 8) + NT - 1) / NT : 1;
    constexpr bool RES_PF = SPEC && RES_K <= 4 && TM * TN < 20;
    f16x8 rres[RES_K];
    bool res_ready = false;
    auto residual_prefetch = [&]() {
        if constexpr (RES_PF) {
            if (p.residual == nullptr || p.no_respf || p.geglu || (p.N % 8) || (p.ldo % 8) || (p.ldr % 8)) return;
            constexpr int CPR = BN / 8;
#pragma unroll
            for (int k = 0; k < RES_K; ++k) {
                const int idx = tid + k * NT;
                const int row = idx / CPR, ch = idx - row * CPR;
                const int m = m0 + row, n = n0 + ch * 8;
                rres[k] = (idx < BM * CPR && m < p.M && n < p.N) ? ldg8(p.residual + (size_t)m * p.ldr + n) : zero8();
            }
            res_ready = true;
        }
    };
    auto store_phase = [&]() {
    const int CPR = p.geglu ? BN / 16 : BN / 8; // 16-byte chunks per output tile row
        const int n_out = p.geglu ? p.N / 2 : p.N;
        const int n0_out = p.geglu ? n0 / 2 : n0;
        const bool vec_ok = (n_out % 8 == 0) && (p.ldo % 8 == 0) && (p.residual == nullptr || p.ldr % 8 == 0);
        if constexpr (RES_PF) {
            if (res_ready) { // same pieces, same order as the loop below; the residual is already in registers
#pragma unroll
                for (int k = 0; k < RES_K; ++k) {
                    const int idx = tid + k * NT;
                    const int row = idx / (BN / 8), ch = idx - row * (BN / 8);
                    const int m = m0 + row, n = n0 + ch * 8;
                    if (idx >= BM * (BN / 8) || m >= p.M || n >= p.N) continue;
                    f16x8 v = *reinterpret_cast<const f16x8*>(sC + row * SC + ch * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] + (float)rres[k][e]);
                    stg8(p.out + (size_t)m * p.ldo + n, v);
                }
                return;
            }
        }
        for (int idx = tid; idx < BM * CPR; idx += NT) {
            const int row = idx / CPR;
            const int ch = idx - row * CPR;
            const int m = m0 + row, n = n0_out + ch * 8;
            if (m >= p.M || n >= n_out) continue;
            f16x8 v = *reinterpret_cast<const f16x8*>(sC + row * SC + ch * 8);
            if (vec_ok) {
                if (p.residual != nullptr) {
                    const f16x8 rr = ldg8(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] + (float)rr[e]);
                }
                stg8(p.out + (size_t)m * p.ldo + n, v);
            } else {
                for (int e = 0; e < 8; ++e) {
                    if (n + e < n_out) {
                        float f = (float)v[e];
                        if (p.residual != nullptr) f += (float)p.residual[(size_t)m * p.ldr + n + e];
                        p.out[(size_t)m * p.ldo + n + e] = (f16)f;
                    }
                }
            }
        }
    };

    if constexpr (SPEC) {
        if (!is_consumer) {
            // ---------------- LOADER program (complete; shares only the barriers with the consumers) ----------------
            if (lean) setup_lean();
            else setup_rows();
#pragma unroll
            for (int i = 0; i < A_LD; ++i) rs1[i] = rs2[i] = 0.f;
            if (lean) {
                // Lean loop (every Linear): the operand pointers and the LDS address of the slab to issue are RUNNING scalars,
                // advanced behind the issue, so that between the barrier and the DMA instructions of the next slab there is
                // nothing but the instructions themselves (the generic form recomputes k0 -> 64-bit pointers -> slot address there:
                // ~12 scalar instructions per slab on the path that paces a short-K loop)
                constexpr unsigned SLAB_BYTES = (unsigned)STAGE * 2u, RING_BYTES_U = (unsigned)NSLOT * SLAB_BYTES;
                const unsigned smem_base = (unsigned)(uintptr_t)(lds_void_ptr)smem;
                const unsigned a_dst0 = smem_base + (unsigned)(lw * 8 * 64) * 2u, b_dst0 = a_dst0 + (unsigned)(BM * 64) * 2u;
                const char* a_ptr = reinterpret_cast<const char*>(p.a0) + (size_t)kt_begin * BK * 2;
                const char* w_ptr = reinterpret_cast<const char*>(w_img) + (size_t)kt_begin * BK * (WQ ? 1 : 2);
                unsigned slot_off = 0;
                auto issue_lean = [&]() {
#pragma unroll
                    for (int i = 0; i < A_LD; ++i) lds_dma16_saddr_raw(a_ptr, a_off[i], a_dst0 + slot_off + (unsigned)(i * NL * 8 * 64) * 2u);
#pragma unroll
                    for (int i = 0; i < B_LD; ++i)
                        if (!WQ || lane < 32) lds_dma16_saddr_raw(w_ptr, b_off[i], b_dst0 + slot_off + (unsigned)(i * NL * 8 * 64) * 2u);
                    a_ptr += BK * 2;
                    w_ptr += BK * (WQ ? 1 : 2);
                    slot_off = slot_off + SLAB_BYTES == RING_BYTES_U ? 0u : slot_off + SLAB_BYTES;
                };
#pragma unroll
                for (int s = 0; s < AHEAD; ++s)
                    if (s < nkt) issue_lean();
                STAMP(1);
                for (int it = 0; it < nkt; ++it) {
                    if (it % KSUB == 0) {
                        wait_younger<LOADS, (STAGES - 2) * KSUB>(max(0, nkt - it - KSUB));
                        __builtin_amdgcn_s_barrier();
                    }
                    if (it + AHEAD < nkt) issue_lean();
                    if (p.ln) ln_accumulate(smem + (it % NSLOT) * STAGE);
                }
            } else {
#pragma unroll
            for (int s = 0; s < AHEAD; ++s)
                if (s < nkt) issue_tile(kt_begin + s, s);
            STAMP(1);
            for (int it = 0; it < nkt; ++it) {
                if (it % KSUB == 0) {
                    // the slabs of this group (it .. it+KSUB-1) of THIS wave have landed once only younger ones are outstanding ...
                    wait_younger<LOADS, (STAGES - 2) * KSUB>(max(0, nkt - it - KSUB));
                    __builtin_amdgcn_s_barrier(); // ... and everybody's; the previous group is no longer read
                }
                // the next slab's DMA first, the LayerNorm-fold statistics of this one behind it: they read slot it % NSLOT, the DMA
                // fills the slot the consumers have just left -- and the loop is paced by how early the DMA goes out, not by them
                // (the statistics in front cost the LayerNorm-folded Linears 1-2.5 us each: profiles/r03_ln_fold_order.txt)
                if (it + AHEAD < nkt) issue_tile(kt_begin + it + AHEAD, (it + AHEAD) % NSLOT);
                if (p.ln) ln_accumulate(smem + (it % NSLOT) * STAGE);
            }
            }
            wait_vmcnt<0>();
            __syncthreads();
            STAMP(2);
            if (p.ln) {
                ln_finalize();
                __syncthreads();
            }
            if (p.splits > 1) return;
            residual_prefetch();
            __syncthreads(); // the consumers have staged the output tile
            STAMP(3);
            store_phase();
#ifdef SDOD_GEMM_STAMP
            wait_vmcnt<0>();
            STAMP(4);
#endif
            return;
        }
    }

    // ---------------- unified program (SPEC = false) / CONSUMER program (SPEC = true) ----------------
    // uint8 weights: z = 1024 - offset of the weight rows this lane reads fragments of (w_off holds offset + 128); requested
    // before the first slab so that the load is the oldest thing in flight
    f16 wq_z[TN];
    if constexpr (WQ) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 16 + frag_row;
            wq_z[j] = (f16)(1152.0f - (n < p.N ? p.w_off[n] : 128.0f));
        }
    } else {
#pragma unroll
        for (int j = 0; j < TN; ++j) wq_z[j] = (f16)0.f;
    }
    if constexpr (!SPEC) {
        if (lean) setup_lean();
        else setup_rows();
#pragma unroll
        for (int i = 0; i < A_LD; ++i) rs1[i] = rs2[i] = 0.f;
        // prologue: STAGES-1 slabs in flight
#pragma unroll
        for (int s = 0; s < AHEAD; ++s)
            if (s < nkt) issue_tile(kt_begin + s, s);
    }
    STAMP(1);

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Wave-specialised consumers of the small / medium tiles: SOFTWARE-PIPELINED K halves with the fragment reads INTERLEAVED
    // between the MFMAs (as in conv_halo_kernel, where the same change took the consumer-only pace of a 128x80 tile from 0.33 to
    // 0.28 us per slab): the reads of half h+1 are requested while half h is multiplied, across the slab boundary too
    // (lgkmcnt(0) -> barrier -> next slab's first reads -> this slab's last MFMAs).  A wave issues in order and an MFMA holds
    // its issue port for 8 of its 16 cycles, so a read placed between two MFMAs is free, while a block of reads in front of the
    // MFMA block leaves the matrix pipe idle.  The consumers issue no vector-memory instruction, so the compiler can count
    // the LDS waits (lgkmcnt(N)) once it knows nothing else is pending: the builtin wait for everything in front of the loop.
    constexpr bool PIPELINED = SPEC && !WQ && TM * TN < 32 && dbg == 0;
    if constexpr (PIPELINED) {
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        f16x8 fa[2][TM], fb[2][TN];
        const unsigned smem_base = (unsigned)(uintptr_t)(lds_void_ptr)smem;
        unsigned a_addr[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) a_addr[i] = smem_base + (unsigned)lds_off(wm * WTM + i * 16 + frag_row, frag_chunk) * 2u;
        const unsigned b_addr0 = smem_base + (unsigned)(BM * 64 + lds_off(wn * WTN + frag_row, frag_chunk)) * 2u;
        auto lds16 = [](unsigned addr) { return *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>((uintptr_t)addr); };
        auto read_half = [&](auto b_c, auto ks_c, int slot) {
            constexpr int b = decltype(b_c)::value, ks = decltype(ks_c)::value;
            const unsigned base = (unsigned)slot * (unsigned)(STAGE * 2);
            const unsigned sb = (b_addr0 + base) ^ (ks << 6);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[b][i] = lds16((a_addr[i] + base) ^ (ks << 6));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[b][j] = lds16(sb + j * 16 * 128);
        };
        auto mfma_half = [&](auto b_c) {
            constexpr int b = decltype(b_c)::value;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = mfma16(fb[b][j], fa[b][i], acc[i][j]);
        };
        auto interleave_reads_with_mfmas = [] { // scheduling directive for the region since the last sched_barrier
            constexpr int NRD = TM + TN, NMF = TM * TN, PAIRS = NRD < NMF ? NRD : NMF;
            __builtin_amdgcn_sched_group_barrier(0x002, TM + 3, 0); // the address arithmetic of the reads first
            static_for<PAIRS>([](auto) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // one LDS read
            });
            if constexpr (NMF > PAIRS) __builtin_amdgcn_sched_group_barrier(0x008, NMF - PAIRS, 0);
            if constexpr (NRD > PAIRS) __builtin_amdgcn_sched_group_barrier(0x100, NRD - PAIRS, 0);
        };
        __builtin_amdgcn_s_waitcnt(0); // (every counter: also retires the epilogue vectors' LDS-DMA, a "pending flat" to the compiler)
        __builtin_amdgcn_s_barrier();  // the first slab group is in LDS (the loaders waited for it)
        int slot = 0;
        read_half(I0{}, I0{}, 0);
        for (int it = 0; it < nkt; ++it) {
            read_half(I1{}, I1{}, slot);
            mfma_half(I0{});
            interleave_reads_with_mfmas();
            __builtin_amdgcn_sched_barrier(0);
            if (it + 1 < nkt) {
                if ((it + 1) % KSUB == 0) {
                    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): my reads of this slab group are done, its slots may be refilled
                    __builtin_amdgcn_s_barrier();
                }
                slot = slot + 1 == NSLOT ? 0 : slot + 1;
                __builtin_amdgcn_sched_barrier(0);
                read_half(I0{}, I0{}, slot);
            }
            mfma_half(I1{});
            interleave_reads_with_mfmas();
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        for (int it = 0; it < nkt; ++it) {
            // slab `it` has landed once at most the younger in-flight slabs remain outstanding
            if (it % KSUB == 0) {
                if constexpr (!SPEC) wait_younger<LOADS, (STAGES - 2) * KSUB>(max(0, nkt - it - KSUB));
                __builtin_amdgcn_s_barrier(); // everyone's slab group is in LDS; everyone is done reading the previous group
            }

            // fragment reads for the whole slab first, then the DMA issue for slab it+STAGES-1 (its address arithmetic and
            // VMEM issue run under the LDS latency), then one uninterrupted MFMA cluster
            const f16* sA = smem + (it % NSLOT) * STAGE;
            const f16* sB = sA + BM * 64;
            // Big consumer tiles (>= 32 accumulator quads: 128 registers) cannot also hold the fragments of both K halves of the
            // slab: they read and multiply one half at a time (below); everyone else reads the whole slab first.
            constexpr bool HALF_AT_A_TIME = TM * TN >= 32;
            f16x8 xa[2][TM], wb[2][TN];
            if (HALF_AT_A_TIME) {
                // nothing to pre-read
            } else if (!(dbg & 4) ) {
    #pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
    #pragma unroll
                for (int i = 0; i < TM; ++i)
                    xa[ks][i] = *reinterpret_cast<const f16x8*>(sA + lds_off(wm * WTM + i * 16 + frag_row, ks * 4 + frag_chunk));
    #pragma unroll
                for (int j = 0; j < TN; ++j)
                    wb[ks][j] = WQ ? wq_frag(sB, wn * WTN + j * 16 + frag_row, ks, frag_chunk, wq_z[j])
                                     : *reinterpret_cast<const f16x8*>(sB + lds_off(wn * WTN + j * 16 + frag_row, ks * 4 + frag_chunk));
            }
            } else {
    #pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
    #pragma unroll
                    for (int i = 0; i < TM; ++i) xa[ks][i] = zero8();
    #pragma unroll
                    for (int j = 0; j < TN; ++j) wb[ks][j] = zero8();
                }
            }
            if constexpr (!SPEC) {
                if (p.ln) ln_accumulate(sA);
            }
            if constexpr (!SPEC) {
                if (it + AHEAD < nkt && !(dbg & 2)) issue_tile(kt_begin + it + AHEAD, (it + AHEAD) % NSLOT);
            }
            if (HALF_AT_A_TIME) {
                {
    #pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
    #pragma unroll
                        for (int i = 0; i < TM; ++i)
                            xa[0][i] = *reinterpret_cast<const f16x8*>(sA + lds_off(wm * WTM + i * 16 + frag_row, ks * 4 + frag_chunk));
    #pragma unroll
                        for (int j = 0; j < TN; ++j)
                            wb[0][j] = WQ ? wq_frag(sB, wn * WTN + j * 16 + frag_row, ks, frag_chunk, wq_z[j])
                                            : *reinterpret_cast<const f16x8*>(sB + lds_off(wn * WTN + j * 16 + frag_row, ks * 4 + frag_chunk));
    #pragma unroll
                        for (int j = 0; j < TN; ++j)
    #pragma unroll
                            for (int i = 0; i < TM; ++i) acc[i][j] = mfma16(wb[0][j], xa[0][i], acc[i][j]);
                        __builtin_amdgcn_sched_barrier(0); // keep the second half's fragment reads behind these MFMAs (registers)
                    }
                }
            } else if (!(dbg & 1) ) {
                __builtin_amdgcn_s_setprio(1);
    #pragma unroll
                for (int ks = 0; ks < 2; ++ks)
    #pragma unroll
                    for (int j = 0; j < TN; ++j)
    #pragma unroll
                        for (int i = 0; i < TM; ++i) acc[i][j] = mfma16(wb[ks][j], xa[ks][i], acc[i][j]);
                __builtin_amdgcn_s_setprio(0);
            } else {
                // keep the fragment reads alive without the matrix pipe
    #pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
    #pragma unroll
                    for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(wb[ks][j]));
    #pragma unroll
                    for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(xa[ks][i]));
                }
            }
        }
    }
    wait_vmcnt<0>();
    __syncthreads(); // all fragment reads done before the epilogue tile overwrites the ring
    STAMP(2);

    if (p.ln) {
        if constexpr (!SPEC) ln_finalize();
        __syncthreads();
    }
    if (WQ) {
        // affine-uint8 weights: acc holds sum_k A (q + offset_n), exactly the integer codes; out = scale_n * acc
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(colv + 3 * BN + wn * WTN + j * 16 + e_n);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] *= sc[r];
        }
    }
    if (p.splits > 1) {
        float* slab = p.partial + (size_t)split * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + e_m;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + e_n;
                if (n + 3 < p.N) {
                    *reinterpret_cast<f32x4*>(slab + (size_t)m * p.N + n) = acc[i][j];
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N) slab[(size_t)m * p.N + n + r] = acc[i][j][r];
                }
            }
        }
        return;
    }

    if constexpr (SPEC) residual_prefetch();
    if (p.geglu) {
        // 16-column blocks alternate [value | gate]; both live in the SAME lane (acc[i][j], acc[i][j+1]), so GEGLU is a
        // register-level product and the tile that goes to memory is half as wide
        if constexpr (TN % 2 == 0) { // odd-TN tiles (21, 22) never get a GEGLU descriptor: make_plan + the host check
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ml = wm * WTM + i * 16 + e_m;
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    const int nl = wn * WTN + j * 16 + e_n; // column of the value block in W-row space
                    f16x4 h;
                    const f32x4 ba = *reinterpret_cast<const f32x4*>(colv + nl), bg = *reinterpret_cast<const f32x4*>(colv + nl + 16);
                    const f32x4 sa = *reinterpret_cast<const f32x4*>(colv + 2 * BN + nl), sg = *reinterpret_cast<const f32x4*>(colv + 2 * BN + nl + 16);
                    // rstd * (alpha * acc - mean * s) + b as two fma: acc * (rstd * alpha) + (b - rstd * mean * s)
                    const float ra = p.ln ? ln_stats[2 * ml + 1] * p.alpha : p.alpha, mr = p.ln ? -ln_stats[2 * ml + 1] * ln_stats[2 * ml] : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float va = fmaf(acc[i][j][r], ra, p.ln ? fmaf(mr, sa[r], ba[r]) : ba[r]); // (the s vectors exist only with ln)
                        const float vg = fmaf(acc[i][j + 1][r], ra, p.ln ? fmaf(mr, sg[r], bg[r]) : bg[r]);
                        h[r] = (f16)(va * gelu_erf_f(vg));
                    }
                    *reinterpret_cast<f16x4*>(sC + ml * SC + (wn * WTN + j * 16) / 2 + e_n) = h;
                }
            }
        }
    } else {
        // Staged as a few short passes over the accumulators instead of one big unrolled body: the common case (bias only)
        // executes ~10 instructions per accumulator quad; the rare features (LayerNorm fold, per-row bias, per-image row
        // bias, activation) are wave-uniform branches around their own small loops.  (The one-body form was 25 KB of
        // straight-line code -- every (i, j) carried all variants -- and cost 3.4 us per workgroup in instruction fetch.)
        const float alpha = p.alpha;
        if (p.ln) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ml = wm * WTM + i * 16 + e_m;
                const float mean = ln_stats[2 * ml], rstd = ln_stats[2 * ml + 1];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f32x4 sv = *reinterpret_cast<const f32x4*>(colv + 2 * BN + wn * WTN + j * 16 + e_n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = rstd * (acc[i][j][r] * alpha - mean * sv[r]);
                }
            }
        } else if (alpha != 1.0f) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] *= alpha;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * WTN + j * 16 + e_n;
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(colv + nl), b2 = *reinterpret_cast<const f32x4*>(colv + BN + nl);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = (acc[i][j][r] + b1[r]) + b2[r];
        }
        if (p.bias != nullptr && p.bias_on_m) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * WTM + i * 16 + e_m;
                const float bm = m < p.M ? p.bias[m] : 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += bm;
            }
        }
        if (p.row_bias != nullptr) {
            const bool rb_vec = (p.ldrb & 3) == 0 && ((uintptr_t)p.row_bias & 7) == 0 && (p.N & 3) == 0;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * WTM + i * 16 + e_m;
                const f16* rbias = p.row_bias + (size_t)((m < p.M ? m : 0) / p.rows_per_img) * p.ldrb;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * WTN + j * 16 + e_n;
                    if (rb_vec) {
                        if (n < p.N) {
                            const f16x4 t = *reinterpret_cast<const f16x4*>(rbias + n);
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)t[r];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < p.N) acc[i][j][r] += (float)rbias[n + r];
                    }
                }
            }
        }
        if (p.act != ACT_NONE) {
#pragma unroll 1
            for (int pass = 0; pass < 1; ++pass) { // (keeps the three variants out of line with the common path)
                if (p.act == ACT_SILU) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[i][j][r] = silu_f(acc[i][j][r]);
                } else if (p.act == ACT_GELU) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[i][j][r] = gelu_erf_f(acc[i][j][r]);
                } else {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[i][j][r] = quick_gelu_f(acc[i][j][r]);
                }
            }
        }
        if constexpr (WTN == 80 && !WQ) {
            // Row softmax over the wave's 80 columns (the folded cross-attention: one head's 77 keys + 3 padding columns whose
            // bias is -30000, scores already in log2 units): a row's columns sit in this lane (TN blocks x 4) and in the three
            // lanes 16 / 32 / 48 away -- two row swaps per reduction (quad_rows_max / _sum), nothing leaves the wave.
            if (p.softmax_g) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float mx = acc[i][0][0];
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, acc[i][j][r]);
                    mx = quad_rows_max(mx);
                    float sum = 0.f;
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float e = __builtin_amdgcn_exp2f(acc[i][j][r] - mx);
                            acc[i][j][r] = e;
                            sum += e;
                        }
                    sum = quad_rows_sum(sum);
                    const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] *= inv;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ml = wm * WTM + i * 16 + e_m;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nl = wn * WTN + j * 16 + e_n;
                f16x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = (f16)acc[i][j][r];
                *reinterpret_cast<f16x4*>(sC + ml * SC + nl) = h;
            }
        }
    }
    __syncthreads();
    STAMP(3);

    store_phase();
#ifdef SDOD_GEMM_STAMP
    wait_vmcnt<0>();
    STAMP(4);
#endif
}

// ------------------------------------------------------------------------------------------------
// A-PANEL kernel: the short-K, wide-N Linear layers of the transformer blocks (to_q|to_k|to_v with N = 3C, the GEGLU
// projection with N = 8C; K = C = 320 / 640, LayerNorm folded in).  In the ring kernel above these are the launches furthest
// from the matrix pipe: a 128x128 tile has 5-10 slabs of K, so a workgroup lives for a handful of slabs between a prologue
// that waits for its first bytes and an epilogue that costs more than the loop (GEGLU at K = 320: 2.4 us of loop, 2.9 us of
// epilogue -- 370 M lane-operations of LayerNorm fold / bias / GELU per launch, as much VALU time as there is MFMA time).  Here
//   * a workgroup owns a row PANEL of BM rows x the whole K (80 KB: 128 x 320 or 64 x 640 halves), fetched into LDS ONCE, and
//     walks several consecutive n-tiles against it: per slab only the W half (BN x 128 bytes) crosses L2 -> LDS, and the
//     LayerNorm row statistics are computed once per panel instead of once per tile;
//   * the W slabs of ALL its tiles are one continuous stream through a ring (4 loader waves, as above): no tile after the
//     first waits for first bytes;
//   * TWO consumer groups of 4 waves (one wave of each per SIMD) take the tiles alternately: while group X multiplies tile
//     j + 1, group Y runs the epilogue of tile j on the SIMDs' vector pipes -- the matrix pipe never waits for an epilogue
//     (profiles/r03_panel_phases.txt: with one group the K loops were 12 of a workgroup's 30 us);
//   * the epilogue goes from the accumulators straight to global memory: v_permlane16_swap pairs the 4-column pieces of two
//     lane groups into 16-byte row segments; no LDS staging tile, so the only workgroup synchronisation is the one barrier per
//     slab -- which the group in its epilogue, and the idle group of the first tile, take part in at the same cadence (KT
//     barriers per tile from every wave).
// Counted vmcnt waits (conservative by the few epilogue-vector DMAs at a tile boundary), XOR-swizzled slab images and fragment
// reads exactly as in gemm_glds_kernel.  Rows mode, fp16 weights, no split-K.
template <int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(768) void gemm_apanel_kernel(const GemmP p, const f16* __restrict__ zeros) {
    constexpr int NL = 4;                                   // loader waves (8..11); consumer groups are waves 0..3 and 4..7
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int A_LD = BM / (8 * NL), B_LD = BN / (8 * NL);
    constexpr int AHEAD = STAGES - 1;
    constexpr int ASLAB = BM * 64, WSLAB = BN * 64;         // halves per slab of the panel / of the ring
    static_assert(WM * WN == 4 && BM % 32 == 0 && BN % 64 == 0 && BN <= 128 && TM >= 1 && TN >= 2 && TN % 2 == 0, "tile shape");
    static_assert(B_LD * AHEAD + 2 < 64, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int KT = p.K / BK;
    f16* sAp = reinterpret_cast<f16*>(smem_raw);            // [KT][BM][64]
    f16* sW = sAp + (size_t)KT * ASLAB;                     // [STAGES][BN][64]
    float* colv = reinterpret_cast<float*>(sW + (size_t)STAGES * WSLAB); // [3 tiles][bias | ln_s][BN]
    float* ln_stats = colv + 6 * BN;                        // [BM][2] mean, rstd

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // workgroup -> (row panel, n-tile group).  Consecutive logical ids share an XCD (xcd_remap): the groups of one panel when
    // the activations are the bigger operand (each L2 fetches its panels once, and all of W), the panels of one group when W is.
    const int lid = xcd_remap(blockIdx.x, (int)gridDim.x);
    int pm, ng;
    if (p.ap_nmajor) {
        ng = lid / p.ap_panels;
        pm = lid - ng * p.ap_panels;
    } else {
        pm = lid / p.ap_groups;
        ng = lid - pm * p.ap_groups;
    }
    const int t_begin = ng * p.ap_tpg;
    const int T = min(p.tiles_n, t_begin + p.ap_tpg) - t_begin; // n-tiles of this workgroup
    if (T <= 0) return;
    STAMPW(0, 0);
    const int m0 = pm * BM;
    const int total = T * KT;                                    // W slabs this workgroup streams
    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;

    if (wave >= 8) {
        // ---------------- LOADER program ----------------
        const int lw = wave - 8;
        unsigned a_off[A_LD], b_off[B_LD];
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
            const int m = min(m0 + (i * NL + lw) * 8 + lrow, p.M - 1);
            a_off[i] = ((unsigned)m * (unsigned)p.lda + (unsigned)lchunk * 8u) * 2u;
        }
        // the whole panel: KT x A_LD DMA instructions per wave, oldest things in flight
        for (int kt = 0; kt < KT; ++kt) {
            const f16* ab = p.a0 + kt * BK;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) lds_dma16_saddr(ab, a_off[i], sAp + (size_t)kt * ASLAB + (i * NL + lw) * 8 * 64);
        }
        int nj = 0, nit = 0, nslot = 0, ncv = 0; // next slab to issue: tile, K slab, ring slot; epilogue-vector buffer of its tile
        auto issue_next = [&]() {
            const int n0 = (t_begin + nj) * BN;
            if (nit == 0) {
                // a new tile: its per-column epilogue vectors (bias | LayerNorm-fold s) -> colv[nj % 3], older than its first
                // slab (three buffers: tile j's are read by its epilogue during the K loop of tile j + 1, while the loaders are
                // already fetching those of tile j + 2); and the per-lane weight row offsets (rows past N clamp to the last one:
                // they feed columns nobody stores)
                const int vec = lw >> 1, q = lw & 1;
                if (q * 64 < BN) {
                    const int n = n0 + q * 64 + lane;
                    const float* base = vec == 0 ? p.bias : (p.ln ? p.ln_s : nullptr);
                    const float* g = (base != nullptr && n < p.N) ? base + n : reinterpret_cast<const float*>(zeros) + lane;
                    __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(colv + (ncv * 2 + vec) * BN + q * 64), 4, 0, 0);
                }
                ncv = ncv + 1 == 3 ? 0 : ncv + 1;
#pragma unroll
                for (int i = 0; i < B_LD; ++i) {
                    const int n = min(n0 + (i * NL + lw) * 8 + lrow, p.N - 1);
                    b_off[i] = ((unsigned)n * (unsigned)p.ldw + (unsigned)lchunk * 8u) * 2u;
                }
            }
            const f16* wb = p.w + nit * BK;
            f16* dst = sW + (size_t)nslot * WSLAB;
#pragma unroll
            for (int i = 0; i < B_LD; ++i) lds_dma16_saddr(wb, b_off[i], dst + (i * NL + lw) * 8 * 64);
            if (++nit == KT) { nit = 0; ++nj; }
            nslot = nslot + 1 == STAGES ? 0 : nslot + 1;
        };
#pragma unroll
        for (int s = 0; s < AHEAD; ++s)
            if (s < total) issue_next();
        for (int g = 0; g < total; ++g) {
            // slab g (and everything older: the panel, the tile's vectors) of THIS wave has landed once only younger slabs are
            // outstanding; an epilogue-vector DMA among the younger ones makes the wait one instruction stricter, never looser
            wait_younger<B_LD, STAGES - 2>(max(0, total - g - 1));
            __builtin_amdgcn_s_barrier(); // ... and everybody's; the consumers have left slab g - 1
            if (g == 0 && p.ln) {
                // LayerNorm fold: row statistics of the panel, once: every lane re-reads the 16 bytes it DMA-ed per slab
                float rs1[A_LD], rs2[A_LD];
#pragma unroll
                for (int i = 0; i < A_LD; ++i) rs1[i] = rs2[i] = 0.f;
                for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                    for (int i = 0; i < A_LD; ++i) {
                        const f16x8 v = *reinterpret_cast<const f16x8*>(sAp + (size_t)kt * ASLAB + (i * NL + lw) * 8 * 64 + lane * 8);
                        const f16x2 one2 = {(f16)1.0f, (f16)1.0f};
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            const f16x2 pr = {v[e], v[e + 1]};
                            rs1[i] = __builtin_amdgcn_fdot2(pr, one2, rs1[i], false);
                            rs2[i] = __builtin_amdgcn_fdot2(pr, pr, rs2[i], false);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < A_LD; ++i) {
                    float a1 = rs1[i], a2 = rs2[i];
                    a1 = oct_sum(a1);
                    a2 = oct_sum(a2);
                    if ((lane & 7) == 0) {
                        const float mean = a1 / (float)p.K;
                        float var = a2 / (float)p.K - mean * mean;
                        var = var < 0.f ? 0.f : var;
                        const int r = (i * NL + lw) * 8 + lrow;
                        ln_stats[2 * r] = mean;
                        ln_stats[2 * r + 1] = rsqrt_fast(var + p.ln_eps);
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the statistics are in LDS before this wave meets the next barrier
            }
            if (g + AHEAD < total) issue_next();
        }
        wait_vmcnt<0>();
        STAMPW(7, 8);
        return;
    }

    // ---------------- CONSUMER program (two groups, alternating tiles) ----------------
    const int grp = wave >> 2, cw = wave & 3;
    const int wm = cw / WN, wn = cw % WN;
    const int frag_row = lane & 15, frag_chunk = lane >> 4;
    const int e_m = lane & 15, e_g = lane >> 4;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    f16x8 fa[2][TM], fb[2][TN];
    const unsigned a_base = (unsigned)(uintptr_t)(lds_void_ptr)sAp, w_base = (unsigned)(uintptr_t)(lds_void_ptr)sW;
    unsigned a_addr[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_addr[i] = a_base + (unsigned)lds_off(wm * WTM + i * 16 + frag_row, frag_chunk) * 2u;
    const unsigned b_addr0 = w_base + (unsigned)lds_off(wn * WTN + frag_row, frag_chunk) * 2u;
    auto lds16 = [](unsigned addr) { return *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>((uintptr_t)addr); };
    f32x4 acc[TM][TN];
    auto read_half = [&](auto b_c, auto ks_c, int it, int slot) {
        constexpr int b = decltype(b_c)::value, ks = decltype(ks_c)::value;
        const unsigned abase = (unsigned)it * (unsigned)(ASLAB * 2);
        const unsigned sb = (b_addr0 + (unsigned)slot * (unsigned)(WSLAB * 2)) ^ (ks << 6);
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[b][i] = lds16((a_addr[i] + abase) ^ (ks << 6));
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[b][j] = lds16(sb + j * 16 * 128);
    };
    auto mfma_half = [&](auto b_c) {
        constexpr int b = decltype(b_c)::value;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = mfma16(fb[b][j], fa[b][i], acc[i][j]);
    };
    auto interleave_reads_with_mfmas = [] {
        constexpr int NRD = TM + TN, NMF = TM * TN, PAIRS = NRD < NMF ? NRD : NMF;
        __builtin_amdgcn_sched_group_barrier(0x002, TM + 3, 0);
        static_for<PAIRS>([](auto) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        });
        if constexpr (NMF > PAIRS) __builtin_amdgcn_sched_group_barrier(0x008, NMF - PAIRS, 0);
        if constexpr (NRD > PAIRS) __builtin_amdgcn_sched_group_barrier(0x100, NRD - PAIRS, 0);
    };
    const float alpha = p.alpha;
    const int n_out = p.geglu ? p.N / 2 : p.N;
    // 4 + 4 packed halves of two lane groups -> one 16-byte row segment per lane: lane group g holds columns 4g .. 4g + 3 of
    // 16-column blocks X and Y; after the two swaps an even group holds columns 4g .. 4g + 7 of X, an odd group columns
    // 4(g - 1) .. 4g + 3 of Y (v_permlane16_swap: odd rows of the first operand <-> even rows of the second)
    auto pair_store = [&](f16x4 hx, f16x4 hy, int m, int col_x, int col_y, int limit) {
        const u32x2 ux = __builtin_bit_cast(u32x2, hx), uy = __builtin_bit_cast(u32x2, hy);
        const auto s0 = __builtin_amdgcn_permlane16_swap(ux[0], uy[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(ux[1], uy[1], false, false);
        const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
        const int col = (e_g & 1) ? col_y + 4 * (e_g - 1) : col_x + 4 * e_g;
        if (m < p.M && col < limit) *reinterpret_cast<u32x4*>(p.out + (size_t)m * p.ldo + col) = v;
    };
    // Epilogue of tile jt from this group's accumulators, in UNITS of two 16-column blocks of one 16-row block; with BARS the
    // group takes part in the KT slab barriers of the tile the other group is multiplying meanwhile, spread over the units.
    auto epilogue = [&](auto bars_c, int jt, int cv) {
        constexpr bool BARS = decltype(bars_c)::value;
        const int n0 = (t_begin + jt) * BN;
        const float* cvb = colv + cv * 2 * BN; // bias
        const float* cvs = cvb + BN;           // LayerNorm-fold s
        auto unit_done = [&](int u, int nu) {
            if constexpr (BARS) {
                const int nb = ((u + 1) * KT) / nu - (u * KT) / nu;
                for (int k = 0; k < nb; ++k) __builtin_amdgcn_s_barrier();
            }
        };
        if (p.geglu) {
            constexpr int NU = TM * (TN / 4 > 0 ? TN / 4 : 1);
            static_assert(TN % 4 == 0, "GEGLU pairs two output blocks = four accumulator blocks per unit");
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ml = wm * WTM + i * 16 + e_m, m = m0 + ml;
                // rstd * (alpha * acc - mean * s) + b as two fma: acc * (rstd * alpha) + (b - rstd * mean * s)
                const float ra = p.ln ? ln_stats[2 * ml + 1] * alpha : alpha, mr = p.ln ? -ln_stats[2 * ml + 1] * ln_stats[2 * ml] : 0.f;
#pragma unroll
                for (int j = 0; j < TN; j += 4) {
                    f16x4 h[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int jj = j + 2 * q;
                        const int nl = wn * WTN + jj * 16 + 4 * e_g; // column of the value block in W-row space
                        const f32x4 ba = *reinterpret_cast<const f32x4*>(cvb + nl), bg = *reinterpret_cast<const f32x4*>(cvb + nl + 16);
                        const f32x4 sa = *reinterpret_cast<const f32x4*>(cvs + nl), sg = *reinterpret_cast<const f32x4*>(cvs + nl + 16);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float va = fmaf(acc[i][jj][r], ra, p.ln ? fmaf(mr, sa[r], ba[r]) : ba[r]);
                            const float vg = fmaf(acc[i][jj + 1][r], ra, p.ln ? fmaf(mr, sg[r], bg[r]) : bg[r]);
                            h[q][r] = (f16)(va * gelu_erf_f(vg));
                        }
                    }
                    const int c0 = (n0 + wn * WTN + j * 16) / 2;
                    pair_store(h[0], h[1], m, c0, c0 + 16, n_out);
                    unit_done(i * (TN / 4) + j / 4, NU);
                }
            }
        } else {
            constexpr int NU = TM * TN / 2;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ml = wm * WTM + i * 16 + e_m, m = m0 + ml;
                const float mean = p.ln ? ln_stats[2 * ml] : 0.f, rstd = p.ln ? ln_stats[2 * ml + 1] : 1.f;
#pragma unroll
                for (int j = 0; j < TN; j += 2) {
                    f16x4 h[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int nl = wn * WTN + (j + q) * 16 + 4 * e_g, n = n0 + nl;
                        const f32x4 b1 = *reinterpret_cast<const f32x4*>(cvb + nl);
                        const f32x4 sv = *reinterpret_cast<const f32x4*>(cvs + nl);
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float f = acc[i][j + q][r] * alpha;
                            if (p.ln) f = rstd * (f - mean * sv[r]);
                            v[r] = f + b1[r];
                        }
                        if (p.act != ACT_NONE) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
                        }
                        if (p.residual != nullptr && m < p.M && n < p.N) {
                            const f16x4 rr = *reinterpret_cast<const f16x4*>(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = (float)(f16)v[r] + (float)rr[r];
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[q][r] = (f16)v[r];
                    }
                    const int c0 = n0 + wn * WTN + j * 16;
                    pair_store(h[0], h[1], m, c0, c0 + 16, n_out);
                    unit_done(i * (TN / 2) + j / 2, NU);
                }
            }
        }
    };

    int slot = 0; // ring slot of the next slab (every wave tracks it through the tiles it does not multiply, too)
    int cv = 0;   // epilogue-vector buffer of tile s
    for (int s = 0; s < T; ++s) {
        if ((s & 1) == grp) {
            // ---- multiply tile s
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_s_barrier(); // the tile's first slab is in LDS (the loaders waited for it); for s = 0 so is the panel
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0) STAMPW(1, 0);
            read_half(I0{}, I0{}, 0, slot);
            for (int it = 0; it < KT; ++it) {
                read_half(I1{}, I1{}, it, slot);
                mfma_half(I0{});
                interleave_reads_with_mfmas();
                __builtin_amdgcn_sched_barrier(0);
                if (it + 1 < KT) {
                    __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): my reads of this slab are done, its slot may be refilled
                    __builtin_amdgcn_s_barrier();
                    slot = slot + 1 == STAGES ? 0 : slot + 1;
                    __builtin_amdgcn_sched_barrier(0);
                    read_half(I0{}, I0{}, it + 1, slot);
                }
                mfma_half(I1{});
                interleave_reads_with_mfmas();
                __builtin_amdgcn_sched_barrier(0);
            }
            slot = slot + 1 == STAGES ? 0 : slot + 1;
            if (s == 0) STAMPW(2, 0);
            if (s == 2) STAMPW(4, 0);
        } else {
            // ---- the other group multiplies tile s: finish tile s - 1 meanwhile, meeting the KT slab barriers of tile s
            if (s >= 1) {
                epilogue(std::true_type{}, s - 1, cv == 0 ? 2 : cv - 1);
                if (s == 1) STAMPW(3, 0);
            } else {
                for (int k = 0; k < KT; ++k) __builtin_amdgcn_s_barrier();
            }
            slot = (slot + KT) % STAGES;
        }
        cv = cv + 1 == 3 ? 0 : cv + 1;
    }
    if (((T - 1) & 1) == grp) {
        if (grp == 0) STAMPW(5, 0);
        epilogue(std::false_type{}, T - 1, cv == 0 ? 2 : cv - 1); // nobody is multiplying any more: no barriers left to meet
        if (grp == 0) STAMPW(6, 0);
    }
}

// patch DMA rounds (256 lanes x 16 bytes each) a halo tile of bm output rows can need: (rows + 2) x (width + 2) pixels of 128
// bytes.  96- and 192-row tiles exist for images whose rows are multiples of 3 (SD v2.1-768: 96 / 48 / 24 / 12 pixels)
constexpr int halo_nrmax(int bm) { return bm <= 64 ? 7 : bm == 96 ? 10 : bm <= 128 ? 9 : 13; }
// ring depth of conv_halo_kernel's tail program: its slots hold A next to B, so fewer of them fit
constexpr int halo_tail_stages(int bm, int bn, int stages) {
    const int fit = (150 * 1024) / ((bm + bn) * 128);
    return fit < stages ? fit : stages;
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_halo_kernel: the 3x3 stride-1 convolution with the input HALO PATCH resident in LDS (tiles 37..).
//
// The implicit-im2col GEMM above re-fetches every input pixel nine times (once per tap) and its main loop is paced by the
// LDS-DMA stream: slab time = fixed part + bytes / ~75 GB/s per CU (tools/gemm_phases.py with and without the A operand,
// profiles/r02_gemm_phases_noA.txt).  Here K is walked CHANNEL-CHUNK major, tap minor: for each 64-channel chunk the
// workgroup holds the (rows + 2) x (width + 2) input patch of its output tile in LDS (zero halo included) and the nine
// taps of the chunk read their A fragments from that one patch at a shifted pixel offset -- the A operand crosses the
// L2 -> LDS path 2-4.5x less often (3 of 9 for a one-row tile, 18 of 144 for a 16x16 image), so tiles can be TALL and
// NARROW (128x80, 256x32: few weight bytes per output) where the im2col kernel needs them square.
//   * 4 LOADER waves stream the weight slabs (B: [BN][64] per tap and chunk) through a STAGES-deep ring exactly as in the
//     wave-specialised GEMM (counted vmcnt, one barrier per slab); the 8-row DMA groups of a slab are dealt round-robin, so
//     a wave may issue one group fewer than its neighbour (BN = 80: 3,3,2,2) -- the loop is instantiated per count.
//   * 4 CONSUMER waves own the accumulators.  They also fetch the patches: two double-buffered patch images; the DMA
//     rounds of chunk c+1 are issued two per tap while chunk c is multiplied, each wave waits for its own rounds
//     (vmcnt(0)) before the barrier that opens chunk c+1.  A patch pixel is one 128-byte line, 16-byte pieces XOR-swizzled
//     with the pixel index (the fragment rows of one MFMA operand are 16 consecutive pixels at any tap shift, so the reads
//     stay conflict-free exactly like the row-major slab).
//   * consumer pipeline: fragments of K-half h+1 are requested before the MFMAs of half h are issued, across the slab
//     boundary too (barrier -> next slab's first reads -> this slab's last MFMAs), so the LDS latency hides under the
//     matrix pipe instead of adding to it.
//   * a fused 1x1 skip connection (k_tail) has no halo to share: it becomes ONE EXTRA split-K slice (blockIdx.z ==
//     h_main_splits) whose workgroups run the plain ring program (A and B slabs by the loaders) on the tail columns; the
//     partial slabs meet in splitk_reduce_kernel like any split.
// Output rows of a tile are consecutive (a row segment, whole rows, or whole images), so the epilogue is the GEMM's.
//   * WQ = affine-uint8 weights (config 5): the weight slabs are 64-byte rows streamed as bytes (lanes 0..31 of a DMA
//     instruction), the consumers expand a fragment to the integers q + offset in fp16 right before its MFMAs (u8x8_to_f16:
//     8 VALU per 8 x 16 weights, amortised over the TM row blocks -- tiles with TM >= 4 hide it under the matrix pipe) and
//     the per-column scale is applied to the accumulators before anything else sees them.  No 1x1 tail (the host keeps the
//     skip convolution of a uint8 graph a separate GEMM: its tensor has its own encoding).
template <int BM, int BN, int WM, int WN, int STAGES, bool WQ = false>
__global__ __launch_bounds__(512) void conv_halo_kernel(const GemmP p, const f16* __restrict__ zeros) {
    static_assert(WM * WN == 4, "four consumer waves (one per SIMD) + four loader waves");
    constexpr int NL = 4;
    constexpr int NT = 512;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0 && BN % 8 == 0 && BM % 32 == 0, "tile shape");
    constexpr int BG = BN / 8;                  // 8-row DMA groups of one weight slab
    constexpr int B_LDX = (BG + NL - 1) / NL;   // ... per loader wave, at most
    constexpr int A_LD = BM / (8 * NL);         // tail program: A groups per loader wave
    constexpr int SC = BN + 8;
    constexpr int NRMAX = halo_nrmax(BM); // patch DMA rounds this tile can need
    constexpr int AHEAD = STAGES - 1;
#ifdef SDOD_GEMM_ABLATE
    constexpr int dbg = SDOD_GEMM_ABLATE; // developer builds: 16 no patch DMA inside the loop, 32 no weight DMA inside the loop, 64 no MFMA,
                                          // 128 no barrier inside the loop, 256 no fragment reads inside the loop
#else
    constexpr int dbg = 0;
#endif
    // the tail program's ring holds A next to B: as many of the STAGES slots as fit
    constexpr int TST = halo_tail_stages(BM, BN, STAGES);
    constexpr int TAHEAD = TST - 1;
    static_assert(TST >= 2 && B_LDX * AHEAD < 64 && (B_LDX + A_LD) * TAHEAD < 64, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f16* smem = reinterpret_cast<f16*>(smem_raw);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_consumer = wave < 4;
    const int lw = wave & 3, cw = wave & 3;
    const int wm = cw / WN, wn = cw % WN;
    STAMP(0);

    const int nwg = p.tiles_m * p.tiles_n;
    const int lid = xcd_remap(blockIdx.x, nwg);
    int tile_m, tile_n;
    tile_of(p, lid, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.z;
    const bool tail_wg = !WQ && p.k_tail != 0 && split == p.h_main_splits;
    const int cin = p.c0 + p.c1;
    // slabs of this workgroup: [kt_begin, kt_end) of the 9 * (cin / 64) tap slabs, or all tail slabs
    int kt_begin, kt_end;
    if (tail_wg) {
        kt_begin = 0;
        kt_end = (p.K - p.k_tail) / BK;
    } else {
        const int ktm = 9 * (cin / BK);
        kt_begin = split * p.kt_per_split;
        kt_end = min(ktm, kt_begin + p.kt_per_split);
    }
    const int nkt = kt_end - kt_begin;
    // CHAINED tail (h_chain: an unsplit convolution with a fused 1x1 skip): the workgroup walks the 3x3 taps and then, behind one
    // workgroup barrier, runs the tail program itself on the same accumulators -- no partial slabs, no reduce launch
    const bool chain = p.h_chain != 0 && !tail_wg;
    const int nkt_t = (p.K - p.k_tail) / BK; // slabs of the 1x1 tail

    const int lrow = lane >> 3;
    const int lchunk = (lane & 7) ^ lrow;
    const int frag_row = lane & 15;
    const int frag_chunk = lane >> 4;
    const int e_m = lane & 15;
    const int e_n = (lane >> 4) * 4;

    // per-column epilogue vectors -> LDS, issued before anything else (they are older than every slab / patch)
    float* colv = reinterpret_cast<float*>(smem_raw + p.h_colv_off); // [2][BN] fp32: bias, bias2; then [BN] fp16: the tile's row bias
    // the per-image row bias (time embedding) rides along when the tile lies inside ONE image: fetched in the epilogue it
    // exposes a full load latency per launch
    constexpr bool RB_LDS_OK = TM * TN < 20; // (the 128x160 tile has no register to spare for the extra path)
    const bool rb_lds = RB_LDS_OK && p.row_bias != nullptr && (p.N & 1) == 0 && (p.ldrb & 1) == 0 && ((uintptr_t)p.row_bias & 3) == 0 &&
                        m0 / p.rows_per_img == (min(m0 + BM, p.M) - 1) / p.rows_per_img;
    {
        constexpr int CHUNKS = (BN + 63) / 64;
        const float* zf = reinterpret_cast<const float*>(zeros) + lane;
        for (int job = wave; job < 2 * CHUNKS; job += 8) { // wave-uniform
            const int vec = job / CHUNKS, q = job - vec * CHUNKS;
            const int c = q * 64 + lane, n = n0 + c;
            const float* base = vec == 0 ? p.bias : p.bias2;
            const float* g = (base != nullptr && n < p.N) ? base + n : zf;
            if (c < BN) __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(colv + vec * BN + q * 64), 4, 0, 0);
        }
        if (rb_lds) { // BN halves = BN / 2 dwords: one or two wave-instructions, by the waves the loop above leaves idle
            constexpr int RCH = (BN / 2 + 63) / 64;
            const f16* rb = p.row_bias + (size_t)(m0 / p.rows_per_img) * p.ldrb;
            for (int job = wave - 2 * CHUNKS; job >= 0 && job < RCH; job += 8) { // wave-uniform
                const int c = job * 64 + lane, n = n0 + 2 * c;
                const float* g = n < p.N ? reinterpret_cast<const float*>(rb + n) : zf;
                if (c < BN / 2) __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(colv + 2 * BN + job * 64), 4, 0, 0);
            }
        }
        if constexpr (WQ) { // per-column scale of the uint8 encoding, behind the row-bias vector; by the waves counted from the top
            for (int q = 7 - wave; q < CHUNKS; q += 8) {
                const int c = q * 64 + lane, n = n0 + c;
                const float* g = n < p.N ? p.w_scale + n : zf;
                if (c < BN) __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(colv + 2 * BN + BN / 2 + 64 + q * 64), 4, 0, 0);
            }
        }
    }

    f16* sC = smem;
    auto store_phase = [&](int nthr) { // nthr = threads of the workgroup still alive (512; 256 behind an in-kernel split-K fixup)
        constexpr int CPR = BN / 8; // 16-byte chunks per output tile row
        const bool vec_ok = (p.N % 8 == 0) && (p.ldo % 8 == 0) && (p.residual == nullptr || p.ldr % 8 == 0);
        if (vec_ok && p.residual != nullptr && nthr == 512) {
            // every residual piece of the thread requested before the first is used: the loop below fetches one per trip and
            // waits for it (a memory round trip per trip, three or four of them at the tail of every ResBlock's second convolution)
            constexpr int IT = (BM * CPR + 511) / 512;
            f16x8 rr[IT];
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int idx = tid + it * 512;
                const int row = idx / CPR, ch = idx - row * CPR;
                const int m = m0 + row, n = n0 + ch * 8;
                rr[it] = (idx < BM * CPR && m < p.M && n < p.N) ? ldg8(p.residual + (size_t)m * p.ldr + n) : zero8();
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int idx = tid + it * 512;
                const int row = idx / CPR, ch = idx - row * CPR;
                const int m = m0 + row, n = n0 + ch * 8;
                if (idx >= BM * CPR || m >= p.M || n >= p.N) continue;
                f16x8 v = *reinterpret_cast<const f16x8*>(sC + row * SC + ch * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] + (float)rr[it][e]);
                stg8(p.out + (size_t)m * p.ldo + n, v);
            }
            return;
        }
        for (int idx = tid; idx < BM * CPR; idx += nthr) {
            const int row = idx / CPR;
            const int ch = idx - row * CPR;
            const int m = m0 + row, n = n0 + ch * 8;
            if (m >= p.M || n >= p.N) continue;
            f16x8 v = *reinterpret_cast<const f16x8*>(sC + row * SC + ch * 8);
            if (vec_ok) {
                if (p.residual != nullptr) {
                    const f16x8 rr = ldg8(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (f16)((float)v[e] + (float)rr[e]);
                }
                stg8(p.out + (size_t)m * p.ldo + n, v);
            } else {
                for (int e = 0; e < 8; ++e) {
                    if (n + e < p.N) {
                        float f = (float)v[e];
                        if (p.residual != nullptr) f += (float)p.residual[(size_t)m * p.ldr + n + e];
                        p.out[(size_t)m * p.ldo + n + e] = (f16)f;
                    }
                }
            }
        }
    };

    if (!is_consumer) {
        // =============================== LOADER waves ===============================
        // CNT = 8-row weight groups this wave issues per slab (groups lw, lw + 4, ...)
        auto loader = [&](auto cnt_c) {
            constexpr int CNT = decltype(cnt_c)::value;
            constexpr int NB = CNT > 0 ? CNT : 1;
            // weight rows: per-lane byte offset from the wave-uniform slab base (saddr DMA); rows past N are clamped to the last
            // row -- they feed only columns the epilogue never stores
            unsigned b_off[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if constexpr (!WQ) {
                    const int n = min(n0 + (i * NL + lw) * 8 + lrow, p.N - 1);
                    b_off[i] = ((unsigned)n * (unsigned)p.ldw + (unsigned)lchunk * 8u) * 2u;
                } else { // 64-byte slab rows: lanes 0..31 cover the 8 rows of the group, 16-byte pieces swizzled with (row >> 2) & 3 (wq_frag)
                    const int r = (i * NL + lw) * 8 + ((lane & 31) >> 2);
                    const int n = min(n0 + r, p.N - 1);
                    b_off[i] = (unsigned)n * (unsigned)p.ldw + (unsigned)(((lane & 3) ^ ((r >> 2) & 3)) * 16); // ldw in bytes
                }
            }
            const unsigned char* w_s = sgpr_pin(reinterpret_cast<const unsigned char*>(p.w));
            constexpr int WB = WQ ? 1 : 2; // bytes per weight
            if (!tail_wg) {
                constexpr int SLOT = BN * 64;
                // ---- the patch pieces this lane fetches (one 16-byte piece per DMA round): byte offset inside either source,
                // validity bit (zero halo, pixels past the patch, images past the batch: those lanes read the zero line).
                // ALWAYS NRMAX rounds -- rounds past the patch move zeros into spare LDS -- so that every wait count below is a
                // compile-time constant (a skipped DMA would make the counted waits under-wait).
                const int hw = p.h_out * p.w_out;
                const int img0 = fast_div(m0, p.mg_hw, p.sh_hw);
                const int rem0 = m0 - img0 * hw;
                const int y0 = fast_div(rem0, p.mg_w, p.sh_w);
                const int x0 = rem0 - y0 * p.w_out;
                unsigned boff0[NRMAX], boff1[NRMAX], vmask = 0;
#pragma unroll
                for (int j = 0; j < NRMAX; ++j) {
                    const int pix = (j * 256 + lw * 64 + lane) >> 3;
                    const int part = fast_div(pix, p.mg_pp, p.sh_pp);
                    const int r2 = pix - part * p.h_ppix;
                    const int py = fast_div(r2, p.mg_pw, p.sh_pw);
                    const int px = r2 - py * p.h_pw;
                    // patch origin in the source image: one pixel above / left of the tile (upsampling: of its pre-image)
                    const int gy = (p.ups ? (y0 - 1) >> 1 : y0 - 1) + py, gx = (p.ups ? (x0 - 1) >> 1 : x0 - 1) + px, img = img0 + part;
                    const bool ok = pix < p.h_npix && img < p.h_nimg && (unsigned)gy < (unsigned)p.h_in && (unsigned)gx < (unsigned)p.w_in;
                    const unsigned gp = ok ? (unsigned)((img * p.h_in + gy) * p.w_in + gx) : 0u;
                    boff0[j] = (gp * (unsigned)p.sa0 + (unsigned)lchunk * 8u) * 2u;
                    boff1[j] = (gp * (unsigned)p.sa1 + (unsigned)lchunk * 8u) * 2u;
                    vmask |= (ok ? 1u : 0u) << j;
                }
                const int c0_s = sgpr_pin(p.c0), ph_s = sgpr_pin(p.h_patch_halves), cin_s = sgpr_pin(cin);
                const unsigned long long a0_s = sgpr_pin((unsigned long long)p.a0), a1_s = sgpr_pin((unsigned long long)p.a1);
                const unsigned long long z_s = sgpr_pin((unsigned long long)zeros);
                auto issue_patch = [&](int chunk, int buf, auto j_c) { // chunk, buf wave-uniform
                    constexpr int j = decltype(j_c)::value;
                    const int cc = chunk * BK;
                    const bool second = cc >= c0_s;
                    const unsigned long long base = (second ? a1_s : a0_s) + (unsigned long long)((second ? cc - c0_s : cc) * 2);
                    const unsigned long long a = base + (second ? boff1[j] : boff0[j]);
                    const unsigned long long g = ((vmask >> j) & 1u) ? a : z_s;
                    f16* dst = smem + STAGES * SLOT + buf * ph_s + (j * 256 + lw * 64) * 8;
                    __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)dst, 16, 0, 0);
                };
                // The patch of chunk c+1 rides on the weight slabs of chunk c: RPT rounds in front of the slab of tap t, for the
                // taps t >= AHEAD only (slab s is issued after barrier s - AHEAD: from tap AHEAD on, every consumer has left
                // chunk c-1, whose buffer the rounds overwrite).  Everything issued in front of slab 9(c+1) -- all of the patch
                // -- has landed when that slab has (vmcnt retires in order), so the one counted wait per slab covers both.
                constexpr int RPT = (NRMAX + (9 - AHEAD) - 1) / (9 - AHEAD);
                static_assert(AHEAD < 9 && (B_LDX + RPT) * (AHEAD - 1) + RPT < 64, "vmcnt is a 6-bit counter");
                const int chunk_begin = kt_begin / 9, chunk_end = kt_end / 9;
                int k0 = chunk_begin * BK;  // weight column of the next slab to issue: tap * cin + chunk * 64
                int s_slot = 0;             // ... and its ring slot
                static_for<NRMAX>([&](auto j_c) { issue_patch(chunk_begin, 0, j_c); });
                STAMP(1);
                // slabs of chunk c; slab (c, t) is issued after the barrier of the slab AHEAD before it.  LAST: no next patch.
                auto pass = [&](auto last_c, int c, bool first) {
                    constexpr bool LAST = decltype(last_c)::value;
                    static_for<9>([&](auto t_c) {
                        constexpr int t = decltype(t_c)::value;
                        if (t >= AHEAD || !first) {
                            // DMA instructions younger than the slab whose barrier is due (tap ti): AHEAD-1 weight slabs and the
                            // patch rounds in front of those of them that sit at taps >= AHEAD of a chunk with a successor
                            constexpr int ti = (t + 9 - AHEAD) % 9;
                            constexpr bool rounds = !(LAST && t >= AHEAD); // (t < AHEAD: the due slab belongs to chunk c-1)
                            constexpr int young = (AHEAD - 1) * CNT + (rounds ? halo_rounds_between(ti, AHEAD, NRMAX, RPT) : 0);
                            if (dbg & (16 | 32)) wait_vmcnt<0>();
                            else wait_vmcnt<young>();
                            if (!(dbg & 128)) __builtin_amdgcn_s_barrier();
                        }
                        if ((dbg & 32) && !(first && t < AHEAD)) return;
                        if constexpr (!LAST && t >= AHEAD) {
                            if (!(dbg & 16)) {
                                static_for<RPT>([&](auto r_c) {
                                    constexpr int j = RPT * (t - AHEAD) + decltype(r_c)::value;
                                    if constexpr (j < NRMAX) issue_patch(c + 1, (c + 1 - chunk_begin) & 1, std::integral_constant<int, j>{});
                                });
                            }
                        }
                        f16* sB = smem + s_slot * SLOT;
#pragma unroll
                        for (int i = 0; i < CNT; ++i)
                            if (!WQ || lane < 32) lds_dma16_saddr(w_s + (size_t)k0 * WB, b_off[i], sB + (i * NL + lw) * 8 * 64);
                        k0 += t == 8 ? BK - 8 * cin_s : cin_s;
                        s_slot = s_slot + 1 == STAGES ? 0 : s_slot + 1;
                    });
                };
                for (int c = chunk_begin; c + 1 < chunk_end; ++c) pass(std::false_type{}, c, c == chunk_begin);
                pass(std::true_type{}, chunk_end - 1, chunk_begin + 1 == chunk_end);
                static_for<AHEAD>([&](auto t_c) { // the barriers of the last AHEAD slabs: only weight slabs are younger
                    constexpr int t = decltype(t_c)::value;
                    if (dbg & (16 | 32)) wait_vmcnt<0>();
                    else wait_vmcnt<(AHEAD - 1 - t) * CNT>();
                    if (!(dbg & 128)) __builtin_amdgcn_s_barrier();
                });
            }
            if (tail_wg || chain) {
                if (chain) { // every slab and patch of the tap walk has landed and been read: the ring becomes the tail's
                    wait_vmcnt<0>();
                    __syncthreads();
                }
                // 1x1 tail: plain [BM][64] A slabs (the centre pixel of each output row) next to the weight slab
                constexpr int SLOT = (BM + BN) * 64;
                int a_pix[A_LD];
#pragma unroll
                for (int i = 0; i < A_LD; ++i) {
                    const int m = m0 + (i * NL + lw) * 8 + lrow;
                    a_pix[i] = m < p.M ? m : -1;
                }
                auto issue = [&](int j, int slot) {
                    const int kk = j * BK;
                    const bool sec = kk >= p.tc0;
                    const f16* src = sec ? p.t1 : p.t0;
                    const int sa = sec ? p.tc1 : p.tc0;
                    const int ccs = (sec ? kk - p.tc0 : kk) + lchunk * 8;
                    f16* sA = smem + slot * SLOT;
                    f16* sB = sA + BM * 64;
#pragma unroll
                    for (int i = 0; i < A_LD; ++i) {
                        const f16* g = a_pix[i] >= 0 ? src + a_pix[i] * sa + ccs : zeros;
                        __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)(sA + (i * NL + lw) * 8 * 64), 16, 0, 0);
                    }
                    const int k0 = p.k_tail + kk;
#pragma unroll
                    for (int i = 0; i < CNT; ++i) lds_dma16_saddr(w_s + (size_t)k0 * WB, b_off[i], sB + (i * NL + lw) * 8 * 64);
                };
#pragma unroll
                for (int s = 0; s < TAHEAD; ++s)
                    if (s < nkt_t) issue(s, s);
                if (!chain) STAMP(1);
                for (int it = 0; it < nkt_t; ++it) {
                    wait_younger<A_LD + CNT, TST - 2>(max(0, nkt_t - it - 1));
                    __builtin_amdgcn_s_barrier();
                    if (it + TAHEAD < nkt_t) issue(it + TAHEAD, (it + TAHEAD) % TST);
                }
            }
        };
        constexpr int REM = BG % NL; // waves lw < REM carry one group more (REM == 0: all the same)
        if (REM == 0 || lw < REM) loader(std::integral_constant<int, B_LDX>{});
        else loader(std::integral_constant<int, B_LDX - 1>{});
        wait_vmcnt<0>();
        __syncthreads();
        STAMP(2);
        if (p.splits > 1) return;
        __syncthreads(); // the consumers have staged the output tile
        STAMP(3);
        store_phase(NT);
#ifdef SDOD_GEMM_STAMP
        wait_vmcnt<0>();
        STAMP(4);
#endif
        return;
    }

    // =============================== CONSUMER waves ===============================
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int b_off0 = lds_off(wn * WTN + frag_row, frag_chunk); // weight fragment (j = 0, K half 0) inside a slab
    // uint8 weights: z = 1024 - offset of the weight rows this lane reads fragments of (w_off holds offset + 128)
    f16 wq_z[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WTN + j * 16 + frag_row;
        wq_z[j] = WQ ? (f16)(1152.0f - (n < p.N ? p.w_off[n] : 128.0f)) : (f16)0.f;
    }

    if (!tail_wg) {
        constexpr int SLOT = BN * 64;
        // ---- LDS byte address of this lane's A fragment (K half 0) for every tap and fragment row, in patch buffer 0:
        // pixel (row + tap shift) * 128 + swizzled 16-byte piece.  Nine taps unrolled below, so these are plain registers;
        // the other K half is the same address with bit 6 flipped, the other patch buffer a constant further.
        const unsigned smem_base = (unsigned)(uintptr_t)(lds_void_ptr)smem;
        int ty0 = 0, tx0 = 0; // first output pixel of the tile (only the upsampling address map needs it)
        if (p.ups) {
            const int img0 = fast_div(m0, p.mg_hw, p.sh_hw);
            const int rem0 = m0 - img0 * (p.h_out * p.w_out);
            ty0 = fast_div(rem0, p.mg_w, p.sh_w);
            tx0 = rem0 - ty0 * p.w_out;
        }
        unsigned a_addr[9][TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm * WTM + i * 16 + frag_row;
            const int ly = fast_div(r, p.mg_tw, p.sh_tw);
            const int lx = r - ly * p.h_tw;
            const int part = fast_div(ly, p.mg_th, p.sh_th);
            const int lyy = ly - part * p.h_th;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                // patch pixel under tap t of this output pixel; with nearest-2x upsampling folded in, output (y, x) reads source
                // (y >> 1, x >> 1) and the patch origin is the pre-image of the pixel above / left of the tile
                int sy = lyy + t / 3, sx = lx + t % 3;
                if (p.ups) {
                    sy = ((ty0 + lyy + t / 3 - 1) >> 1) - ((ty0 - 1) >> 1);
                    sx = ((tx0 + lx + t % 3 - 1) >> 1) - ((tx0 - 1) >> 1);
                }
                const int pix = part * p.h_ppix + sy * p.h_pw + sx;
                a_addr[t][i] = smem_base + (unsigned)(STAGES * SLOT * 2) + (unsigned)(pix * 128 + ((frag_chunk ^ (pix & 7)) << 4));
            }
        }
        unsigned pdelta = (unsigned)sgpr_pin(p.h_patch_halves) * 2u; // byte distance to the other patch buffer (sign flips per chunk)
        unsigned b_addr0 = smem_base + (unsigned)b_off0 * 2u;  // weight fragment j = 0, K half 0, ring slot 0
        if constexpr (WQ) { // byte image of the slab (wq_frag): 8-row groups at 1 KiB, 64-byte rows, 8 bytes per lane and K half
            const int row = wn * WTN + frag_row;
            b_addr0 = smem_base + (unsigned)((row >> 3) * 1024 + (row & 7) * 64 + (((frag_chunk >> 1) ^ ((row >> 2) & 3)) << 4) + (frag_chunk & 1) * 8);
        }
        STAMP(1);

        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        f16x8 fa[2][TM], fb[2][WQ ? 1 : TN];
        u32x2 fq[2][WQ ? TN : 1]; // uint8 weights: the fragment's 8 codes as read, expanded right before their MFMAs
        auto lds16 = [](unsigned addr) { return *reinterpret_cast<const __attribute__((address_space(3))) f16x8*>((uintptr_t)addr); };
        auto lds8 = [](unsigned addr) { return *reinterpret_cast<const __attribute__((address_space(3))) u32x2*>((uintptr_t)addr); };
        // fragments of (tap t, K half ks) of the slab in ring slot `slot` -> register set b
        auto read_half = [&](auto b_c, auto t_c, auto ks_c, int slot) {
            constexpr int b = decltype(b_c)::value, t = decltype(t_c)::value, ks = decltype(ks_c)::value;
            if (dbg & 256) return;
            const unsigned sb = (b_addr0 + (unsigned)slot * (SLOT * 2)) ^ (ks << (WQ ? 5 : 6));
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[b][i] = lds16(a_addr[t][i] ^ (ks << 6));
            if constexpr (WQ) {
#pragma unroll
                for (int j = 0; j < TN; ++j) fq[b][j] = lds8(sb + j * 2048);
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[b][j] = lds16(sb + j * 16 * 128);
            }
        };
        auto mfma_half = [&](auto b_c) {
            constexpr int b = decltype(b_c)::value;
            if constexpr (WQ) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f16x8 w = u8x8_to_f16(fq[b][j], wq_z[j]);
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] = mfma16(w, fa[b][i], acc[i][j]);
                }
                return;
            }
            if (dbg & 64) { // keep the fragment reads alive without the matrix pipe
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[b][j]));
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[b][i]));
                return;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = mfma16(fb[b][j], fa[b][i], acc[i][j]);
        };

        auto interleave_reads_with_mfmas = [] { // scheduling directive for the region since the last sched_barrier
            if constexpr (WQ) return; // (the expansion's VALU work is left to the scheduler)
            constexpr int NRD = TM + TN, NMF = TM * TN, PAIRS = NRD < NMF ? NRD : NMF;
            __builtin_amdgcn_sched_group_barrier(0x002, TM + 2, 0); // the address arithmetic of the reads first
            static_for<PAIRS>([](auto) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // one LDS read
            });
            if constexpr (NMF > PAIRS) __builtin_amdgcn_sched_group_barrier(0x008, NMF - PAIRS, 0);
            if constexpr (NRD > PAIRS) __builtin_amdgcn_sched_group_barrier(0x100, NRD - PAIRS, 0);
        };

        wait_vmcnt<0>(); // the epilogue vectors this wave requested have landed
        // ... and once more as a BUILTIN, all counters (the waitcnt pass cannot see the asm form): when the loop starts it must
        // know that neither a scalar load nor the epilogue vectors' LDS-DMA is pending, or every LDS wait inside the loop
        // degrades to lgkmcnt(0) (SMEM returns out of order; a pending "flat" LDS access forces full waits) and the
        // fragment double-buffer is lost.  (The consumers issue NO vector-memory instruction inside the loop: one issued
        // behind the loaders' saturated queue cost ~100 ns of the wave's time: an s_memtime timeline of an earlier form)
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_s_barrier(); // slab 0 and the first patch are in LDS (the loaders waited for them)
        int slot = 0;
        const int nchunk = nkt / 9;
        read_half(I0{}, I0{}, I0{}, 0);
        for (int c = 0; c < nchunk; ++c) {
            const bool last_chunk = c + 1 == nchunk;
            static_for<9>([&](auto t_c) {
                constexpr int t = decltype(t_c)::value;
                using TNEXT = std::integral_constant<int, (t + 1) % 9>;
                // K half 1 of this slab is requested while K half 0 is multiplied, READS INTERLEAVED WITH THE MFMAs: a wave issues
                // in order, an MFMA holds its issue port for 8 of its 16 cycles, so a fragment read placed between two MFMAs is
                // free while a block of reads in front of the MFMA block leaves the matrix pipe idle for its whole issue time
                read_half(I1{}, t_c, I1{}, slot);
                mfma_half(I0{});
                interleave_reads_with_mfmas();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): my reads of this slab are done, the ring slot may be refilled
                if (!(t == 8 && last_chunk)) {
                    if (!(dbg & 128)) __builtin_amdgcn_s_barrier();
                    slot = slot + 1 == STAGES ? 0 : slot + 1;
                    if constexpr (t == 8) { // next chunk: its patch sits in the other buffer
#pragma unroll
                        for (int tt = 0; tt < 9; ++tt)
#pragma unroll
                            for (int i = 0; i < TM; ++i) a_addr[tt][i] += pdelta;
                        pdelta = 0u - pdelta;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    read_half(I0{}, TNEXT{}, I0{}, slot);
                }
                mfma_half(I1{});
                interleave_reads_with_mfmas();
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    }
    if (tail_wg || chain) {
        if (chain) {
            __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): my last fragment reads are done
            __syncthreads();
        }
        // 1x1 tail slice: plain ring program
        constexpr int SLOT = (BM + BN) * 64;
        if (!chain) STAMP(1);
        for (int it = 0; it < nkt_t; ++it) {
            __builtin_amdgcn_s_barrier();
            const f16* sA = smem + (it % TST) * SLOT;
            const f16* sB = sA + BM * 64;
            f16x8 xa[2][TM], wb[2][TN];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    xa[ks][i] = *reinterpret_cast<const f16x8*>(sA + lds_off(wm * WTM + i * 16 + frag_row, ks * 4 + frag_chunk));
#pragma unroll
                for (int j = 0; j < TN; ++j) wb[ks][j] = *reinterpret_cast<const f16x8*>(sB + ((b_off0 + j * 16 * 64) ^ (ks << 5)));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j] = mfma16(wb[ks][j], xa[ks][i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    wait_vmcnt<0>();
    __syncthreads(); // all fragment reads done before the epilogue tile overwrites the ring
    STAMP(2);

    if constexpr (WQ) { // acc holds sum_k A (q + offset_n), the integer codes exactly; the column's scale first
        const float* scv = colv + 2 * BN + BN / 2 + 64;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(scv + wn * WTN + j * 16 + e_n);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] *= sc[r];
        }
    }
    int n_store = NT; // threads that take part in the store phase
    if (p.splits > 1 && p.fixup) {
        // ---- split-K WITHOUT a reduce launch: every slice publishes its fp32 tile write-through, the slice that arrives LAST at
        // the tile's counter adds the others to its own accumulators (in slice order, its own in its place: the sum the
        // reduce kernel forms, bit for bit) and runs the fused epilogue.  Cross-XCD hand-off as in gn_grid_kernel: 8-byte
        // agent-scope atomics (= sc1 stores / loads) both sides, every storing wave waits for its stores, one lane signals;
        // the kernel's N is a multiple of 4 (host check).  Only the four consumer waves are still alive here (the loaders
        // returned behind the barrier above), so the barriers below are theirs.
        // 16-byte sc1 accesses by inline asm (the 8-byte agent-scope atomics the compiler offers cost 2.7x per byte on the store
        // side: one fabric write each); the loads are waited for by hand below
        auto pub = [&](float* dst, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory"); };
        auto sub = [&](const float* src) {
            f32x4 v;
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(src) : "memory");
            return v;
        };
        const size_t slab_floats = (size_t)p.M * p.N;
        float* mine = p.partial + (size_t)split * slab_floats;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + e_m;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + e_n;
                if (n < p.N) pub(mine + (size_t)m * p.N + n, acc[i][j]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem_raw); // (the ring is dead: every fragment read is done)
        if (tid == 0) {
            unsigned* cnt = p.fix_counters + (size_t)tile_m * p.tiles_n + tile_n;
            const unsigned prev = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool last = prev + 1u == (unsigned)p.splits;
            if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-armed for the next launch
            *flag = last ? 1 : 0;
        }
        __syncthreads();
        const bool last = *flag != 0;
        __syncthreads(); // (the flag word is overwritten by the output tile below)
        if (!last) return;
        f32x4 sum[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) sum[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s2 = 0; s2 < p.splits; ++s2) {
            const float* slab = p.partial + (size_t)s2 * slab_floats;
            f32x4 t[TM][TN];
            if (s2 != split) { // the whole tile of that slice in flight at once, then one wait
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int m = min(m0 + wm * WTM + i * 16 + e_m, p.M - 1);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int n = min(n0 + wn * WTN + j * 16 + e_n, p.N - 4);
                        t[i][j] = sub(slab + (size_t)m * p.N + n); // (clamped rows / columns are never stored)
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(t[i][j])); // uses stay behind the wait
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) sum[i][j] += s2 != split ? t[i][j] : acc[i][j];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = sum[i][j];
        n_store = 256;
    } else if (p.splits > 1) {
        float* slab = p.partial + (size_t)split * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + e_m;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + e_n;
                if (n + 3 < p.N) {
                    *reinterpret_cast<f32x4*>(slab + (size_t)m * p.N + n) = acc[i][j];
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N) slab[(size_t)m * p.N + n + r] = acc[i][j][r];
                }
            }
        }
        return;
    }

    const float alpha = p.alpha;
    if (alpha != 1.0f) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] *= alpha;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int nl = wn * WTN + j * 16 + e_n;
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(colv + nl), b2 = *reinterpret_cast<const f32x4*>(colv + BN + nl);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = (acc[i][j][r] + b1[r]) + b2[r];
    }
    if (rb_lds) {
        const f16* rbl = reinterpret_cast<const f16*>(colv + 2 * BN);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const f16x4 t = *reinterpret_cast<const f16x4*>(rbl + wn * WTN + j * 16 + e_n);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)t[r];
        }
    } else if (p.row_bias != nullptr) {
        const bool rb_vec = (p.ldrb & 3) == 0 && ((uintptr_t)p.row_bias & 7) == 0 && (p.N & 3) == 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + e_m;
            const f16* rbias = p.row_bias + (size_t)((m < p.M ? m : 0) / p.rows_per_img) * p.ldrb;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + e_n;
                if (rb_vec) {
                    if (n < p.N) {
                        const f16x4 t = *reinterpret_cast<const f16x4*>(rbias + n);
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] += (float)t[r];
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N) acc[i][j][r] += (float)rbias[n + r];
                }
            }
        }
    }
    if (p.act != ACT_NONE) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = apply_act(acc[i][j][r], p.act);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = wm * WTM + i * 16 + e_m;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = wn * WTN + j * 16 + e_n;
            f16x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = (f16)acc[i][j][r];
            *reinterpret_cast<f16x4*>(sC + ml * SC + nl) = h;
        }
    }
    __syncthreads();
    STAMP(3);
    store_phase(n_store);
#ifdef SDOD_GEMM_STAMP
    wait_vmcnt<0>();
    STAMP(4);
#endif
}

// The reduce of every split-K plan in the graphs (whole quads, aligned vectors, bias per column): a workgroup is 64 quads x 4 rows
// (no index divisions -- the general kernel below derives (m, n) from a 64-bit linear index, ~150 instructions in front of its
// first load), every operand and four partial slabs are in flight before the first use, the slabs are added in slice order
// (the bits of the general kernel and of the in-kernel fixup).
__global__ __launch_bounds__(256) void splitk_reduce_vec_kernel(const GemmP p) {
    const int n = ((int)blockIdx.x * 64 + (int)threadIdx.x) * 4;
    const int m = (int)blockIdx.y * 4 + (int)threadIdx.y;
    if (n >= p.N || m >= p.M) return;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float* src = p.partial + (size_t)m * p.N + n;
    const size_t slab = (size_t)p.M * p.N;
    const f32x4 b1 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : zero4;
    const f32x4 b2 = p.bias2 ? *reinterpret_cast<const f32x4*>(p.bias2 + n) : zero4;
    f16x4 rb = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f}, rs = rb;
    if (p.row_bias) rb = *reinterpret_cast<const f16x4*>(p.row_bias + (size_t)fast_div(m, p.mg_rpi, p.sh_rpi) * p.ldrb + n);
    if (p.residual) rs = *reinterpret_cast<const f16x4*>(p.residual + (size_t)m * p.ldr + n);
    f32x4 v = zero4;
    int s = 0;
    for (; s + 4 <= p.splits; s += 4) {
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(src + (size_t)s * slab), t1 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 1) * slab);
        const f32x4 t2 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 2) * slab), t3 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 3) * slab);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (((v[r] + t0[r]) + t1[r]) + t2[r]) + t3[r];
    }
    for (; s < p.splits; ++s) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + (size_t)s * slab);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += t[r];
    }
    f16x4 h;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float f = v[r] * p.alpha;
        if (p.bias) f += b1[r];
        if (p.bias2) f += b2[r];
        if (p.row_bias) f += (float)rb[r];
        f = apply_act(f, p.act);
        f = (float)(f16)f; // same rounding point as the un-split path
        if (p.residual) f += (float)rs[r];
        h[r] = (f16)f;
    }
    *reinterpret_cast<f16x4*>(p.out + (size_t)m * p.ldo + n) = h;
}

// Reduce split-K slabs and apply the fused epilogue.  One thread per 4 consecutive columns.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmP p) {
    const int n4 = (p.N + 3) / 4;
    const size_t total = (size_t)p.M * n4;
    // fast path (every use in the graphs): whole quads, 16-byte aligned vectors -> one load per operand and one 8-byte store
    // per thread, all issued before the first use (the per-element scalar form was a chain of dependent L2 round trips)
    const bool vec = (p.N % 4 == 0) && (p.ldo % 4 == 0) && !p.bias_on_m && (p.residual == nullptr || p.ldr % 4 == 0) &&
                     (p.row_bias == nullptr || p.ldrb % 4 == 0) && (((uintptr_t)p.out | (uintptr_t)p.residual | (uintptr_t)p.row_bias) & 7) == 0 &&
                     (((uintptr_t)p.bias | (uintptr_t)p.bias2) & 15) == 0;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / n4);
        const int n = (int)(idx - (size_t)m * n4) * 4;
        if (vec) {
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 b1 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : zero4;
            const f32x4 b2 = p.bias2 ? *reinterpret_cast<const f32x4*>(p.bias2 + n) : zero4;
            f16x4 rb = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f}, rs = rb;
            if (p.row_bias) rb = *reinterpret_cast<const f16x4*>(p.row_bias + (size_t)(m / p.rows_per_img) * p.ldrb + n);
            if (p.residual) rs = *reinterpret_cast<const f16x4*>(p.residual + (size_t)m * p.ldr + n);
            f32x4 v = zero4;
            for (int s = 0; s < p.splits; ++s) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(p.partial + ((size_t)s * p.M + m) * p.N + n);
                v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
            }
            f16x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float f = v[r] * p.alpha;
                if (p.bias) f += b1[r];
                if (p.bias2) f += b2[r];
                if (p.row_bias) f += (float)rb[r];
                f = apply_act(f, p.act);
                f = (float)(f16)f; // same rounding point as the un-split path
                if (p.residual) f += (float)rs[r];
                h[r] = (f16)f;
            }
            *reinterpret_cast<f16x4*>(p.out + (size_t)m * p.ldo + n) = h;
            continue;
        }
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const bool full = n + 3 < p.N;
        for (int s = 0; s < p.splits; ++s) {
            const float* src = p.partial + ((size_t)s * p.M + m) * p.N + n;
            if (full) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(src);
                v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
            } else {
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.N) v[r] += src[r];
            }
        }
        const f16* rbias = p.row_bias ? p.row_bias + (size_t)(m / p.rows_per_img) * p.ldrb : nullptr;
        for (int r = 0; r < 4; ++r) {
            if (n + r >= p.N) break;
            float f = v[r] * p.alpha;
            if (p.bias) f += p.bias_on_m ? p.bias[m] : p.bias[n + r];
            if (p.bias2) f += p.bias2[n + r];
            if (rbias) f += (float)rbias[n + r];
            f = apply_act(f, p.act);
            f = (float)(f16)f; // same rounding point as the un-split path
            if (p.residual) f += (float)p.residual[(size_t)m * p.ldr + n + r];
            p.out[(size_t)m * p.ldo + n + r] = (f16)f;
        }
    }
}

struct TileCfg {
    int bm, bn;
};
// id 1..5: register-staged kernel; 6..8: LDS-DMA ring kernel (v2), 4 waves; 9..16: v2 with 8 waves (2 per SIMD);
// 17..20: deep rings for the weight-streaming layers (small M, weights from HBM: bytes in flight per CU is what counts)
// 21..22: 160-column tiles: N = 320 / 640 / 1280 divide without the 17 % padding a 128-wide tile pays at N = 320, and
//         M = 8192 x N = 320 becomes exactly 256 workgroups (one per CU)
const TileCfg kTiles[] = {{0, 0},     {128, 128}, {128, 64}, {64, 64},   {256, 16}, {64, 128}, {128, 128},
                          {128, 64},  {64, 64},   {128, 128}, {256, 128}, {128, 64}, {256, 64},
                          {128, 128}, {128, 128}, {256, 128}, {256, 256}, {64, 64},  {128, 64}, {64, 128}, {128, 256},
                          {64, 160},  {32, 160},
                          // 23..31: wave-specialised LDS-DMA kernel (4 consumer + 4 loader waves)
                          {128, 128}, {128, 128}, {128, 256}, {256, 128}, {64, 64}, {64, 64}, {128, 64}, {64, 128}, {64, 160},
                          // 32..36: wave-specialised, two slabs per barrier
                          {64, 64}, {128, 64}, {64, 128}, {64, 160}, {128, 128},
                          // 37..45: halo-patch 3x3 convolution (conv_halo_kernel)
                          {64, 160}, {128, 80}, {128, 160}, {64, 80}, {256, 32}, {128, 32}, {256, 64}, {128, 64}, {64, 64},
                          // 46..48: wave-specialised 32-row tiles for the small-M Linear layers (M = 512 at the 16x16 level: twice the
                          // workgroups of a 64-row tile at three quarters of its bytes per slab)
                          {32, 64}, {32, 128}, {32, 160},
                          // 49..52: halo-patch tiles of 96 / 192 rows, for images whose rows are multiples of 3 (config 5: 96 / 48 / 24 / 12)
                          {96, 160}, {96, 64}, {192, 80}, {192, 64},
                          // 53..55: A-panel kernel (gemm_apanel_kernel): row panel x whole K resident in LDS, n-tiles streamed past it --
                          // K = 320 / 640 / 1280 at 128 / 64 / 32 rows (80 KB panels)
                          {128, 128}, {64, 128}, {32, 128},
                          // 56..58: 160-wide wave-specialised ring tiles for the score GEMM of the folded cross-attention (the row softmax
                          // of its epilogue needs a wave that owns a head's 80 columns): a deep ring for the row-starved levels (M = 512 /
                          // 128: 64 / 16 workgroups, one per CU -- the bytes in flight set the pace), a 128-row tile (M = 8192 x N = 640
                          // is exactly 256 workgroups) and a two-stage 64-row one that leaves room for two workgroups per CU
                          {32, 160}, {128, 160}, {64, 160},
                          // 59..60: ONE head (80 columns) per workgroup, 2 consumer + 2 loader waves: twice the workgroups of the 160-wide
                          // tiles for the row-starved score GEMMs (M = 512: 128 instead of 64) at 58 % of their bytes each
                          {32, 80}, {64, 80},
                          // 61: 128 x 160 with the four consumer waves stacked on the rows (32 x 160 each: an EVEN number of 16-column blocks,
                          // so the GEGLU epilogue can pair value / gate): M = 512 x N = 10240 is exactly 256 workgroups
                          {128, 160}};
constexpr int kNumTiles = 61;
constexpr bool is_ring_tile(int t) { return (t >= 6 && t <= 36) || (t >= 46 && t <= 48) || (t >= 56 && t <= 61); } // gemm_glds_kernel
constexpr bool is_wave80_tile(int t) { return t == 21 || t == 22 || t == 31 || t == 35 || t == 48 || (t >= 56 && t <= 60); } // 80 columns per wave
constexpr int kFirstPanelTile = 53, kLastPanelTile = 55;
constexpr bool is_panel_tile(int t) { return t >= kFirstPanelTile && t <= kLastPanelTile; }
constexpr int kPanelStages = 4;
constexpr int kFirstHaloTile = 37, kLastHaloTile = 45, kFirstHaloTile3 = 49, kLastHaloTile3 = 52;
constexpr bool is_halo_tile(int t) { return (t >= kFirstHaloTile && t <= kLastHaloTile) || (t >= kFirstHaloTile3 && t <= kLastHaloTile3); }
// {STAGES} of the halo tiles (WM x WN is 2x2 for the 160- and 64-wide square-ish ones, 4x1 for the tall ones: launch switch)
const int kHaloStages[] = {4, 4, 3, 4, 6, 6, 4, 4, 4, /* 49.. */ 3, 4, 4, 4};
constexpr int halo_index(int t) { return t <= kLastHaloTile ? t - kFirstHaloTile : t - kFirstHaloTile3 + (kLastHaloTile - kFirstHaloTile + 1); }

const f16* zero_line() { // one per device (the pointer is only valid on the device that allocated it)
    static std::atomic<f16*> z[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    f16* cur = z[dev].load(std::memory_order_acquire);
    if (!cur) {
        f16* fresh = nullptr;
        if (hipMalloc((void**)&fresh, 1024) != hipSuccess) return nullptr;
        (void)hipMemset(fresh, 0, 1024);
        if (z[dev].compare_exchange_strong(cur, fresh, std::memory_order_acq_rel)) cur = fresh;
        else (void)hipFree(fresh); // another thread won the race
    }
    return cur;
}

template <int BM, int BN, int WM, int WN, int STAGES, bool SPEC = false, bool WQ = false, int KSUB = 1>
hipError_t launch_glds(const GemmP& p, dim3 grid, hipStream_t st) {
    constexpr size_t ring = (size_t)STAGES * KSUB * (BM + BN) * 64 * sizeof(f16);
    static_assert(ring + 4 * BN * sizeof(float) <= 160 * 1024, "ring exceeds the 160 KiB of LDS");
    constexpr size_t ctile = (size_t)BM * (BN + 8) * sizeof(f16) + (size_t)BM * 2 * sizeof(float); // + LayerNorm row stats
    constexpr size_t smem = (ring > ctile ? ring : ctile) + (size_t)4 * BN * sizeof(float); // + per-column epilogue vectors
    static std::atomic<unsigned long long> attr_devs{0};
    if (sdod::first_use_on_device(attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_glds_kernel<BM, BN, WM, WN, STAGES, SPEC, WQ, KSUB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    const f16* z = zero_line();
    if (!z) return hipErrorOutOfMemory;
    SDOD_LAUNCH((gemm_glds_kernel<BM, BN, WM, WN, STAGES, SPEC, WQ, KSUB>), grid, dim3(64 * WM * WN * (SPEC ? 2 : 1)), smem, st, p, z);
    return hipGetLastError();
}

template <int BM, int BN, int WM, int WN, int STAGES, bool WQ = false>
hipError_t launch_halo_q(const GemmP& p, dim3 grid, size_t smem, hipStream_t st) {
    static std::atomic<unsigned long long> attr_devs{0};
    if (sdod::first_use_on_device(attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<BM, BN, WM, WN, STAGES, WQ>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    const f16* z = zero_line();
    if (!z) return hipErrorOutOfMemory;
    SDOD_LAUNCH((conv_halo_kernel<BM, BN, WM, WN, STAGES, WQ>), grid, dim3(512), smem, st, p, z);
    return hipGetLastError();
}
template <int BM, int BN, int WM, int WN, int STAGES>
hipError_t launch_halo(const GemmP& p, dim3 grid, size_t smem, hipStream_t st) {
    return p.wq ? launch_halo_q<BM, BN, WM, WN, STAGES, true>(p, grid, smem, st) : launch_halo_q<BM, BN, WM, WN, STAGES, false>(p, grid, smem, st);
}

size_t panel_smem(int bm, int bn, int K) {
    return (size_t)(K / BK) * bm * 128 + (size_t)kPanelStages * bn * 128 + (size_t)6 * bn * sizeof(float) + (size_t)bm * 2 * sizeof(float);
}
// Does A-panel tile `tile` take descriptor d?  (plain row-major fp16 operands, the epilogues of the transformer Linears)
bool panel_ok(const sdod_gemm_desc* d, int tile) {
    if (!is_panel_tile(tile)) return false;
    if (d->a_mode != SDOD_A_ROWS || d->wq || d->k_tail || d->bias_on_m || d->row_bias || d->bias2 || d->split_k > 1) return false;
    if (d->w_img_stride || d->vec_img_stride || d->softmax_cols) return false; // per-image operands / the softmax epilogue live in the ring kernel
    if (d->K % BK || d->K / BK < 3 || d->N % 8 || d->ldo % 8 || ((uintptr_t)d->out & 15) || (d->geglu && d->N % 32)) return false;
    if (d->residual && (d->geglu || d->ldr % 4 || ((uintptr_t)d->residual & 7))) return false;
    if ((unsigned long long)d->M * d->lda * 2 >= (1ull << 32) || (unsigned long long)d->N * d->ldw * 2 >= (1ull << 32)) return false;
    return panel_smem(kTiles[tile].bm, kTiles[tile].bn, d->K) <= 160 * 1024;
}

// n-tiles per workgroup (and groups per panel) of the A-panel kernel: whole waves of 256 workgroups (one per CU) where the
// shape allows, and as many tiles per workgroup as that leaves -- the panel fetch (about 1.5 tiles' worth of time) is paid
// once per workgroup
void panel_grid(int panels, int tiles_n, int* tpg_out, int* groups_out) {
    int best_tpg = 1;
    double best = -1.0;
    for (int tpg = tiles_n; tpg >= 1; --tpg) {
        const int groups = (tiles_n + tpg - 1) / tpg;
        const double wgs = (double)panels * groups;
        const double eff = wgs / (std::ceil(wgs / 256.0) * 256.0);
        const double score = eff * tpg / (tpg + 1.5);
        if (score > best * 1.0001) {
            best = score;
            best_tpg = tpg;
        }
    }
    *tpg_out = best_tpg;
    *groups_out = (tiles_n + best_tpg - 1) / best_tpg;
}

template <int BM, int BN, int WM, int WN>
hipError_t launch_panel(const GemmP& p, dim3 grid, size_t smem, hipStream_t st) {
    static std::atomic<unsigned long long> attr_devs{0};
    if (sdod::first_use_on_device(attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_apanel_kernel<BM, BN, WM, WN, kPanelStages>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    const f16* z = zero_line();
    if (!z) return hipErrorOutOfMemory;
    SDOD_LAUNCH((gemm_apanel_kernel<BM, BN, WM, WN, kPanelStages>), grid, dim3(768), smem, st, p, z);
    return hipGetLastError();
}

template <int BM, int BN, int WM, int WN>
hipError_t launch_cfg(const GemmP& p, dim3 grid, hipStream_t st) {
    constexpr size_t smem = (size_t)2 * (BM + BN) * 64 * sizeof(f16);
    static std::atomic<unsigned long long> attr_devs{0};
    if (sdod::first_use_on_device(attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<BM, BN, WM, WN>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    SDOD_LAUNCH((gemm_kernel<BM, BN, WM, WN>), grid, dim3(256), smem, st, p);
    return hipGetLastError();
}

bool lean_disabled() { // SDOD_GEMM_LEAN_OFF=1: developer switch (A/B timing of the short issue path)
    static const bool off = std::getenv("SDOD_GEMM_LEAN_OFF") != nullptr;
    return off;
}

struct Plan {
    int tile;
    int splits;
    int kt_per_split;
    int main_splits; // halo tiles: splits over the 3x3 taps (a fused 1x1 tail is one more)
    bool chain = false; // halo tiles, unsplit with a fused 1x1 tail: the tail runs behind the tap walk in the same workgroups
};

// Can halo tile `tile` run descriptor d?  Fills the geometry fields of *p (may be null) and the LDS bytes of the launch.
bool halo_geometry(const sdod_gemm_desc* d, int tile, GemmP* p, size_t* smem_bytes) {
    if (!is_halo_tile(tile)) return false;
    if (d->a_mode != SDOD_A_CONV3X3 || d->ksize == 1 || d->stride != 1 || d->geglu || d->ln || d->bias_on_m) return false;
    if (d->wq && d->k_tail) return false;
    if (d->c0 <= 0 || d->c0 % 64 || d->c1 % 64 || d->h_in <= 0 || d->w_in <= 0 || d->n_img <= 0) return false;
    if (d->upsample && d->k_tail) return false;
    const int BMt = kTiles[tile].bm, BNt = kTiles[tile].bn, stages = kHaloStages[halo_index(tile)];
    const int ups = d->upsample ? 1 : 0;
    const int H = d->h_in << ups, W = d->w_in << ups; // OUTPUT size: the tile lives there, the patch in the (smaller) source
    int tw, th, parts;
    if (W >= BMt) {
        if (W % BMt) return false;
        tw = BMt; th = 1; parts = 1;
    } else {
        if (BMt % W) return false;
        tw = W;
        const int rows = BMt / W;
        if (rows <= H) {
            if (H % rows) return false;
            th = rows; parts = 1;
        } else {
            if (rows % H) return false;
            th = H; parts = rows / H;
        }
    }
    // nearest-2x upsampling folded in: the patch is cut from the SOURCE image -- output rows y0-1 .. y0+th map onto source rows
    // (y0-1)>>1 .. (y0+th)>>1, i.e. th/2 + 2 of them (th even, or one output row: 2); same for columns
    if (ups && ((th > 1 && (th & 1)) || (tw & 1))) return false;
    const int pw = (ups ? tw / 2 : tw) + 2, ppix = ((ups ? th / 2 : th) + 2) * pw, npix = parts * ppix;
    const int nr = (npix * 8 + 255) / 256;
    const int nrmax = halo_nrmax(BMt);
    if (nr > nrmax) return false;
    const long long maxc = std::max(std::max(d->c0, d->c1), std::max(d->tc0, d->tc1));
    if ((long long)d->n_img * H * W * maxc >= (1ll << 31)) return false;
    const size_t patch_bytes = (size_t)nrmax * 256 * 16; // the loaders always issue the tile's maximum of DMA rounds
    const size_t halo = (size_t)stages * BNt * 128 + 2 * patch_bytes;
    const size_t tail = d->k_tail ? (size_t)halo_tail_stages(BMt, BNt, stages) * (BMt + BNt) * 128 : 0;
    const size_t ctile = (size_t)BMt * (BNt + 8) * sizeof(f16);
    const size_t body = std::max(std::max(halo, tail), ctile);
    const size_t total = body + (size_t)2 * BNt * sizeof(float) + (size_t)BNt * sizeof(f16) + 256 // + the row-bias vector (whole DMA instructions)
                         + (d->wq ? (size_t)BNt * sizeof(float) + 256 : 0);                        // + the uint8 encoding's scale vector
    if (total > 160 * 1024) return false;
    if (smem_bytes) *smem_bytes = total;
    if (p) {
        p->h_tw = tw; p->h_th = th; p->h_pw = pw; p->h_ppix = ppix; p->h_npix = npix; p->h_nr = nr;
        p->h_nimg = d->n_img;
        p->h_patch_halves = (int)(patch_bytes / sizeof(f16));
        p->h_colv_off = (int)body;
        make_magic((unsigned)tw, &p->mg_tw, &p->sh_tw);
        make_magic((unsigned)th, &p->mg_th, &p->sh_th);
        make_magic((unsigned)pw, &p->mg_pw, &p->sh_pw);
        make_magic((unsigned)ppix, &p->mg_pp, &p->sh_pp);
    }
    return true;
}

// Panels of n-tiles for tile_of(): the xg in {1, 2, 4, 8} that minimises the bytes the eight L2s fetch between them,
// xg * A + (8 / xg) * W (A = the activation operand as it sits in memory -- for a convolution the image, not its im2col; W =
// the weight matrix), with xg <= tiles_n and 8 / xg <= tiles_m where the grid allows (a panel narrower than one tile, or a
// band shorter than one, would spread nothing).  SDOD_GEMM_XCD=<1|2|4|8> forces one value (developer A/B switch);
// sdod_gemm_desc::xcd_panels > 0 overrides both (tune table).
int xcd_panels(const sdod_gemm_desc* d, int tiles_m, int tiles_n) {
    static const int forced = [] {
        const char* e = std::getenv("SDOD_GEMM_XCD");
        const int v = e ? std::atoi(e) : 0;
        return (v == 1 || v == 2 || v == 4 || v == 8) ? v : 0;
    }();
    int want = d->xcd_panels > 0 ? d->xcd_panels : forced;
    if (want != 1 && want != 2 && want != 4 && want != 8) want = 0;
    if (want) return std::min(want, std::max(1, tiles_n));
    const double elem_w = d->wq ? 1.0 : 2.0;
    const double a_bytes = d->a_mode == SDOD_A_ROWS ? (double)d->M * d->K * 2.0
                                                    : (double)d->n_img * d->h_in * d->w_in * (double)(d->c0 + d->c1) * 2.0 +
                                                          (double)d->M * (double)(d->tc0 + d->tc1) * 2.0;
    const double w_bytes = (double)d->N * d->K * elem_w;
    int best = 1;
    double best_cost = 1e300;
    for (int xg = 1; xg <= 8; xg *= 2) {
        if (xg > tiles_n) break;
        const int gm = 8 / xg;
        if (gm > tiles_m && xg != 8) continue; // fewer m-tiles than bands: take more panels instead
        const double cost = xg * a_bytes + gm * w_bytes;
        if (cost < best_cost * 0.999) { // ties keep the smaller xg
            best_cost = cost;
            best = xg;
        }
    }
    // Leave the m-major order unless the saving is worth having (>= 20 % of the bytes): panels also cut every output ROW
    // between XCDs, and at the 64x64 level -- where A is the bigger operand anyway -- two panels bought the convolutions nothing
    // and cost the GroupNorm behind them 0.8 us per launch (profiles/r03_xcd_orders.txt)
    if (best != 1 && tiles_m >= 8 && best_cost > 0.8 * (a_bytes + 8 * w_bytes)) best = 1;
    return best;
}

Plan make_plan(const sdod_gemm_desc* d) {
    Plan pl;
    const int KT = d->K / BK;
    auto ntiles = [&](int t) { return ((d->M + kTiles[t].bm - 1) / kTiles[t].bm) * ((d->N + kTiles[t].bn - 1) / kTiles[t].bn); };
    int tile = d->tile;
    const bool fused = d->geglu || d->k_tail || d->ln || d->wq;
    if (fused && (tile < 6 || tile > kNumTiles)) tile = 14; // fusions live in the LDS-DMA kernel family only
    if (d->geglu && is_wave80_tile(tile)) tile = 14;
    if (d->wq && !(tile == 8 || tile == 13 || tile == 23 || tile == 24 || (tile >= 27 && tile <= 31) || is_halo_tile(tile)))
        tile = 23; // uint8-weight variants
    if (d->wq && d->geglu && tile == 31) tile = 23;  // value/gate pairing needs an even number of 16-column blocks per wave
    // the row softmax of the epilogue lives in one wave: tiles whose waves own 80 columns (the 160-wide ones)
    if (d->softmax_cols && !is_wave80_tile(tile)) tile = 31;
    // per-image weights: an LDS-DMA ring tile whose rows divide the image (a tile must not straddle two weight matrices)
    if (d->w_img_stride && d->rows_per_img > 0 &&
        (!is_ring_tile(tile) || d->rows_per_img % kTiles[tile].bm != 0))
        tile = d->softmax_cols ? (d->rows_per_img % 64 == 0 ? 31 : 48) : (d->rows_per_img % 64 == 0 ? 27 : 46);
    if (tile <= 0 || tile > kNumTiles) {
        if (d->N <= 16) {
            tile = 4;
        } else {
            // largest tile that still gives the 256 CUs at least ~1.5 workgroups each; else the smallest
            const int target = 384;
            if (ntiles(1) >= target) tile = 1;
            else if (ntiles(2) >= target) tile = 2;
            else tile = 3;
        }
    }
    pl.tile = tile;
    int splits = d->split_k;
    if (d->geglu || d->ln || d->softmax_cols || d->vec_img_stride) splits = 1; // the split-K reducer neither pairs value/gate columns nor sees whole rows (nor per-image vectors)
    if (splits <= 0) {
        splits = 1;
        const int nt = ntiles(tile);
        // weight-bandwidth-bound small-M layers: spread K over idle CUs (keep >= 4 slabs of K per split)
        if (nt < 192 && KT >= 8 && d->N % 4 == 0) {
            splits = (512 + nt - 1) / nt;
            if (splits > KT / 4) splits = KT / 4;
            if (splits > 32) splits = 32;
            if (splits < 1) splits = 1;
        }
    }
    pl.main_splits = 0;
    if (is_panel_tile(tile)) { // one workgroup walks the whole K of its tiles (a descriptor it cannot run is rejected at launch)
        pl.splits = 1;
        pl.kt_per_split = KT;
        return pl;
    }
    if (is_halo_tile(tile) && halo_geometry(d, tile, nullptr, nullptr)) { // (a descriptor it cannot run is rejected at launch)
        // whole 64-channel chunks (9 tap slabs) per split; the 1x1 tail, if any, is one more slice
        const int nmain = (d->c0 + d->c1) / BK;
        int want = d->split_k;
        if (want <= 0) {
            const int nt = ntiles(tile);
            want = nt >= 192 ? 1 : (256 + nt / 2) / nt;
        }
        if (want > nmain) want = nmain;
        if (want < 1) want = 1;
        const int cps = (nmain + want - 1) / want;
        pl.main_splits = (nmain + cps - 1) / cps;
        pl.kt_per_split = 9 * cps;
        pl.splits = pl.main_splits + (d->k_tail ? 1 : 0);
        // an EXPLICIT split_k = 1 on a convolution with a fused 1x1 skip: no split at all -- the workgroups chain the tail behind
        // the tap walk (conv_halo_kernel: h_chain).  uint8 weights keep the two-slice form (the tail has its own encoding).
        if (d->k_tail && d->split_k == 1 && !d->wq) {
            pl.main_splits = 1;
            pl.kt_per_split = 9 * nmain;
            pl.splits = 1;
            pl.chain = true;
        }
        return pl;
    }
    if (splits > KT) splits = KT;
    if (splits < 1) splits = 1;
    pl.kt_per_split = (KT + splits - 1) / splits;
    pl.splits = (KT + pl.kt_per_split - 1) / pl.kt_per_split;
    return pl;
}

} // namespace

namespace {
constexpr size_t kFixupCounters = (size_t)64 << 10;
bool fixup_applies(const sdod_gemm_desc* d, const Plan& pl) {
    if (!d->fix_counters || d->phase != 0 || pl.splits <= 1 || !is_halo_tile(pl.tile) || d->N % 4 != 0) return false;
    const size_t tiles = (size_t)((d->M + kTiles[pl.tile].bm - 1) / kTiles[pl.tile].bm) * ((d->N + kTiles[pl.tile].bn - 1) / kTiles[pl.tile].bn);
    return tiles <= kFixupCounters && halo_geometry(d, pl.tile, nullptr, nullptr);
}
} // namespace

extern "C" size_t sdod_gemm_fixup_counters(void) { return kFixupCounters; }

extern "C" int sdod_gemm_fixup(const sdod_gemm_desc* d) {
    if (!d || d->K <= 0 || d->K % BK) return 0;
    return fixup_applies(d, make_plan(d)) ? 1 : 0;
}

extern "C" size_t sdod_gemm_workspace_bytes(const sdod_gemm_desc* d) {
    if (!d || d->K <= 0 || d->K % BK) return 0;
    const Plan pl = make_plan(d);
    return pl.splits > 1 ? (size_t)pl.splits * d->M * d->N * sizeof(float) : 0;
}

extern "C" int sdod_gemm_reduce_info(const sdod_gemm_desc* d, sdod_gn_reduce* out) {
    SDOD_TRY
    SDOD_REQUIRE(d != nullptr && out != nullptr, "null argument");
    SDOD_REQUIRE(d->K > 0 && d->K % BK == 0, "K must be a multiple of 64");
    const Plan pl = make_plan(d);
    SDOD_REQUIRE(pl.splits > 1, "not a split-K plan");
    SDOD_REQUIRE(!d->bias_on_m && !d->geglu && !d->ln && d->ldo == d->N && d->N % 8 == 0, "epilogue cannot move into a GroupNorm");
    SDOD_REQUIRE(d->workspace != nullptr && d->workspace_bytes >= (size_t)pl.splits * d->M * d->N * sizeof(float), "split-K workspace missing");
    out->partial = d->workspace;
    out->splits = pl.splits;
    out->slab_floats = (size_t)d->M * d->N;
    out->bias = d->bias;
    out->bias2 = d->bias2;
    out->row_bias = d->row_bias;
    out->ld_row_bias = d->ld_row_bias > 0 ? d->ld_row_bias : d->N;
    out->residual = d->residual;
    out->ldr = d->ldr;
    out->x_out = d->out;
    out->alpha = d->alpha;
    out->act = d->act;
    out->M = d->M;
    out->N = d->N;
    return 0;
    SDOD_CATCH
}

#ifdef SDOD_GEMM_STAMP
extern "C" __attribute__((visibility("default"))) int sdod_gemm_stamps(unsigned long long* out, int n_wg) {
    if (!out || n_wg <= 0 || n_wg > 8192) return 1;
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), (size_t)n_wg * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 3;
}
#endif


extern "C" int sdod_gemm_num_tiles(void) { return kNumTiles; }

// template arguments of tile `tile` as the launch switch in sdod_gemm_f16 instantiates it: {BM, BN, WM, WN, STAGES (0 = the
// register-staged gemm_kernel), SPEC, KSUB}; tools and bench.py build the kernel symbol a profiler prints from these
extern "C" int sdod_gemm_tile_info(int tile, int out[7]) {
    static const int kInfo[][7] = {
        {0, 0, 0, 0, 0, 0, 0},
        {128, 128, 2, 2, 0, 0, 1}, {128, 64, 2, 2, 0, 0, 1}, {64, 64, 2, 2, 0, 0, 1}, {256, 16, 4, 1, 0, 0, 1}, {64, 128, 2, 2, 0, 0, 1},
        {128, 128, 2, 2, 3, 0, 1}, {128, 64, 2, 2, 4, 0, 1}, {64, 64, 2, 2, 4, 0, 1}, {128, 128, 2, 4, 3, 0, 1}, {256, 128, 4, 2, 2, 0, 1},
        {128, 64, 4, 2, 4, 0, 1}, {256, 64, 4, 2, 3, 0, 1}, {128, 128, 2, 4, 4, 0, 1}, {128, 128, 2, 4, 2, 0, 1}, {256, 128, 4, 2, 3, 0, 1},
        {256, 256, 2, 4, 2, 0, 1}, {64, 64, 2, 2, 8, 0, 1}, {128, 64, 2, 2, 6, 0, 1}, {64, 128, 2, 2, 6, 0, 1}, {128, 256, 2, 4, 3, 0, 1},
        {64, 160, 2, 2, 4, 0, 1}, {32, 160, 2, 2, 6, 0, 1},
        {128, 128, 2, 2, 4, 1, 1}, {128, 128, 2, 2, 3, 1, 1}, {128, 256, 2, 2, 3, 1, 1}, {256, 128, 2, 2, 3, 1, 1}, {64, 64, 2, 2, 8, 1, 1},
        {64, 64, 2, 2, 4, 1, 1}, {128, 64, 2, 2, 6, 1, 1}, {64, 128, 2, 2, 6, 1, 1}, {64, 160, 2, 2, 4, 1, 1},
        {64, 64, 2, 2, 4, 1, 2}, {128, 64, 2, 2, 3, 1, 2}, {64, 128, 2, 2, 3, 1, 2}, {64, 160, 2, 2, 2, 1, 2}, {128, 128, 2, 2, 2, 1, 2},
        // SPEC column 2 = conv_halo_kernel<BM, BN, WM, WN, STAGES>
        {64, 160, 2, 2, 4, 2, 1}, {128, 80, 4, 1, 4, 2, 1}, {128, 160, 2, 2, 3, 2, 1}, {64, 80, 4, 1, 4, 2, 1}, {256, 32, 4, 1, 6, 2, 1},
        {128, 32, 4, 1, 6, 2, 1}, {256, 64, 4, 1, 4, 2, 1}, {128, 64, 2, 2, 4, 2, 1}, {64, 64, 2, 2, 4, 2, 1},
        {32, 64, 2, 2, 6, 1, 1}, {32, 128, 1, 4, 6, 1, 1}, {32, 160, 2, 2, 4, 1, 1},
        {96, 160, 2, 2, 3, 2, 1}, {96, 64, 2, 2, 4, 2, 1}, {192, 80, 4, 1, 4, 2, 1}, {192, 64, 4, 1, 4, 2, 1},
        // SPEC column 3 = gemm_apanel_kernel<BM, BN, WM, WN, STAGES>
        {128, 128, 2, 2, kPanelStages, 3, 1}, {64, 128, 2, 2, kPanelStages, 3, 1}, {32, 128, 2, 2, kPanelStages, 3, 1},
        {32, 160, 2, 2, 6, 1, 1}, {128, 160, 2, 2, 3, 1, 1}, {64, 160, 2, 2, 2, 1, 1}, {32, 80, 2, 1, 5, 1, 1}, {64, 80, 2, 1, 5, 1, 1}, {128, 160, 4, 1, 3, 1, 1}};
    static_assert(sizeof(kInfo) / sizeof(kInfo[0]) == kNumTiles + 1, "one row per tile");
    if (tile < 1 || tile > kNumTiles || !out) return sdod::INVALID_ARGUMENT;
    for (int i = 0; i < 7; ++i) out[i] = kInfo[tile][i];
    return 0;
}

extern "C" int sdod_gemm_tile_shape(int tile, int* bm, int* bn, int* lds_dma) {
    if (tile < 1 || tile > kNumTiles) return sdod::INVALID_ARGUMENT;
    if (bm) *bm = kTiles[tile].bm;
    if (bn) *bn = kTiles[tile].bn;
    if (lds_dma) *lds_dma = tile >= 6 ? 1 : 0;
    return 0;
}

extern "C" int sdod_gemm_panel_ok(const sdod_gemm_desc* d, int tile) {
    if (!d || d->K <= 0 || tile < 1 || tile > kNumTiles) return 0;
    return panel_ok(d, tile) ? 1 : 0;
}

extern "C" int sdod_gemm_halo_ok(const sdod_gemm_desc* d, int tile) {
    if (!d || d->K <= 0 || d->K % BK || tile < 1 || tile > kNumTiles) return 0;
    return halo_geometry(d, tile, nullptr, nullptr) ? 1 : 0;
}

extern "C" int sdod_gemm_xcd_panels(const sdod_gemm_desc* d) {
    if (!d || d->K <= 0 || d->K % BK || d->M <= 0 || d->N <= 0) return 0;
    const Plan pl = make_plan(d);
    const TileCfg tc = kTiles[pl.tile];
    return xcd_panels(d, (d->M + tc.bm - 1) / tc.bm, (d->N + tc.bn - 1) / tc.bn);
}

extern "C" int sdod_gemm_plan(const sdod_gemm_desc* d, int* tile, int* splits) {
    if (!d || d->K <= 0 || d->K % BK) return sdod::INVALID_ARGUMENT;
    const Plan pl = make_plan(d);
    if (tile) *tile = pl.tile;
    if (splits) *splits = pl.splits;
    return 0;
}

extern "C" int sdod_gemm_f16(const sdod_gemm_desc* d, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(d != nullptr, "null descriptor");
    SDOD_REQUIRE(d->a && d->w && d->out, "null operand pointer");
    SDOD_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "M, N, K must be positive");
    SDOD_REQUIRE(d->K % BK == 0, "K must be a multiple of 64 (pad the weight / use im2col for Cin<64)");
    SDOD_REQUIRE(d->ldw >= d->K && d->ldw % (d->wq ? 16 : 8) == 0, "ldw must be >= K and keep rows 16-byte aligned");
    SDOD_REQUIRE(d->ldo >= (d->geglu ? d->N / 2 : d->N), "ldo must be >= N");
    SDOD_REQUIRE(((uintptr_t)d->a & 15) == 0 && ((uintptr_t)d->w & 15) == 0, "operands must be 16-byte aligned");
    GemmP p{};
    p.a0 = (const f16*)d->a;
    p.a1 = (const f16*)d->a2;
    p.w = (const f16*)d->w;
    p.bias = (const float*)d->bias;
    p.row_bias = (const f16*)d->row_bias;
    p.ldrb = d->ld_row_bias > 0 ? d->ld_row_bias : d->N;
    p.residual = (const f16*)d->residual;
    p.out = (f16*)d->out;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.lda = d->lda; p.ldw = d->ldw; p.ldo = d->ldo; p.ldr = d->ldr;
    p.mode = d->a_mode;
    p.act = d->act;
    p.alpha = d->alpha;
    p.bias_on_m = d->bias_on_m;
    p.rows_per_img = d->rows_per_img > 0 ? d->rows_per_img : 1;
    p.geglu = d->geglu ? 1 : 0;
    p.k_tail = d->k_tail;
    p.t0 = (const f16*)d->t0; p.t1 = (const f16*)d->t1; p.tc0 = d->tc0; p.tc1 = d->tc1;
    p.bias2 = (const float*)d->bias2;
    p.ln = d->ln ? 1 : 0;
    p.ln_s = (const float*)d->ln_s;
    p.ln_eps = d->ln_eps;
    if (d->ln) SDOD_REQUIRE(d->a_mode == SDOD_A_ROWS && d->ln_s != nullptr && !d->bias_on_m, "ln fold needs rows mode and ln_s");
    p.wq = d->wq ? 1 : 0;
    p.w_scale = (const float*)d->w_scale;
    p.w_off = (const float*)d->w_off;
    if (d->wq) SDOD_REQUIRE(d->w_scale && d->w_off && !d->ln && !d->k_tail && !d->bias_on_m, "uint8 weights need w_scale / w_off and exclude ln, k_tail, bias_on_m");
    if (d->geglu) {
        SDOD_REQUIRE(d->N % 32 == 0 && !d->residual && !d->row_bias && !d->bias_on_m && d->act == 0, "geglu: N % 32 == 0, no residual/row_bias/act");
        SDOD_REQUIRE(d->ldo >= d->N / 2, "geglu: ldo must be >= N/2");
    }
    if (d->k_tail) {
        SDOD_REQUIRE(d->a_mode == SDOD_A_CONV3X3 && d->stride == 1 && !d->upsample, "tail segment needs a stride-1 conv");
        SDOD_REQUIRE(d->t0 && d->tc0 > 0 && d->tc0 % 64 == 0 && d->tc1 % 64 == 0 && (d->t1 || d->tc1 == 0), "bad tail sources");
        SDOD_REQUIRE(d->K == d->k_tail + d->tc0 + d->tc1, "K must equal k_tail + tc0 + tc1");
    }
    if (d->residual) SDOD_REQUIRE(d->ldr >= d->N, "ldr must be >= N");
    if (d->row_bias) SDOD_REQUIRE(d->rows_per_img > 0 && !d->bias_on_m, "row_bias needs rows_per_img");
    if (d->a_mode == SDOD_A_CONV3X3) {
        SDOD_REQUIRE(d->stride == 1 || d->stride == 2, "conv stride must be 1 or 2");
        SDOD_REQUIRE(d->c0 > 0 && d->c0 % 64 == 0 && d->c1 % 64 == 0, "conv channels must be multiples of 64");
        SDOD_REQUIRE(d->a2 != nullptr || d->c1 == 0, "c1 > 0 needs a2");
        const int ks = d->ksize == 1 ? 1 : 3;
        SDOD_REQUIRE(d->ksize == 0 || d->ksize == 1 || d->ksize == 3, "conv kernel size must be 1 or 3");
        SDOD_REQUIRE((d->k_tail ? d->k_tail : d->K) == ks * ks * (d->c0 + d->c1), "conv K must equal ksize^2*(c0+c1)");
        p.ksize = ks;
        SDOD_REQUIRE(!(d->upsample && d->stride != 1), "upsample implies stride 1");
        p.h_in = d->h_in; p.w_in = d->w_in; p.c0 = d->c0; p.c1 = d->c1;
        p.stride = d->stride; p.ups = d->upsample ? 1 : 0;
        p.sa0 = d->c0; p.sa1 = d->c1;
        const int hup = d->h_in << p.ups, wup = d->w_in << p.ups;
        p.h_out = (hup + 2 * (ks / 2) - ks) / d->stride + 1;
        p.w_out = (wup + 2 * (ks / 2) - ks) / d->stride + 1;
        SDOD_REQUIRE(d->M == d->n_img * p.h_out * p.w_out, "conv M must equal n_img*h_out*w_out");
    } else {
        SDOD_REQUIRE(d->a_mode == SDOD_A_ROWS, "unknown a_mode");
        SDOD_REQUIRE(d->lda >= d->K && d->lda % 8 == 0, "lda must be >= K and a multiple of 8");
        p.c0 = d->K; p.c1 = 0; p.h_in = p.w_in = p.h_out = p.w_out = 1; p.stride = 1; p.ksize = 1;
        p.sa0 = d->lda; p.sa1 = 0;
    }
    p.w_img_stride = d->w_img_stride;
    p.vec_img_stride = d->vec_img_stride;
    p.softmax_g = d->softmax_cols;
    make_magic((unsigned)p.rows_per_img, &p.mg_rpi, &p.sh_rpi);
    if (d->w_img_stride || d->vec_img_stride)
        SDOD_REQUIRE(d->w_img_stride > 0 && d->vec_img_stride >= 0 && d->a_mode == SDOD_A_ROWS && d->rows_per_img > 0 && d->rows_per_img % 32 == 0 &&
                         d->M % d->rows_per_img == 0 && !d->wq && !d->bias_on_m && !d->k_tail && !d->bias2 && d->w_img_stride % 8 == 0,
                     "per-image weights: rows mode, rows_per_img a multiple of 32 that divides M, fp16 weights, no bias_on_m / bias2 / tail");
    if (d->softmax_cols)
        SDOD_REQUIRE(d->softmax_cols == 80 && d->N % 160 == 0 && !d->geglu && !d->wq && !d->residual && !d->row_bias && d->act == 0 && !d->bias_on_m,
                     "softmax_cols: 80-column groups (N a multiple of 160), no geglu / uint8 weights / residual / row_bias / act");
    p.lean = (d->a_mode == SDOD_A_ROWS && !d->k_tail && (unsigned long long)d->M * d->lda * 2 < (1ull << 32) &&
              (unsigned long long)d->N * d->ldw * 2 < (1ull << 32) && !lean_disabled()) ? 1 : 0;
    {
        static const bool respf_off = [] { const char* e2 = std::getenv("SDOD_GEMM_RESPF"); return e2 && e2[0] == '0'; }();
        p.no_respf = respf_off ? 1 : 0;
    }
    const Plan pl = make_plan(d);
    SDOD_REQUIRE(!(d->geglu && is_wave80_tile(pl.tile)), "geglu needs a tile with an even number of 16-column blocks per wave");
    SDOD_REQUIRE(!(d->softmax_cols || d->w_img_stride) || is_ring_tile(pl.tile), "softmax_cols / per-image weights need an LDS-DMA ring tile");
    SDOD_REQUIRE(!(d->geglu || d->k_tail || d->bias2 || d->ln || d->wq) || pl.tile >= 6, "geglu / tail segment / bias2 / ln / uint8 weights need an LDS-DMA tile (6..48)");
    p.splits = pl.splits;
    p.kt_per_split = pl.kt_per_split;
    size_t halo_smem = 0;
    if (is_halo_tile(pl.tile)) {
        SDOD_REQUIRE(halo_geometry(d, pl.tile, &p, &halo_smem), "this halo-patch tile does not take the convolution (3x3, stride 1, tile rows must divide the image)");
        p.h_main_splits = pl.main_splits;
        p.h_chain = pl.chain ? 1 : 0;
        if (fixup_applies(d, pl)) {
            p.fixup = 1;
            p.fix_counters = (unsigned*)d->fix_counters;
        }
    }
    if (pl.splits > 1) {
        SDOD_REQUIRE(d->workspace != nullptr && d->workspace_bytes >= (size_t)pl.splits * d->M * d->N * sizeof(float),
                     "split-K workspace missing or too small");
        p.partial = (float*)d->workspace;
    }
    const TileCfg tc = kTiles[pl.tile];
    p.tiles_m = (d->M + tc.bm - 1) / tc.bm;
    p.tiles_n = (d->N + tc.bn - 1) / tc.bn;
    make_magic((unsigned)(p.h_out * p.w_out), &p.mg_hw, &p.sh_hw);
    make_magic((unsigned)p.w_out, &p.mg_w, &p.sh_w);
    make_magic((unsigned)p.tiles_n, &p.mg_tn, &p.sh_tn);
    {
        const int xg = xcd_panels(d, p.tiles_m, p.tiles_n);
        const int w = p.tiles_n / xg, r = p.tiles_n % xg;
        p.xg_w = w;
        p.xg_s1 = p.tiles_m * (w + 1);
        p.xg_s0 = p.tiles_m * w;
        p.xg_big = r * p.xg_s1;
        p.xg_nbig = r * (w + 1);
        make_magic((unsigned)p.xg_s1, &p.mg_s1, &p.sh_s1);
        make_magic((unsigned)p.xg_s0, &p.mg_s0, &p.sh_s0);
        make_magic((unsigned)(w + 1), &p.mg_w1, &p.sh_w1);
        make_magic((unsigned)w, &p.mg_w0, &p.sh_w0);
    }
    make_magic((unsigned)(p.c0 + p.c1), &p.mg_cin, &p.sh_cin);
    dim3 grid(p.tiles_m * p.tiles_n, 1, pl.splits);
    hipStream_t st = (hipStream_t)stream;
    SDOD_REQUIRE(d->phase >= 0 && d->phase <= 2 && (d->phase == 0 || pl.splits > 1), "phase 1/2 only apply to a split-K plan");
    hipError_t e = hipSuccess;
    if (is_panel_tile(pl.tile)) {
        SDOD_REQUIRE(panel_ok(d, pl.tile), "this A-panel tile does not take the GEMM (plain rows x fp16 weights, K >= 192, the row panel must fit LDS)");
        // grid: one workgroup per (row panel, group of consecutive n-tiles), about one per CU
        p.ap_panels = p.tiles_m;
        panel_grid(p.ap_panels, p.tiles_n, &p.ap_tpg, &p.ap_groups);
        p.ap_nmajor = (double)d->N * d->K > (double)d->M * d->K ? 1 : 0; // W the bigger operand: an XCD keeps a group's W, not a panel's A
        const size_t smem = panel_smem(tc.bm, tc.bn, d->K);
        const dim3 pgrid(p.ap_panels * p.ap_groups);
        switch (pl.tile) {
        case 53: e = launch_panel<128, 128, 2, 2>(p, pgrid, smem, st); break;
        case 54: e = launch_panel<64, 128, 2, 2>(p, pgrid, smem, st); break;
        default: e = launch_panel<32, 128, 2, 2>(p, pgrid, smem, st); break;
        }
        SDOD_HIP_CHECK(e);
        return 0;
    }
    const bool halo_tile = is_halo_tile(pl.tile); // (takes uint8 weights itself: launch_halo)
    if (d->phase != 2 && d->wq && !halo_tile)
    switch (pl.tile) { // the uint8-weight variants (make_plan maps every other tile onto one of these)
    case 8: e = launch_glds<64, 64, 2, 2, 4, false, true>(p, grid, st); break;
    case 13: e = launch_glds<128, 128, 2, 4, 4, false, true>(p, grid, st); break;
    case 24: e = launch_glds<128, 128, 2, 2, 3, true, true>(p, grid, st); break;
    case 27: e = launch_glds<64, 64, 2, 2, 8, true, true>(p, grid, st); break;
    case 28: e = launch_glds<64, 64, 2, 2, 4, true, true>(p, grid, st); break;
    case 29: e = launch_glds<128, 64, 2, 2, 6, true, true>(p, grid, st); break;
    case 30: e = launch_glds<64, 128, 2, 2, 6, true, true>(p, grid, st); break;
    case 31: e = launch_glds<64, 160, 2, 2, 4, true, true>(p, grid, st); break;
    default: e = launch_glds<128, 128, 2, 2, 4, true, true>(p, grid, st); break; // 23
    }
    else if (d->phase != 2)
    switch (pl.tile) {
    case 37: e = launch_halo<64, 160, 2, 2, 4>(p, grid, halo_smem, st); break;
    case 38: e = launch_halo<128, 80, 4, 1, 4>(p, grid, halo_smem, st); break;
    case 39: e = launch_halo<128, 160, 2, 2, 3>(p, grid, halo_smem, st); break;
    case 40: e = launch_halo<64, 80, 4, 1, 4>(p, grid, halo_smem, st); break;
    case 41: e = launch_halo<256, 32, 4, 1, 6>(p, grid, halo_smem, st); break;
    case 42: e = launch_halo<128, 32, 4, 1, 6>(p, grid, halo_smem, st); break;
    case 43: e = launch_halo<256, 64, 4, 1, 4>(p, grid, halo_smem, st); break;
    case 44: e = launch_halo<128, 64, 2, 2, 4>(p, grid, halo_smem, st); break;
    case 45: e = launch_halo<64, 64, 2, 2, 4>(p, grid, halo_smem, st); break;
    case 49: e = launch_halo<96, 160, 2, 2, 3>(p, grid, halo_smem, st); break;
    case 50: e = launch_halo<96, 64, 2, 2, 4>(p, grid, halo_smem, st); break;
    case 51: e = launch_halo<192, 80, 4, 1, 4>(p, grid, halo_smem, st); break;
    case 52: e = launch_halo<192, 64, 4, 1, 4>(p, grid, halo_smem, st); break;
    case 1: e = launch_cfg<128, 128, 2, 2>(p, grid, st); break;
    case 2: e = launch_cfg<128, 64, 2, 2>(p, grid, st); break;
    case 3: e = launch_cfg<64, 64, 2, 2>(p, grid, st); break;
    case 4: e = launch_cfg<256, 16, 4, 1>(p, grid, st); break;
    case 5: e = launch_cfg<64, 128, 2, 2>(p, grid, st); break;
    case 6: e = launch_glds<128, 128, 2, 2, 3>(p, grid, st); break;
    case 7: e = launch_glds<128, 64, 2, 2, 4>(p, grid, st); break;
    case 8: e = launch_glds<64, 64, 2, 2, 4>(p, grid, st); break;
    case 9: e = launch_glds<128, 128, 2, 4, 3>(p, grid, st); break;
    case 10: e = launch_glds<256, 128, 4, 2, 2>(p, grid, st); break;
    case 11: e = launch_glds<128, 64, 4, 2, 4>(p, grid, st); break;
    case 12: e = launch_glds<256, 64, 4, 2, 3>(p, grid, st); break;
    case 13: e = launch_glds<128, 128, 2, 4, 4>(p, grid, st); break;
    case 14: e = launch_glds<128, 128, 2, 4, 2>(p, grid, st); break;
    case 15: e = launch_glds<256, 128, 4, 2, 3>(p, grid, st); break;
    case 16: e = launch_glds<256, 256, 2, 4, 2>(p, grid, st); break;
    case 17: e = launch_glds<64, 64, 2, 2, 8>(p, grid, st); break;
    case 18: e = launch_glds<128, 64, 2, 2, 6>(p, grid, st); break;
    case 19: e = launch_glds<64, 128, 2, 2, 6>(p, grid, st); break;
    case 20: e = launch_glds<128, 256, 2, 4, 3>(p, grid, st); break;
    case 21: e = launch_glds<64, 160, 2, 2, 4>(p, grid, st); break;
    case 22: e = launch_glds<32, 160, 2, 2, 6>(p, grid, st); break;
    case 23: e = launch_glds<128, 128, 2, 2, 4, true>(p, grid, st); break;
    case 24: e = launch_glds<128, 128, 2, 2, 3, true>(p, grid, st); break;
    case 25: e = launch_glds<128, 256, 2, 2, 3, true>(p, grid, st); break;
    case 26: e = launch_glds<256, 128, 2, 2, 3, true>(p, grid, st); break;
    case 27: e = launch_glds<64, 64, 2, 2, 8, true>(p, grid, st); break;
    case 28: e = launch_glds<64, 64, 2, 2, 4, true>(p, grid, st); break;
    case 29: e = launch_glds<128, 64, 2, 2, 6, true>(p, grid, st); break;
    case 30: e = launch_glds<64, 128, 2, 2, 6, true>(p, grid, st); break;
    case 31: e = launch_glds<64, 160, 2, 2, 4, true>(p, grid, st); break;
    case 32: e = launch_glds<64, 64, 2, 2, 4, true, false, 2>(p, grid, st); break;
    case 33: e = launch_glds<128, 64, 2, 2, 3, true, false, 2>(p, grid, st); break;
    case 34: e = launch_glds<64, 128, 2, 2, 3, true, false, 2>(p, grid, st); break;
    case 35: e = launch_glds<64, 160, 2, 2, 2, true, false, 2>(p, grid, st); break;
    case 46: e = launch_glds<32, 64, 2, 2, 6, true>(p, grid, st); break;
    case 47: e = launch_glds<32, 128, 1, 4, 6, true>(p, grid, st); break;
    case 48: e = launch_glds<32, 160, 2, 2, 4, true>(p, grid, st); break;
    case 56: e = launch_glds<32, 160, 2, 2, 6, true>(p, grid, st); break;
    case 57: e = launch_glds<128, 160, 2, 2, 3, true>(p, grid, st); break;
    case 58: e = launch_glds<64, 160, 2, 2, 2, true>(p, grid, st); break;
    case 59: e = launch_glds<32, 80, 2, 1, 5, true>(p, grid, st); break;
    case 60: e = launch_glds<64, 80, 2, 1, 5, true>(p, grid, st); break;
    case 61: e = launch_glds<128, 160, 4, 1, 3, true>(p, grid, st); break;
    default: e = launch_glds<128, 128, 2, 2, 2, true, false, 2>(p, grid, st); break;
    }
    SDOD_HIP_CHECK(e);
    if (pl.splits > 1 && d->phase != 1 && !p.fixup) {
        const bool vec = (p.N % 4 == 0) && (p.ldo % 4 == 0) && !p.bias_on_m && (p.residual == nullptr || p.ldr % 4 == 0) &&
                         (p.row_bias == nullptr || p.ldrb % 4 == 0) && (((uintptr_t)p.out | (uintptr_t)p.residual | (uintptr_t)p.row_bias) & 7) == 0 &&
                         (((uintptr_t)p.bias | (uintptr_t)p.bias2 | (uintptr_t)p.partial) & 15) == 0 && (d->M + 3) / 4 <= 65535;
        if (vec) {
            SDOD_LAUNCH(splitk_reduce_vec_kernel, dim3((d->N / 4 + 63) / 64, (d->M + 3) / 4), dim3(64, 4), 0, st, p);
        } else {
            const size_t total = (size_t)d->M * ((d->N + 3) / 4);
            int blocks = (int)((total + 255) / 256);
            if (blocks > 2048) blocks = 2048;
            SDOD_LAUNCH(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
        }
        SDOD_HIP_CHECK(hipGetLastError());
    }
    return 0;
    SDOD_CATCH
}

// developer aid: average duration (ms) of `iters` back-to-back launches, timed with HIP events on `stream`
// (the Python launch path costs ~15 us per call, which hides kernels shorter than that)
extern "C" int sdod_gemm_time(const sdod_gemm_desc* d, void* stream, int iters, float* ms_avg) {
    SDOD_TRY
    SDOD_REQUIRE(d && ms_avg && iters > 0, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < 3; ++i) {
        const int rc = sdod_gemm_f16(d, stream);
        if (rc) return rc;
    }
    hipEvent_t e0, e1;
    SDOD_HIP_CHECK(hipEventCreate(&e0));
    SDOD_HIP_CHECK(hipEventCreate(&e1));
    SDOD_HIP_CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) {
        const int rc = sdod_gemm_f16(d, stream);
        if (rc) return rc;
    }
    SDOD_HIP_CHECK(hipEventRecord(e1, st));
    SDOD_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    SDOD_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *ms_avg = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 0;
    SDOD_CATCH
}

// ---- cold-cache timing.  Inside a graph replay every GEMM finds its WEIGHTS in HBM (the UNet's 1.7 GB of fp16 weights
// sweep the 256 MiB Infinity Cache many times per evaluation) while its activations were written by the previous launch
// and are still in L2 / Infinity Cache.  Back-to-back launches of one descriptor measure the opposite (weights hot), which
// mis-ranks tile shapes for the weight-streaming layers.  So: sweep a scratch buffer larger than the caches (read + write),
// re-touch the activation operands, then time ONE launch; repeat.
namespace {
__global__ __launch_bounds__(256) void cache_sweep_kernel(float4* buf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = buf[i];
        v.x += 1.0f;
        buf[i] = v;
    }
}
__global__ __launch_bounds__(256) void touch_kernel(const float4* a, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float4 v = a[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345e38f) *sink = acc; // never true; keeps the loads
}
void touch(const void* ptr, size_t bytes, float* sink, hipStream_t st) {
    if (!ptr || bytes < 16) return;
    const size_t n = bytes / 16;
    int blocks = (int)std::min<size_t>((n + 255) / 256, 1024);
    SDOD_LAUNCH(touch_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)ptr, n, sink);
}
} // namespace

extern "C" int sdod_gemm_time_cold(const sdod_gemm_desc* d, void* stream, int iters, void* scratch, size_t scratch_bytes,
                                   float* ms_avg, float* ms_min) {
    SDOD_TRY
    SDOD_REQUIRE(d && ms_avg && iters > 0 && iters <= 16, "bad argument");
    SDOD_REQUIRE(scratch && scratch_bytes >= ((size_t)64 << 20) && ((uintptr_t)scratch & 15) == 0, "scratch must be >= 64 MiB, 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    int rc = sdod_gemm_f16(d, stream); // code load, LDS attribute, argument checks
    if (rc) return rc;
    hipEvent_t ev[32];
    for (int i = 0; i < 2 * iters; ++i) SDOD_HIP_CHECK(hipEventCreate(&ev[i]));
    float* sink = (float*)scratch;
    const size_t a_bytes = d->a_mode == SDOD_A_ROWS ? (size_t)d->M * d->lda * 2 : (size_t)d->n_img * d->h_in * d->w_in * d->c0 * 2;
    for (int i = 0; i < iters; ++i) {
        SDOD_LAUNCH(cache_sweep_kernel, dim3(2048), dim3(256), 0, st, (float4*)scratch, scratch_bytes / 16);
        touch(d->a, a_bytes, sink, st);
        if (d->a2 && d->c1) touch(d->a2, (size_t)d->n_img * d->h_in * d->w_in * d->c1 * 2, sink, st);
        if (d->residual) touch(d->residual, (size_t)d->M * d->ldr * 2, sink, st);
        if (d->t0 && d->k_tail) touch(d->t0, (size_t)d->M * d->tc0 * 2, sink, st);
        if (d->t1 && d->k_tail && d->tc1) touch(d->t1, (size_t)d->M * d->tc1 * 2, sink, st);
        SDOD_HIP_CHECK(hipEventRecord(ev[2 * i], st));
        rc = sdod_gemm_f16(d, stream);
        if (rc) return rc;
        SDOD_HIP_CHECK(hipEventRecord(ev[2 * i + 1], st));
    }
    SDOD_HIP_CHECK(hipEventSynchronize(ev[2 * iters - 1]));
    float tot = 0.f, best = 1e30f;
    for (int i = 0; i < iters; ++i) {
        float ms = 0.f;
        SDOD_HIP_CHECK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        tot += ms;
        best = ms < best ? ms : best;
    }
    for (int i = 0; i < 2 * iters; ++i) (void)hipEventDestroy(ev[i]);
    *ms_avg = tot / iters;
    if (ms_min) *ms_min = best; // disturbances only ever add time: the minimum is the steadier ranking key
    return 0;
    SDOD_CATCH
}

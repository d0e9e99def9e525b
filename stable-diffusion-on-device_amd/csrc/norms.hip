// norms.hip -- GroupNorm(+SiLU) and LayerNorm for NHWC activations on gfx950.
//
// GroupNorm is the operator the reference exposes as sdod.EfficientGN / efficient_group_norm
// (sdod/efficient_gn.py:9-30, :61-70) and declares -- without ever implementing a kernel -- as the
// custom op `sdod::GroupNorm(in[0], weight, bias; num_groups, eps)` (csrc/sdod_ops/config/group_norm.xml:17-106,
// group_norm.json:5-21, layout NHWC).  Semantics = torch.nn.functional.group_norm: biased variance over
// (C/G, spatial) per (n, g), y = (x-mean)/sqrt(var+eps)*w_c+b_c.
//
// HBM-bound design: in NHWC a group's data is a strided set of short channel runs, so a block-per-(n,g)
// kernel would read 20..160-byte fragments.  Instead the statistics pass reads whole pixel rows (fully
// coalesced 16-byte lanes, many workgroups), every thread owning a fixed 8-channel chunk and keeping
// per-channel shifted sums in registers; per-group partials go through LDS to a small fp32 scratch.
// The apply pass first reduces those partials to mean/rstd per (n,g) (fixed order, one wave per group -- cheaper than
// a third launch), then runs as a vectorised elementwise kernel whose per-channel scale/shift live in LDS, with SiLU fused (every ResBlock site) and the
// channel concat of the UNet skip connections folded into the reads.  Sums are shifted by a per-group
// pilot value (first element of the group) so E[x^2]-E[x]^2 cancellation stays harmless in fp32.
// Algorithmic bytes: N*HW*C*sizeof(T) read twice + written once (the second read is L2/MALL resident).
#include "common.h"
#include "sdod_hip.h"
#include "host_util.h"

#include <algorithm>
#include <cstdlib>

namespace {

template <typename T>
struct Chunk8;
template <>
struct Chunk8<f16> {
    static SDOD_DEVICE void load(const f16* p, float (&v)[8]) {
        const f16x8 h = ldg8(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
    }
    static SDOD_DEVICE void store(f16* p, const float (&v)[8]) {
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (f16)v[e];
        stg8(p, h);
    }
};
template <>
struct Chunk8<float> {
    static SDOD_DEVICE void load(const float* p, float (&v)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
        v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    static SDOD_DEVICE void store(float* p, const float (&v)[8]) {
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
};

struct GnP {
    const void* x0;
    const void* x1;
    void* y;
    const float* w;
    const float* b;
    float* partial; // [N][nchunks][G][2]
    float* stats;   // [N][G][2] mean, rstd
    float* shift;   // [N][G] pilot value of every group, written by the statistics pass (the apply pass must not re-read x:
                    // with y == x another workgroup may already have overwritten pixel 0)
    void* sync;     // grid-barrier words of the one-launch kernel for big maps (zeroed once by the workspace's owner)
    int N, HW, C0, C1, C, G, Cg;
    float eps;
    int silu;
    int nchunks, pix_per_chunk;
    int cpp, npass, pp;
};

// address of channel c of pixel `pix` of image n in the (possibly concatenated) input
template <typename T>
SDOD_DEVICE const T* gn_src(const GnP& p, int n, int pix, int c) {
    if (c < p.C0) return reinterpret_cast<const T*>(p.x0) + ((size_t)n * p.HW + pix) * p.C0 + c;
    return reinterpret_cast<const T*>(p.x1) + ((size_t)n * p.HW + pix) * p.C1 + (c - p.C0);
}

template <typename T, int NPASS>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw); // [pp][C][2]
    const int n = blockIdx.y;
    const int chunk_id = blockIdx.x;
    const int cx = threadIdx.x, py = threadIdx.y;
    const int pix_begin = chunk_id * p.pix_per_chunk;
    const int pix_end = min(p.HW, pix_begin + p.pix_per_chunk);

    float s1[NPASS][8], s2[NPASS][8], shift[NPASS][8];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int c0 = (cx + ps * p.cpp) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s1[ps][e] = 0.f;
            s2[ps][e] = 0.f;
            const int g = (c0 + e) / p.Cg;
            shift[ps][e] = (float)*gn_src<T>(p, n, 0, g * p.Cg);
        }
    }
    for (int pix = pix_begin + py; pix < pix_end; pix += p.pp) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int c0 = (cx + ps * p.cpp) * 8;
            float v[8];
            Chunk8<T>::load(gn_src<T>(p, n, pix, c0), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[e] - shift[ps][e];
                s1[ps][e] += d;
                s2[ps][e] += d * d;
            }
        }
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int c0 = (cx + ps * p.cpp) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[((size_t)py * p.C + c0 + e) * 2 + 0] = s1[ps][e];
            red[((size_t)py * p.C + c0 + e) * 2 + 1] = s2[ps][e];
        }
    }
    __syncthreads();
    const int tid = py * blockDim.x + cx;
    const int nthreads = blockDim.x * blockDim.y;
    for (int g = tid; g < p.G; g += nthreads) {
        float a = 0.f, b = 0.f;
        for (int q = 0; q < p.pp; ++q)
            for (int c = g * p.Cg; c < (g + 1) * p.Cg; ++c) {
                a += red[((size_t)q * p.C + c) * 2 + 0];
                b += red[((size_t)q * p.C + c) * 2 + 1];
            }
        float* dst = p.partial + (((size_t)n * p.nchunks + chunk_id) * p.G + g) * 2;
        dst[0] = a;
        dst[1] = b;
        if (chunk_id == 0) p.shift[(size_t)n * p.G + g] = (float)*gn_src<T>(p, n, 0, g * p.Cg);
    }
}

// Large maps (VAE 256^2 / 512^2 levels) need hundreds of statistics workgroups to stream at HBM rate; their partials are
// collapsed to one entry per (n, group) by this small kernel so that the apply pass keeps reading a short list.
__global__ __launch_bounds__(256) void gn_collapse_kernel(const float* partial, float* collapsed, int nchunks, int G) {
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    float a = 0.f, b = 0.f;
    for (int ch = lane; ch < nchunks; ch += 64) {
        const float* src = partial + (((size_t)n * nchunks + ch) * G + g) * 2;
        a += src[0];
        b += src[1];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
        collapsed[((size_t)n * G + g) * 2 + 0] = a;
        collapsed[((size_t)n * G + g) * 2 + 1] = b;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnP p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sc = reinterpret_cast<float*>(smem_raw); // [C] scale
    float* sh = sc + p.C;                            // [C] shift
    float* gm = sh + p.C;                            // [G] mean
    float* gr = gm + p.G;                            // [G] rstd
    const int n = blockIdx.y;
    // finalize the statistics here (every workgroup redoes this small, fixed-order reduction; it costs less than the
    // extra launch a separate finalize kernel would): one wave per group, lanes stride over the chunk partials
    {
        const int lane = threadIdx.x & 63;
        for (int g = threadIdx.x >> 6; g < p.G; g += blockDim.x >> 6) {
            float a = 0.f, b = 0.f;
            for (int ch = lane; ch < p.nchunks; ch += 64) {
                const float* src = p.partial + (((size_t)n * p.nchunks + ch) * p.G + g) * 2;
                a += src[0];
                b += src[1];
            }
            a = wave_sum(a);
            b = wave_sum(b);
            if (lane == 0) {
                const float cnt = (float)p.HW * (float)p.Cg;
                const float shift = p.shift[(size_t)n * p.G + g];
                const float icnt = __builtin_amdgcn_rcpf(cnt);
                const float md = a * icnt;
                float var = b * icnt - md * md;
                var = var < 0.f ? 0.f : var;
                gm[g] = shift + md;
                gr[g] = rsqrt_fast(var + p.eps);
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        const int g = c / p.Cg;
        const float mean = gm[g];
        const float rstd = gr[g];
        const float w = p.w ? p.w[c] : 1.0f;
        const float b = p.b ? p.b[c] : 0.0f;
        sc[c] = rstd * w;
        sh[c] = b - mean * rstd * w;
    }
    __syncthreads();
    const int cp = p.C / 8;
    const size_t total = (size_t)p.HW * cp;
    T* yout = reinterpret_cast<T*>(p.y) + (size_t)n * p.HW * p.C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int pix = (int)(i / cp);
        const int c0 = (int)(i - (size_t)pix * cp) * 8;
        float v[8];
        Chunk8<T>::load(gn_src<T>(p, n, pix, c0), v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = v[e] * sc[c0 + e] + sh[c0 + e];
            if (p.silu) f = silu_f(f);
            v[e] = f;
        }
        Chunk8<T>::store(yout + (size_t)pix * p.C + c0, v);
    }
}

// Small feature maps (the 8x8 / 16x16 / 32x32 levels: 2/3 of the UNet's GroupNorms): ONE launch.  A workgroup owns all
// pixels of image n for a channel set of `sw` channels (whole groups, 16-byte aligned: sw = lcm(C/G, 8)); the slab is
// read once into LDS, statistics are an exact two-pass reduction over LDS, then normalise + SiLU + store.
template <typename T, int NT>
__global__ __launch_bounds__(NT) void gn_small_kernel(const GnP p, int sw) {
    constexpr int NWV = NT / 64;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* slab = reinterpret_cast<T*>(smem_raw);                               // [HW][sw]
    float* sc = reinterpret_cast<float*>(smem_raw + (size_t)p.HW * sw * sizeof(T)); // [sw] scale
    float* sh = sc + sw;                                                    // [sw] shift
    float* red = sh + sw;                                                   // [NWV waves][4 groups][2], then the group table
    const int n = blockIdx.y;
    const int cbase = blockIdx.x * sw;
    const int cps = sw / 8;
    const int total = p.HW * cps;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the affine parameters of "my" channel are requested first: their L2 round trip hides behind the slab load instead of
    // standing alone between the statistics and the normalisation
    float pw = 1.0f, pb = 0.0f;
    if (tid < sw && p.w != nullptr) {
        pw = p.w[cbase + tid];
        pb = p.b[cbase + tid];
    }
    for (int i = tid; i < total; i += NT) {
        const int pix = i / cps, cc = i - pix * cps;
        const T* src = gn_src<T>(p, n, pix, cbase + cc * 8);
        if (sizeof(T) == 2) {
            *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(slab) + (size_t)pix * sw + cc * 8) = ldg8(reinterpret_cast<const f16*>(src));
        } else {
            float v[8];
            Chunk8<T>::load(src, v);
            Chunk8<T>::store(slab + (size_t)pix * sw + cc * 8, v);
        }
    }
    // group of every channel of the set (sw small divisions instead of one per element)
    unsigned char* gidx = reinterpret_cast<unsigned char*>(red + NWV * 8);
    for (int c = tid; c < sw; c += NT) gidx[c] = (unsigned char)(c / p.Cg);
    __syncthreads();
    const int ng = sw / p.Cg; // <= 4 for every SD channel count (asserted on the host)
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f}, shf[4];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) shf[gi] = gi < ng ? (float)slab[gi * p.Cg] : 0.f; // pilot shift: first element of the group
    for (int i = tid; i < total; i += NT) {
        const int pix = i / cps, cc = i - pix * cps;
        float v[8];
        Chunk8<T>::load(slab + (size_t)pix * sw + cc * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int g = gidx[cc * 8 + e];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const float d = g == gi ? v[e] - shf[gi] : 0.f;
                s1[gi] += d;
                s2[gi] += d * d;
            }
        }
    }
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
        s1[gi] = wave_sum(s1[gi]);
        s2[gi] = wave_sum(s2[gi]);
        if (lane == 0) {
            red[(wave * 4 + gi) * 2 + 0] = s1[gi];
            red[(wave * 4 + gi) * 2 + 1] = s2[gi];
        }
    }
    __syncthreads();
    const float inv_cnt = 1.0f / ((float)p.HW * (float)p.Cg);
    for (int c = tid; c < sw; c += NT) {
        const int gi = gidx[c];
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) {
            a += red[(w * 4 + gi) * 2 + 0];
            b += red[(w * 4 + gi) * 2 + 1];
        }
        const float md = a * inv_cnt;
        float var = b * inv_cnt - md * md;
        var = var < 0.f ? 0.f : var;
        const float mean = (float)slab[gi * p.Cg] + md;
        const float rstd = rsqrt_fast(var + p.eps);
        const int ch = cbase + c;
        const float w = c == tid ? pw : (p.w ? p.w[ch] : 1.0f), bb = c == tid ? pb : (p.b ? p.b[ch] : 0.0f);
        sc[c] = rstd * w;
        sh[c] = bb - mean * rstd * w;
    }
    __syncthreads();
    T* yout = reinterpret_cast<T*>(p.y) + (size_t)n * p.HW * p.C + cbase;
    for (int i = tid; i < total; i += NT) {
        const int pix = i / cps, cc = i - pix * cps;
        float v[8];
        Chunk8<T>::load(slab + (size_t)pix * sw + cc * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = v[e] * sc[cc * 8 + e] + sh[cc * 8 + e];
            if (p.silu) f = silu_f(f);
            v[e] = f;
        }
        Chunk8<T>::store(yout + (size_t)pix * p.C + cc * 8, v);
    }
}


// ------------------------------------------------------------------------------------------------
// One-launch GroupNorm, a workgroup per (image, group): the form every GroupNorm of the UNet takes (HW <= 4096 at
// C = 320 / 640, every map of the 32x32 .. 8x8 levels) and the VAE's 64x64 maps.  A group is HW runs of Cg channels
// (20 .. 160 bytes each); thread t owns vector `t % vpp` of the run (V = 2 / 4 / 8 halves, the widest that divides Cg)
// for pixels t / vpp, t / vpp + ppp, ..., so its channels -- and their affine parameters -- are fixed, every load is
// issued before the first use (<= KMAX vectors per thread, held in registers), the statistics are an exact two-pass
// mean / variance on those registers (two block reductions), and normalise + SiLU + store follow without touching x again.
// In-place safe: a thread only overwrites what it has read itself.
//
// RED = the split-K reduction of the producing conv folded into the load (sdod_gn_reduce): source 0 is still S fp32 partial
// slabs; x = fp16(act(alpha * sum_s partial + bias + bias2 + row_bias)) + residual, exactly the arithmetic of
// splitk_reduce_kernel (gemm.hip), is formed in registers, stored to x_out for the tensor's later readers (skip
// connections, residuals) and normalised in the same pass: one launch instead of three.
struct GnRed {
    const float* partial;
    int splits;
    size_t slab; // floats per split slab (M * C0)
    const float* bias;
    const float* bias2;
    const f16* row_bias;
    int ldrb;
    const f16* residual;
    int ldr;
    f16* xout;
    float alpha;
    int act;
};
struct GnG {
    const f16* x0;
    const f16* x1;
    f16* y;
    const float* w;
    const float* b;
    int HW, C0, C1, C, Cg;
    float eps;
    int silu;
    int vpp, ppp; // vectors per pixel of one group; pixels per pass (= active threads / vpp)
    GnRed r;
};

template <int V> struct HalfVec;
template <> struct HalfVec<2> { typedef f16x2 T; };
template <> struct HalfVec<4> { typedef f16x4 T; };
template <> struct HalfVec<8> { typedef f16x8 T; };
template <int V> struct FloatVec { float v[V]; };

template <int V>
SDOD_DEVICE FloatVec<V> ldf(const float* p) {
    FloatVec<V> o;
    if constexpr (V == 2) {
        const f32x2 t = *reinterpret_cast<const f32x2*>(p);
        o.v[0] = t[0]; o.v[1] = t[1];
    } else {
#pragma unroll
        for (int q = 0; q < V / 4; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
            o.v[4 * q] = t[0]; o.v[4 * q + 1] = t[1]; o.v[4 * q + 2] = t[2]; o.v[4 * q + 3] = t[3];
        }
    }
    return o;
}

// keeps the PACKED fp16 registers as the live form between the phases (the compiler would otherwise hoist the fp32
// conversions of all KMAX vectors above the reductions and spill)
template <typename VT>
SDOD_DEVICE void pin(VT& v) {
    asm volatile("" : "+v"(v));
}

template <int NT>
SDOD_DEVICE float block_sum(float v, float* red, int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) t += red[w];
    __syncthreads(); // `red` is reused by the next reduction
    return t;
}

template <int V, int NT, int KMAX, bool RED>
__global__ __launch_bounds__(NT) void gn_group_kernel(const GnG p) {
    typedef typename HalfVec<V>::T VT;
    __shared__ float red[NT / 64];
    // workgroups b and b + 8 share an XCD (round-robin dispatch; speed only): give each XCD a run of CONSECUTIVE groups, whose
    // 20..160-byte channel runs share cache lines, instead of every 8th group (each XCD's L2 would pull the whole tensor)
    const int gpx = gridDim.x >> 3;
    const int g = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * gpx + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int n = blockIdx.y, tid = threadIdx.x;
    const bool active = tid < p.ppp * p.vpp;
    const int pix0 = tid / p.vpp;
    const int c = g * p.Cg + (tid - pix0 * p.vpp) * V; // first of this thread's V channels (concatenated channel space)
    const bool src1 = c >= p.C0;
    const size_t row0 = (size_t)n * p.HW;
    const f16* src = src1 ? p.x1 + row0 * p.C1 + (c - p.C0) : p.x0 + row0 * p.C0 + c;
    const int sstride = src1 ? p.C1 : p.C0;

    // affine parameters of "my" channels first: their round trip hides behind the loads below
    float pw[V], pb[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        pw[e] = (active && p.w) ? p.w[c + e] : 1.0f;
        pb[e] = (active && p.b) ? p.b[c + e] : 0.0f;
    }

    VT vals[KMAX];
    if (RED && !src1) {
        if constexpr (RED) {
            float acc[KMAX][V];
#pragma unroll
            for (int k = 0; k < KMAX; ++k)
#pragma unroll
                for (int e = 0; e < V; ++e) acc[k][e] = 0.f;
            // U slabs are requested before the first is summed (a one-slab-at-a-time loop pays a full memory round trip per
            // split); the sums are still taken in ascending split order, as splitk_reduce_kernel does (+0 for the tail slots)
            // staged floats per thread: 128 in a 256-thread workgroup (512 registers per thread available), 64 at 1024 threads
            constexpr int UB = (NT == 256 ? 128 : KMAX <= 2 ? 64 : 32) / (KMAX * V);
            constexpr int U = UB >= 8 ? 8 : UB >= 4 ? 4 : UB >= 2 ? 2 : 1;
            for (int s0 = 0; s0 < p.r.splits; s0 += U) {
                FloatVec<V> t[U][KMAX];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool sok = s0 + u < p.r.splits;
                    const float* ps = p.r.partial + (size_t)(s0 + u) * p.r.slab + row0 * p.C0 + c;
#pragma unroll
                    for (int k = 0; k < KMAX; ++k) {
                        const int pix = pix0 + k * p.ppp;
                        if (sok && active && pix < p.HW) {
                            t[u][k] = ldf<V>(ps + (size_t)pix * p.C0);
                        } else {
#pragma unroll
                            for (int e = 0; e < V; ++e) t[u][k].v[e] = 0.f;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int k = 0; k < KMAX; ++k)
#pragma unroll
                        for (int e = 0; e < V; ++e) acc[k][e] += t[u][k].v[e];
            }
            float b1[V], b2[V], rb[V];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                b1[e] = (active && p.r.bias) ? p.r.bias[c + e] : 0.f;
                b2[e] = (active && p.r.bias2) ? p.r.bias2[c + e] : 0.f;
                rb[e] = (active && p.r.row_bias) ? (float)p.r.row_bias[(size_t)n * p.r.ldrb + c + e] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int pix = pix0 + k * p.ppp;
                VT h;
#pragma unroll
                for (int e = 0; e < V; ++e) h[e] = (f16)0.f;
                if (active && pix < p.HW) {
                    VT rs;
#pragma unroll
                    for (int e = 0; e < V; ++e) rs[e] = (f16)0.f;
                    if (p.r.residual) rs = *reinterpret_cast<const VT*>(p.r.residual + (row0 + pix) * p.r.ldr + c);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        float f = acc[k][e] * p.r.alpha;
                        if (p.r.bias) f += b1[e];
                        if (p.r.bias2) f += b2[e];
                        if (p.r.row_bias) f += rb[e];
                        f = apply_act(f, p.r.act);
                        f = (float)(f16)f; // same rounding point as the un-split path
                        if (p.r.residual) f += (float)rs[e];
                        h[e] = (f16)f;
                    }
                    *reinterpret_cast<VT*>(p.r.xout + (row0 + pix) * p.C0 + c) = h;
                }
                vals[k] = h;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int pix = pix0 + k * p.ppp;
            VT h;
#pragma unroll
            for (int e = 0; e < V; ++e) h[e] = (f16)0.f;
            if (active && pix < p.HW) h = *reinterpret_cast<const VT*>(src + (size_t)pix * sstride);
            vals[k] = h;
        }
    }

    // exact two-pass statistics on the registers (invalid slots hold zeros and are excluded from the second pass)
#pragma unroll
    for (int k = 0; k < KMAX; ++k) pin(vals[k]);
    float s1 = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) s1 += (float)vals[k][e];
    const float inv_cnt = 1.0f / ((float)p.HW * (float)p.Cg);
    const float mean = block_sum<NT>(s1, red, tid) * inv_cnt;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) pin(vals[k]);
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const bool ok = active && (pix0 + k * p.ppp) < p.HW;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float d = (float)vals[k][e] - mean;
            s2 += ok ? d * d : 0.f;
        }
    }
    const float var = block_sum<NT>(s2, red, tid) * inv_cnt;
    const float rstd = rsqrt_fast(var + p.eps);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) pin(vals[k]);

    float sc[V], sh[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        sc[e] = rstd * pw[e];
        sh[e] = pb[e] - mean * rstd * pw[e];
    }
    f16* dst = p.y + row0 * p.C + c;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int pix = pix0 + k * p.ppp;
        if (active && pix < p.HW) {
            VT o;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float f = (float)vals[k][e] * sc[e] + sh[e];
                if (p.silu) f = silu_f(f);
                o[e] = (f16)f;
            }
            *reinterpret_cast<VT*>(dst + (size_t)pix * p.C) = o;
        }
    }
}

// tier of the group kernel for a shape: 0 = not eligible.  id = V * 1000 + (NT == 1024 ? 100 : 0) + KMAX
struct GroupPlan {
    int v = 0, nt = 0, kmax = 0, vpp = 0, ppp = 0;
};
static GroupPlan gn_group_plan(int hw, int c0, int c1, int cg, bool red) {
    GroupPlan pl;
    const int v = cg % 8 == 0 ? 8 : cg % 4 == 0 ? 4 : cg % 2 == 0 ? 2 : 0;
    if (!v || c0 % 8 || c1 % 8) return pl;
    const int vpp = cg / v;
    if (vpp > 256) return pl;
    int nt = (size_t)hw * vpp <= 512 ? 256 : 1024;
    int ppp = nt / vpp;
    int k = (hw + ppp - 1) / ppp;
    int kmax = 0;
    if (nt == 256) {
        if (k <= 2) kmax = 2;
        else { nt = 1024; ppp = nt / vpp; k = (hw + ppp - 1) / ppp; }
    }
    // register budget at 1024 threads (128 VGPRs): the packed vectors may take about 96 of them
    if (nt == 1024) kmax = k <= 2 ? 2 : k <= 8 ? 8 : (k <= 16 && v == 8) ? 16 : (k <= 24 && v <= 4) ? 24 : (k <= 48 && v == 2) ? 48 : 0;
    if (!kmax || (red && kmax * v > 32)) return pl; // fused reduce: fp32 sums + one staged slab must fit the register budget
    pl.v = v; pl.nt = nt; pl.kmax = kmax; pl.vpp = vpp; pl.ppp = ppp;
    return pl;
}

template <int V, int NT, int KMAX>
static void gn_group_launch2(const GnG& g, int G, int N, bool red, hipStream_t st) {
    if constexpr (KMAX * V <= 32) {
        if (red) {
            SDOD_LAUNCH((gn_group_kernel<V, NT, KMAX, true>), dim3(G, N), dim3(NT), 0, st, g);
            return;
        }
    }
    SDOD_LAUNCH((gn_group_kernel<V, NT, KMAX, false>), dim3(G, N), dim3(NT), 0, st, g);
}
template <int V>
static void gn_group_launch1(const GnG& g, const GroupPlan& pl, int G, int N, bool red, hipStream_t st) {
    if (pl.nt == 256) gn_group_launch2<V, 256, 2>(g, G, N, red, st);
    else if (pl.kmax == 2) gn_group_launch2<V, 1024, 2>(g, G, N, red, st);
    else if (pl.kmax == 8) gn_group_launch2<V, 1024, 8>(g, G, N, red, st);
    else if (pl.kmax == 16) {
        if constexpr (V == 8) gn_group_launch2<8, 1024, 16>(g, G, N, red, st);
    } else if (pl.kmax == 24) {
        if constexpr (V <= 4) gn_group_launch2<V, 1024, 24>(g, G, N, red, st);
    } else if constexpr (V == 2) gn_group_launch2<2, 1024, 48>(g, G, N, red, st);
}
// returns false when the shape is not eligible (the caller falls back to the other paths)
static bool gn_group_try(const GnP& p, const GnRed* red, hipStream_t st) {
    const GroupPlan pl = gn_group_plan(p.HW, p.C0, p.C1, p.Cg, red != nullptr);
    if (!pl.v) return false;
    GnG g{};
    g.x0 = (const f16*)p.x0; g.x1 = (const f16*)p.x1; g.y = (f16*)p.y; g.w = p.w; g.b = p.b;
    g.HW = p.HW; g.C0 = p.C0; g.C1 = p.C1; g.C = p.C; g.Cg = p.Cg; g.eps = p.eps; g.silu = p.silu;
    g.vpp = pl.vpp; g.ppp = pl.ppp;
    if (red) g.r = *red;
    if (pl.v == 8) gn_group_launch1<8>(g, pl, p.G, p.N, red != nullptr, st);
    else if (pl.v == 4) gn_group_launch1<4>(g, pl, p.G, p.N, red != nullptr, st);
    else gn_group_launch1<2>(g, pl, p.G, p.N, red != nullptr, st);
    SDOD_HIP_CHECK(hipGetLastError());
    return true;
}


// ------------------------------------------------------------------------------------------------
// Lean two-launch GroupNorm for fp16 maps the (image, group) kernel does not take (or takes slowly: at Cg = 10 a group is
// 20 of every 640 bytes, so that kernel fetches 6x the bytes it uses): both passes read whole pixel rows (16-byte lanes,
// every CU busy), thread t owns chunk t % cp of the row (its 8 channels, their <= 2 groups and their affine parameters
// are fixed), all loads of a pass are issued before the first use, and both reductions run in a fixed order
// (bit-reproducible).  Statistics: shifted sums per (image, pixel chunk, group) -> workspace; apply: every workgroup
// re-reduces the short partial list (8 lanes per group + shuffles), builds scale / shift in LDS and streams its pixels.
struct Gn2P {
    const f16* x0;
    const f16* x1;
    f16* y;
    const float* w;
    const float* b;
    float* partial; // [N][nchunks][G][2]
    float* shift;   // [N][G]
    int HW, C0, C1, C, G, Cg, cp, rp; // cp = 16-byte chunks per pixel row, rp = pixel rows per pass (256 / cp)
    int nchunks, ppc;                 // statistics pass: pixel chunks per image, pixels per chunk
    int ppb;                          // apply pass: pixels per workgroup
    float eps;
    int silu;
};

SDOD_DEVICE const f16* gn2_src(const Gn2P& p, size_t row, int c) {
    return c < p.C0 ? p.x0 + row * p.C0 + c : p.x1 + row * p.C1 + (c - p.C0);
}

__global__ __launch_bounds__(256) void gn2_stats_kernel(const Gn2P p) {
    __shared__ float red[256 * 4];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const bool active = tid < p.rp * p.cp;
    const int r0 = tid / p.cp, j = tid - r0 * p.cp;
    const int c0 = j * 8;
    const int gA = c0 / p.Cg;
    const int eb = min(8, (gA + 1) * p.Cg - c0); // channels [0, eb) of the chunk belong to group gA, the rest to gA + 1
    const size_t row0 = (size_t)n * p.HW;
    const float shA = (float)*gn2_src(p, row0, gA * p.Cg);
    const float shB = eb < 8 ? (float)*gn2_src(p, row0, (gA + 1) * p.Cg) : 0.f;
    const int pix_end = min(p.HW, (chunk + 1) * p.ppc);
    float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
    for (int pix0 = chunk * p.ppc + r0; pix0 < pix_end; pix0 += 8 * p.rp) { // one trip for the UNet's maps (ppc <= 8 rp)
        f16x8 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int pix = pix0 + k * p.rp;
            v[k] = (active && pix < pix_end) ? ldg8(gn2_src(p, row0 + pix, c0)) : zero8();
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool ok = active && (pix0 + k * p.rp) < pix_end;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool inA = e < eb;
                const float d = ok ? (float)v[k][e] - (inA ? shA : shB) : 0.f;
                a1 += inA ? d : 0.f; a2 += inA ? d * d : 0.f;
                b1 += inA ? 0.f : d; b2 += inA ? 0.f : d * d;
            }
        }
    }
    red[tid * 4 + 0] = a1; red[tid * 4 + 1] = a2; red[tid * 4 + 2] = b1; red[tid * 4 + 3] = b2;
    __syncthreads();
    if (tid < p.G) {
        const int g = tid;
        const int j0 = (g * p.Cg) >> 3, j1 = ((g + 1) * p.Cg - 1) >> 3;
        float s1 = 0.f, s2 = 0.f;
        for (int jj = j0; jj <= j1; ++jj) {
            const int first = (jj * 8) / p.Cg;      // group of the chunk's first channel
            const int slot = first == g ? 0 : 2;    // else g is the chunk's second group
            for (int r = 0; r < p.rp; ++r) {
                s1 += red[(r * p.cp + jj) * 4 + slot];
                s2 += red[(r * p.cp + jj) * 4 + slot + 1];
            }
        }
        float* dst = p.partial + (((size_t)n * p.nchunks + chunk) * p.G + g) * 2;
        dst[0] = s1;
        dst[1] = s2;
        if (chunk == 0) p.shift[(size_t)n * p.G + g] = (float)*gn2_src(p, row0, g * p.Cg);
    }
}

__global__ __launch_bounds__(256) void gn2_apply_kernel(const Gn2P p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sc = reinterpret_cast<float*>(smem_raw); // [C] scale, [C] shift, [G] mean, [G] rstd
    float* sh = sc + p.C;
    float* gm = sh + p.C;
    float* gr = gm + p.G;
    const int n = blockIdx.y, tid = threadIdx.x;
    // (1) mean / rstd of every group from the per-chunk partials: 8 lanes per group, fixed order
    for (int g = tid >> 3; g < p.G; g += 32) {
        const int l = tid & 7;
        float a = 0.f, b = 0.f;
        for (int ch = l; ch < p.nchunks; ch += 8) {
            const float* src = p.partial + (((size_t)n * p.nchunks + ch) * p.G + g) * 2;
            a += src[0];
            b += src[1];
        }
        a = oct_sum(a);
        b = oct_sum(b);
        if (l == 0) {
            const float cnt = (float)p.HW * (float)p.Cg;
            const float icnt = __builtin_amdgcn_rcpf(cnt);
            const float md = a * icnt;
            float var = b * icnt - md * md;
            var = var < 0.f ? 0.f : var;
            gm[g] = p.shift[(size_t)n * p.G + g] + md;
            gr[g] = rsqrt_fast(var + p.eps);
        }
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        const int g = c / p.Cg;
        const float wv = p.w ? p.w[c] : 1.0f, bv = p.b ? p.b[c] : 0.0f;
        sc[c] = gr[g] * wv;
        sh[c] = bv - gm[g] * gr[g] * wv;
    }
    __syncthreads();
    // (2) stream this workgroup's pixels
    const bool active = tid < p.rp * p.cp;
    const int r0 = tid / p.cp, j = tid - r0 * p.cp;
    const int c0 = j * 8;
    float s8[8], t8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s8[e] = active ? sc[c0 + e] : 0.f;
        t8[e] = active ? sh[c0 + e] : 0.f;
    }
    const size_t row0 = (size_t)n * p.HW;
    const int pix_begin = blockIdx.x * p.ppb, pix_end = min(p.HW, pix_begin + p.ppb);
    for (int base = pix_begin + r0; base < pix_end; base += 4 * p.rp) {
        f16x8 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pix = base + k * p.rp;
            v[k] = (active && pix < pix_end) ? ldg8(gn2_src(p, row0 + pix, c0)) : zero8();
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int pix = base + k * p.rp;
            if (active && pix < pix_end) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (float)v[k][e] * s8[e] + t8[e];
                    if (p.silu) f = silu_f(f);
                    o[e] = (f16)f;
                }
                stg8(p.y + (row0 + pix) * p.C + c0, o);
            }
        }
    }
}

// true when the lean pair takes the shape (fp16, <= 256 chunks per pixel row, a chunk touches <= 2 groups, G <= 32 x ...)
static bool gn2_fits(int c0, int c1, int groups) {
    const int c = c0 + c1;
    if (c % 8 || c0 % 8 || c1 % 8 || c / 8 > 256 || c > 2048 || groups > 256) return false;
    const int cg = c / groups;
    return cg >= 4 && (cg >= 8 || 8 % cg == 0);
}

static void gn2_launch(const GnP& q, hipStream_t st) {
    Gn2P p{};
    p.x0 = (const f16*)q.x0; p.x1 = (const f16*)q.x1; p.y = (f16*)q.y; p.w = q.w; p.b = q.b;
    p.partial = q.partial; p.shift = q.shift;
    p.HW = q.HW; p.C0 = q.C0; p.C1 = q.C1; p.C = q.C; p.G = q.G; p.Cg = q.Cg; p.eps = q.eps; p.silu = q.silu;
    p.cp = q.C / 8;
    p.rp = 256 / p.cp;
    // statistics: <= 8 rows of the pass pattern per workgroup, and enough workgroups to cover the chip (~512), capped by the
    // workspace layout (GN_MAX_CHUNKS chunks per image)
    int ppc = p.rp * 8;
    while (ppc > p.rp && (size_t)((q.HW + ppc / 2 - 1) / (ppc / 2)) * q.N <= 512) ppc /= 2;
    p.nchunks = (q.HW + ppc - 1) / ppc;
    while (p.nchunks > 1024) { ppc *= 2; p.nchunks = (q.HW + ppc - 1) / ppc; }
    p.ppc = ppc;
    SDOD_LAUNCH(gn2_stats_kernel, dim3(p.nchunks, q.N), dim3(256), 0, st, p);
    SDOD_HIP_CHECK(hipGetLastError());
    int ppb = p.rp * 8;
    while (ppb > p.rp && (size_t)((q.HW + ppb / 2 - 1) / (ppb / 2)) * q.N <= 1024) ppb /= 2;
    p.ppb = ppb;
    const int bx = (q.HW + ppb - 1) / ppb;
    SDOD_LAUNCH(gn2_apply_kernel, dim3(bx, q.N), dim3(256), ((size_t)q.C * 2 + (size_t)q.G * 2) * sizeof(float), st, p);
    SDOD_HIP_CHECK(hipGetLastError());
}

constexpr int GN_MAX_CHUNKS_GRID = 256;

// ------------------------------------------------------------------------------------------------
// One-launch GroupNorm for the BIG fp16 maps (>= 5 MB): every CU streams whole pixel rows ONCE.
//   The (image, group) kernel reads a group as Cg of every C channels -- 20 of every 640 bytes at 64x64 x 320 -- from only
//   N*G = 64 workgroups (18-33 us for 10-30 MB); two launches that each read whole rows pay the launch + first-byte latency
//   twice (the lean pair above: no faster).  Here one launch covers the chip with one workgroup per CU, image n owning
//   wpi = (#CUs / N) of them; a workgroup loads its ppw pixels x C channels into REGISTERS (16-byte lanes, all loads in
//   flight at once), reduces shifted sums per group, publishes them, meets the other workgroups at a grid barrier, reduces
//   the wpi partials of its image in a fixed order (bit-reproducible), and normalises + stores from the registers.
//   HBM/L2 traffic: x once, y once.
//   Grid barrier: a small counter tree + generation words in the caller's workspace (sdod_group_norm_workspace_bytes;
//   zeroed once by its owner, re-armed by every launch).  Every workgroup must be resident: the grid is at most one
//   workgroup per CU (512 threads, < 32 KB LDS: several fit per CU, so a few such kernels on different streams still
//   co-reside).  The wait is bounded: a launch that cannot meet (counter clobbered, or two grids that
//   starve each other of CUs) gives up after 2^20 polls (~1 s) instead of hanging the device, and says so: a sticky error word
//   (gn_error_word(), sdod_group_norm_status) that turns the caller's next host-side check into LIBSDOD_RUNTIME_ERROR.
struct GnGridP {
    const f16* x0;
    const f16* x1;
    f16* y;
    const float* w;
    const float* b;
    float* partial;   // [N][wpi][G][2]
    unsigned* sync;   // 25 words, one per 128-byte line: [0] top counter, [1..16] shard counters, [17..24] generation replicas;
                      // line 25 = the workspace's sticky timeout word
    unsigned* err_host; // host-mapped sticky error word of this device (may be null): a timed-out barrier is reported, see below
    int N, HW, C0, C1, C, G, Cg;
    int cp, rp;       // 16-byte chunks per pixel row; pixel rows per pass of the 512 threads
    int wpi, ppw, nwg; // workgroups per image, pixels per workgroup, workgroups in the grid
    int nloop;         // passes of KMAX vectors per thread (1: the pixels stay in registers between the phases)
    float eps;
    int silu;
};

template <int KMAX>
__global__ __launch_bounds__(512) void gn_grid_kernel(const GnGridP p) {
    __shared__ float red[512 * 4];
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sc = reinterpret_cast<float*>(smem_raw); // [C] scale, [C] shift, [G] mean, [G] rstd
    float* sh = sc + p.C;
    float* gm = sh + p.C;
    float* gr = gm + p.G;
    const int tid = threadIdx.x;
    const int n = blockIdx.x / p.wpi, wi = blockIdx.x - n * p.wpi;
    unsigned gen0 = 0;
    if (tid == 0) gen0 = __hip_atomic_load(p.sync + 32 * (17 + (blockIdx.x & 7)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // before this workgroup arrives
    const bool active = tid < p.rp * p.cp;
    const int r0 = tid / p.cp, j = tid - r0 * p.cp;
    const int c0 = j * 8;
    const int gA = c0 / p.Cg;
    const int eb = min(8, (gA + 1) * p.Cg - c0); // channels [0, eb) of the chunk belong to group gA, the rest to gA + 1
    const size_t row0 = (size_t)n * p.HW;
    auto src = [&](size_t row, int c) { return c < p.C0 ? p.x0 + row * p.C0 + c : p.x1 + row * p.C1 + (c - p.C0); };
    const int pix_begin = wi * p.ppw, pix_end = min(p.HW, pix_begin + p.ppw);
    // pilot shift: the image's first pixel (read by every workgroup before any of them stores: y may be x)
    const float shA = active ? (float)*src(row0, gA * p.Cg) : 0.f;
    const float shB = (active && eb < 8) ? (float)*src(row0, (gA + 1) * p.Cg) : 0.f;
    float pilot = 0.f;
    if (tid < p.G) pilot = (float)*src(row0, tid * p.Cg);
    // nloop == 1: the workgroup's pixels stay in registers between the two phases.  nloop > 1 (the VAE's 512x512 maps: more than
    // KMAX vectors per thread): the statistics phase streams them KMAX at a time and the apply phase reads them again -- from
    // the Infinity Cache, which holds the whole map -- still ONE launch instead of the statistics + apply pair.
    const int nloop = p.nloop;
    f16x8 v[KMAX];
    float a1 = 0.f, a2 = 0.f, b1 = 0.f, b2 = 0.f;
    for (int l = 0; l < nloop; ++l) {
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int pix = pix_begin + r0 + (l * KMAX + k) * p.rp;
            v[k] = (active && pix < pix_end) ? ldg8(src(row0 + pix, c0)) : zero8();
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const bool ok = active && (pix_begin + r0 + (l * KMAX + k) * p.rp) < pix_end;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool inA = e < eb;
                const float d = ok ? (float)v[k][e] - (inA ? shA : shB) : 0.f;
                a1 += inA ? d : 0.f; a2 += inA ? d * d : 0.f;
                b1 += inA ? 0.f : d; b2 += inA ? 0.f : d * d;
            }
        }
    }
    red[tid * 4 + 0] = a1; red[tid * 4 + 1] = a2; red[tid * 4 + 2] = b1; red[tid * 4 + 3] = b2;
    __syncthreads();
    if (tid < p.G) {
        const int g = tid;
        const int j0 = (g * p.Cg) >> 3, j1 = ((g + 1) * p.Cg - 1) >> 3;
        float s1 = 0.f, s2 = 0.f;
        for (int jj = j0; jj <= j1; ++jj) {
            const int first = (jj * 8) / p.Cg;      // group of the chunk's first channel
            const int slot = first == g ? 0 : 2;    // else g is the chunk's second group
            for (int r = 0; r < p.rp; ++r) {
                s1 += red[(r * p.cp + jj) * 4 + slot];
                s2 += red[(r * p.cp + jj) * 4 + slot + 1];
            }
        }
        // cross-XCD hand-off WITHOUT fences (an agent-scope release writes back the whole L2 -- the previous GEMM's output --
        // and cost ~35 us here): the payload is stored write-through (sc1: 8-byte agent-scope atomics, both sides), every storing
        // wave waits for its stores, then ONE lane signals; readers poll the generation word with sc1 loads and read the
        // payload with sc1 loads only (MI355X_MICROARCH.md, cross-workgroup hand-offs, first row of the table)
        unsigned long long bits;
        {
            const float2 pr = {s1, s2};
            __builtin_memcpy(&bits, &pr, 8);
        }
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.partial) + ((size_t)n * p.wpi + wi) * p.G + g;
        __hip_atomic_store(dst, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // ---- grid barrier.  Atomics execute at the memory side, ~45 ns apiece on one address: 256 arrivals on ONE counter cost
    // ~10 us, so arrivals go to 16 shard counters (one 128-byte line each), the last arrival of a shard reports to the top
    // counter, and the last of those re-arms everything and publishes the new generation in 8 replicas (pollers spread over
    // them; one poll per ~0.2 us so that 255 pollers do not eat the memory channel they share).  Measured alternatives
    // (tools/gn_bench.py, 64x64 x 320, whole launch): one counter 16.5 us, this tree 13.4 us, a flag per workgroup polled by
    // its readers 14.9 us (32 K polling loads per round on a handful of lines); the (image, group) kernel: 18.3 us.
    if (tid == 0) {
        const int shard = blockIdx.x & 15;
        const unsigned in_shard = (unsigned)((p.nwg - shard + 15) >> 4);
        bool opened = false;
        unsigned prev = __hip_atomic_fetch_add(p.sync + 32 * (1 + shard), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u == in_shard) {
            __hip_atomic_store(p.sync + 32 * (1 + shard), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            prev = __hip_atomic_fetch_add(p.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev + 1u == (unsigned)min(16, p.nwg)) {
                __hip_atomic_store(p.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every counter is re-armed before anybody can leave
                for (int r = 0; r < 8; ++r) __hip_atomic_store(p.sync + 32 * (17 + r), gen0 + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                opened = true;
            }
        }
        if (!opened) {
            const unsigned* flag = p.sync + 32 * (17 + (blockIdx.x & 7));
            bool met = false;
            for (int spin = 0; spin < (1 << 20); ++spin) { // bounded (~1 s): a clobbered workspace must not hang the device ...
                if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gen0) { met = true; break; }
                __builtin_amdgcn_s_sleep(6);
            }
            if (!met) {
                // ... and must not pass for a result either: the partials below are incomplete.  Sticky words, never cleared by a
                // kernel: one in the workspace (its owner re-zeroes a workspace that reports this) and the device's host-mapped
                // word, which sdod_group_norm_nhwc / sdod_group_norm_status / the graph engine read after the next host sync.
                __hip_atomic_store(p.sync + 32 * 25, 0x0bad0bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (p.err_host) __hip_atomic_store(p.err_host, 0x0bad0bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    __syncthreads();
    // ---- mean / rstd of every group of this image: 16 lanes per group, fixed order
    {
        const int g = tid >> 4, l = tid & 15;
        float a = 0.f, b = 0.f;
        if (g < p.G) {
            for (int ch = l; ch < p.wpi; ch += 16) {
                const unsigned long long bits = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p.partial) + ((size_t)n * p.wpi + ch) * p.G + g,
                                                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                float2 pr;
                __builtin_memcpy(&pr, &bits, 8);
                a += pr.x;
                b += pr.y;
            }
        }
        a = group_sum<16>(a);
        b = group_sum<16>(b);
        if (g < p.G && l == 0) {
            const float cnt = (float)p.HW * (float)p.Cg;
            const float icnt = __builtin_amdgcn_rcpf(cnt);
            const float md = a * icnt;
            float var = b * icnt - md * md;
            var = var < 0.f ? 0.f : var;
            gm[g] = md; // + pilot, added below by the thread that holds it
            gr[g] = rsqrt_fast(var + p.eps);
        }
    }
    __syncthreads();
    if (tid < p.G) gm[tid] += pilot;
    __syncthreads();
    for (int c = tid; c < p.C; c += 512) {
        const int g = c / p.Cg;
        const float wv = p.w ? p.w[c] : 1.0f, bv = p.b ? p.b[c] : 0.0f;
        sc[c] = gr[g] * wv;
        sh[c] = bv - gm[g] * gr[g] * wv;
    }
    __syncthreads();
    if (!active) return;
    float s8[8], t8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s8[e] = sc[c0 + e];
        t8[e] = sh[c0 + e];
    }
    for (int l = 0; l < nloop; ++l) {
        if (nloop > 1) { // streamed map: this thread's pixels again (its own: y == x stays safe)
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int pix = pix_begin + r0 + (l * KMAX + k) * p.rp;
                v[k] = pix < pix_end ? ldg8(src(row0 + pix, c0)) : zero8();
            }
        }
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int pix = pix_begin + r0 + (l * KMAX + k) * p.rp;
            if (pix < pix_end) {
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float f = (float)v[k][e] * s8[e] + t8[e];
                    if (p.silu) f = silu_f(f);
                    o[e] = (f16)f;
                }
                stg8(p.y + (row0 + pix) * p.C + c0, o);
            }
        }
    }
}

// Smaller maps: the (image, group) kernel is as fast or faster -- the launch has a floor of ~11 us (a chain of ~8 fabric round
// trips: first bytes, write-through partials, two counter hops, flag, poll, partial reads), tools/gn_bench.py
constexpr size_t GN_GRID_MIN_BYTES = (size_t)5 << 20;

static int gn_device_cus() { // per device, cached
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = -1;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v > 0 ? v : 0;
}

// vectors per thread the grid kernel would need (0: shape not taken)
static int gn_grid_plan(int n, int hw, int c0, int c1, int groups, GnGridP* out) {
    const int c = c0 + c1;
    if (groups > 32 || n <= 0 || !gn2_fits(c0, c1, groups) || c / 8 > 512) return 0;
    if ((size_t)n * hw * c * 2 < GN_GRID_MIN_BYTES) return 0;
    int cus = gn_device_cus();
    if (cus > 256) cus = 256;
    const int wpi = cus / n;
    if (wpi < 8 || wpi > GN_MAX_CHUNKS_GRID) return 0;
    const int cp = c / 8, rp = 512 / cp;
    const int ppw = (hw + wpi - 1) / wpi;
    const int k = (ppw + rp - 1) / rp;
    if (k > 16 * 8) return 0;
    if (out) {
        out->cp = cp; out->rp = rp; out->wpi = wpi; out->ppw = ppw; out->nwg = wpi * n;
        out->nloop = k <= 16 ? 1 : (k + 15) / 16;
    }
    return k <= 4 ? 4 : k <= 8 ? 8 : 16;
}

// Host-mapped sticky error word, one per device (the pointer is only valid on the device that allocated it).  Allocated on a
// host-side entry point that always precedes the first launch (sdod_group_norm_workspace_bytes), never inside a stream capture.
static unsigned* gn_error_word(bool allocate) {
    static std::atomic<unsigned*> words[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    unsigned* cur = words[dev].load(std::memory_order_acquire);
    if (!cur && allocate) {
        unsigned* fresh = nullptr;
        if (hipHostMalloc((void**)&fresh, 128, hipHostMallocMapped) != hipSuccess || !fresh) {
            (void)hipGetLastError();
            return nullptr;
        }
        *fresh = 0u;
        if (words[dev].compare_exchange_strong(cur, fresh, std::memory_order_acq_rel)) cur = fresh;
        else (void)hipHostFree(fresh);
    }
    return cur;
}

static bool gn_grid_try(const GnP& q, hipStream_t st) {
    GnGridP p{};
    const int kmax = gn_grid_plan(q.N, q.HW, q.C0, q.C1, q.G, &p);
    if (!kmax) return false;
    p.x0 = (const f16*)q.x0; p.x1 = (const f16*)q.x1; p.y = (f16*)q.y; p.w = q.w; p.b = q.b;
    p.partial = q.partial;
    p.sync = reinterpret_cast<unsigned*>(q.sync);
    p.err_host = gn_error_word(false);
    p.N = q.N; p.HW = q.HW; p.C0 = q.C0; p.C1 = q.C1; p.C = q.C; p.G = q.G; p.Cg = q.Cg; p.eps = q.eps; p.silu = q.silu;
    const size_t smem = ((size_t)q.C * 2 + (size_t)q.G * 2) * sizeof(float);
    if (kmax == 4) SDOD_LAUNCH((gn_grid_kernel<4>), dim3(p.nwg), dim3(512), smem, st, p);
    else if (kmax == 8) SDOD_LAUNCH((gn_grid_kernel<8>), dim3(p.nwg), dim3(512), smem, st, p);
    else SDOD_LAUNCH((gn_grid_kernel<16>), dim3(p.nwg), dim3(512), smem, st, p);
    SDOD_HIP_CHECK(hipGetLastError());
    return true;
}

static int gcd_i(int a, int b) { return b ? gcd_i(b, a % b) : a; }

// single-launch path eligibility (also what sdod_group_norm_launches reports).  Up to 256 pixels a 256-thread workgroup
// per (image, channel set) is enough; up to 1024 pixels (the 32x32 level: slab of 80 KB) it takes 1024 threads to keep
// enough loads in flight for the few workgroups there are (c = 640: 32 of them).
static size_t gn_small_smem(int hw, int sw, size_t elem) { return (size_t)hw * sw * elem + ((size_t)2 * sw + 128) * sizeof(float) + (size_t)sw; }
static bool gn_small_fits(int hw, int c, int cg, size_t elem) {
    const int sw = cg / gcd_i(cg, 8) * 8; // lcm(Cg, 8)
    if (c % sw != 0) return false;
    return hw <= 1024 && sw / cg <= 4 && gn_small_smem(hw, sw, elem) <= 96 * 1024;
}

template <typename T, int NT>
void gn_small_launch(GnP& p, int sw, size_t smem, hipStream_t st) {
    static std::atomic<unsigned long long> attr_devs{0};
    if (sdod::first_use_on_device(attr_devs)) {
        SDOD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_small_kernel<T, NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           96 * 1024 + 4096));
    }
    SDOD_LAUNCH((gn_small_kernel<T, NT>), dim3(p.C / sw, p.N), dim3(NT), smem, st, p, sw);
    SDOD_HIP_CHECK(hipGetLastError());
}

template <typename T>
bool gn_try_small(GnP& p, hipStream_t st) {
    const int sw = p.Cg / gcd_i(p.Cg, 8) * 8; // lcm(Cg, 8)
    if (!gn_small_fits(p.HW, p.C, p.Cg, sizeof(T))) return false;
    const size_t smem = gn_small_smem(p.HW, sw, sizeof(T));
    if (p.HW > 256) gn_small_launch<T, 1024>(p, sw, smem, st);
    else gn_small_launch<T, 256>(p, sw, smem, st);
    return true;
}

constexpr int GN_INLINE_CHUNKS = 128; // up to here the apply pass reduces the partials itself (two launches per GroupNorm)

// which fp16 path: the (image, group) kernel, or the lean statistics + apply pair (SDOD_GN_PATH=two: developer switch for
// tools/gn_bench.py and the kernel tests)
static int gn_path_override() {
    static const int v = [] {
        const char* e = std::getenv("SDOD_GN_PATH");
        return !e ? 0 : e[0] == 'g' ? 1 : e[0] == 't' ? 2 : 0; // "group" also keeps the big maps off the grid-barrier kernel
    }();
    return v;
}
static bool gn_prefer_pair(const GnP& p) {
    if (gn_path_override() == 1) return false;
    // Measured on MI355X (profiles/r02_gn_bench.txt): the pair is no faster than the group kernel where it was meant to win
    // (64x64 x 320 channels: 18.9 vs 18.4 us) and 2-3x slower on small maps, so it is never chosen by default.
    return gn_path_override() == 2;
}

template <typename T>
void gn_launch(GnP& p, hipStream_t st) {
    if (sizeof(T) == 2 && p.G <= 32 && gn2_fits(p.C0, p.C1, p.G) && gn_prefer_pair(p)) {
        gn2_launch(p, st);
        return;
    }
    if (sizeof(T) == 2 && gn_path_override() == 0 && gn_grid_try(p, st)) return;
    if (sizeof(T) == 2 && gn_group_try(p, nullptr, st)) return;
    if (gn_try_small<T>(p, st)) return;
    const int cp = p.C / 8;
    p.npass = (cp + 255) / 256;
    SDOD_REQUIRE(p.npass <= 2 && cp % p.npass == 0, "unsupported channel count for GroupNorm");
    p.cpp = cp / p.npass;
    p.pp = 256 / p.cpp;
    if (p.pp < 1) p.pp = 1;
    if (p.pp > p.pix_per_chunk) p.pp = p.pix_per_chunk;
    const size_t smem_stats = (size_t)p.pp * p.C * 2 * sizeof(float);
    dim3 sgrid(p.nchunks, p.N), sblock(p.cpp, p.pp);
    if (p.npass == 1)
        SDOD_LAUNCH((gn_stats_kernel<T, 1>), sgrid, sblock, smem_stats, st, p);
    else
        SDOD_LAUNCH((gn_stats_kernel<T, 2>), sgrid, sblock, smem_stats, st, p);
    SDOD_HIP_CHECK(hipGetLastError());
    if (p.nchunks > GN_INLINE_CHUNKS) {
        SDOD_LAUNCH(gn_collapse_kernel, dim3((p.G + 3) / 4, p.N), dim3(256), 0, st, p.partial, p.stats, p.nchunks, p.G);
        SDOD_HIP_CHECK(hipGetLastError());
        p.partial = p.stats; // [N][1][G][2]
        p.nchunks = 1;
    }
    const size_t total = (size_t)p.HW * cp;
    int bx = (int)((total + 255) / 256);
    const int cap = 1024 / (p.N > 0 ? p.N : 1) + 1;
    if (bx > cap) bx = cap;
    SDOD_LAUNCH((gn_apply_kernel<T>), dim3(bx, p.N), dim3(256), ((size_t)p.C * 2 + (size_t)p.G * 2) * sizeof(float), st, p);
    SDOD_HIP_CHECK(hipGetLastError());
}

constexpr int GN_MAX_CHUNKS = 1024;

// ---------------------------------------------------------------- LayerNorm: LPR lanes per row, 64 / LPR rows per wave
// A row of C = 320 .. 1280 channels is 40 .. 160 16-byte chunks: with a whole wave per row most lanes idle and every row pays
// two full-wave shuffle reductions (27 us for 18432 x 320: 0.87 TB/s).  Here LPR = 8 / 16 / 32 / 64 lanes share a row (at most
// KC = 6 chunks per lane, all loads of a row in flight at once), the reductions stay inside the lane group, and the affine
// parameters sit in LDS (8 bytes per channel, loaded once per workgroup).  Exact two-pass statistics on the registers, as before.
template <int LPR>
__global__ __launch_bounds__(256) void layer_norm_kernel(const f16* x, f16* y, const float* w, const float* b, int M,
                                                         int C, float eps) {
    constexpr int KC = 6, RPW = 64 / LPR;
    extern __shared__ __attribute__((aligned(16))) float ln_par[]; // [C] weight, [C] bias
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPR, j = lane % LPR;
    const int cp = C / 8;
    for (int i = threadIdx.x; i < C; i += 256) {
        ln_par[i] = w ? w[i] : 1.0f;
        ln_par[C + i] = b ? b[i] : 0.0f;
    }
    __syncthreads();
    const float inv_c = 1.0f / (float)C;
    const int rows_per_pass = gridDim.x * 4 * RPW;
    for (int row0 = (blockIdx.x * 4 + wave) * RPW; row0 < M; row0 += rows_per_pass) { // wave-uniform trip count (the shuffles)
        const int row = row0 + sub;
        const bool ok = row < M;
        const f16* xr = x + (size_t)(ok ? row : 0) * C;
        f16x8 v[KC];
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int ch = j + LPR * k;
            v[k] = (ok && ch < cp) ? ldg8(xr + ch * 8) : zero8();
        }
#pragma unroll
        for (int k = 0; k < KC; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)v[k][e];
        sum = group_sum<LPR>(sum);
        const float mean = sum * inv_c;
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            if (j + LPR * k < cp) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = (float)v[k][e] - mean;
                    sq += d * d;
                }
            }
        }
        sq = group_sum<LPR>(sq);
        const float rstd = rsqrt_fast(sq * inv_c + eps);
        f16* yr = y + (size_t)(ok ? row : 0) * C;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int ch = j + LPR * k;
            if (ok && ch < cp) {
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(ln_par + ch * 8), w1 = *reinterpret_cast<const f32x4*>(ln_par + ch * 8 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(ln_par + C + ch * 8), b1 = *reinterpret_cast<const f32x4*>(ln_par + C + ch * 8 + 4);
                f16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o[e] = (f16)(((float)v[k][e] - mean) * rstd * w0[e] + b0[e]);
                    o[e + 4] = (f16)(((float)v[k][e + 4] - mean) * rstd * w1[e] + b1[e]);
                }
                stg8(yr + ch * 8, o);
            }
        }
    }
}

// LayerNorm -> Linear folding (one workgroup per weight row)
__global__ __launch_bounds__(256) void ln_fold_kernel(f16* w, int N, int K, int ldw, const float* gamma, const float* beta,
                                                      const float* bias_in, float* s_out, float* t_out) {
    __shared__ float red[8];
    const int n = blockIdx.x;
    f16* row = w + (size_t)n * ldw;
    float t = 0.f, sacc = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float wv = (float)row[k];
        t += beta[k] * wv;
        const f16 folded = (f16)(wv * gamma[k]);
        row[k] = folded;
        sacc += (float)folded;
    }
    t = wave_sum(t);
    sacc = wave_sum(sacc);
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = t;
        red[4 + (threadIdx.x >> 6)] = sacc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        t_out[n] = red[0] + red[1] + red[2] + red[3] + (bias_in ? bias_in[n] : 0.f);
        s_out[n] = red[4] + red[5] + red[6] + red[7];
    }
}

// Linear o Linear composition (one-time, at graph build): c[o][k] = sum_j p[o][j] * w[j][k] in fp32, rounded to fp16 once;
// bias_out[o] = sum_j p[o][j] * bias_w[j] + bias_p[o].  One thread per output element; w rows are read coalesced along k.
__global__ __launch_bounds__(256) void compose_linear_kernel(const f16* __restrict__ pmat, int ldp, const f16* __restrict__ wmat, int ldw,
                                                             f16* __restrict__ cmat, int ldc, int n_out, int n_mid, int k,
                                                             const float* bias_w, const float* bias_p, float* bias_out) {
    const int kk = blockIdx.x * 256 + threadIdx.x;
    const int o = blockIdx.y;
    if (kk < k) {
        float acc = 0.f;
        const f16* prow = pmat + (size_t)o * ldp;
        for (int j = 0; j < n_mid; ++j) acc += (float)prow[j] * (float)wmat[(size_t)j * ldw + kk];
        cmat[(size_t)o * ldc + kk] = (f16)acc;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && bias_out) {
        float b = bias_p ? bias_p[o] : 0.f;
        if (bias_w) {
            const f16* prow = pmat + (size_t)o * ldp;
            for (int j = 0; j < n_mid; ++j) b += (float)prow[j] * bias_w[j];
        }
        bias_out[o] = b;
    }
}


// ------------------------------------------------------------------------------------------------
// GroupNorm over NCHW ("[N][C][spatial]", torch's default layout) -- the layout `sdod.EfficientGN` receives from a model that
// was not converted to channels_last (efficient_gn.py:61-86 takes whatever the UNet hands it).  In this layout a group is ONE
// contiguous slab of L = (C / G) * spatial elements, so the kernel needs no transpose (the NHWC kernels above would pay a
// torch .contiguous() round trip first: twice the operator's algorithmic bytes) and no divisibility of C: any channel count,
// fp16 / bf16 / fp32.
//   * slabs up to 32 Ki elements: ONE launch, a 512-thread workgroup per (image, group) holds its slab in registers (read
//     once), exact two-pass mean / variance by block reductions, normalise (+ SiLU) + store;
//   * bigger slabs: statistics launch (grid: slabs x chunks, shifted sums, one partial per workgroup -- no atomics, fixed
//     reduction order) + apply launch (every workgroup re-reduces the <= 64 partials of its slab).
struct GnNchwP {
    const void* x;
    void* y;
    const float* w;
    const float* b;
    float* partial; // [slabs][chunks][2], big-slab path only
    int G, Cg, S;   // groups, channels per group, spatial size
    long long L;    // Cg * S
    int chunks;
    long long per_chunk;
    float eps;
    int silu;
};

template <typename T> struct NchwIo;
template <> struct NchwIo<f16> {
    static SDOD_DEVICE float ld(const void* p, long long i) { return (float)reinterpret_cast<const f16*>(p)[i]; }
    static SDOD_DEVICE void st(void* p, long long i, float v) { reinterpret_cast<f16*>(p)[i] = (f16)v; }
    static SDOD_DEVICE void ld8(const void* p, long long i, float (&v)[8]) {
        const f16x8 h = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16*>(p) + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
    }
    static SDOD_DEVICE void st8(void* p, long long i, const float (&v)[8]) {
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = (f16)v[e];
        *reinterpret_cast<f16x8*>(reinterpret_cast<f16*>(p) + i) = h;
    }
};
template <> struct NchwIo<float> {
    static SDOD_DEVICE float ld(const void* p, long long i) { return reinterpret_cast<const float*>(p)[i]; }
    static SDOD_DEVICE void st(void* p, long long i, float v) { reinterpret_cast<float*>(p)[i] = v; }
    static SDOD_DEVICE void ld8(const void* p, long long i, float (&v)[8]) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + i);
        const f32x4 b = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p) + i + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    }
    static SDOD_DEVICE void st8(void* p, long long i, const float (&v)[8]) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p) + i) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p) + i + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
};
struct bf16_tag {};
SDOD_DEVICE float bf16_to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
SDOD_DEVICE uint16_t f32_to_bf16(float v) {
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    // round to nearest even; a NaN stays a NaN (the integer form alone would turn some NaNs into zero or infinity)
    return (v != v) ? (uint16_t)0x7FC0 : (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
template <> struct NchwIo<bf16_tag> {
    static SDOD_DEVICE float ld(const void* p, long long i) { return bf16_to_f32(reinterpret_cast<const uint16_t*>(p)[i]); }
    static SDOD_DEVICE void st(void* p, long long i, float v) { reinterpret_cast<uint16_t*>(p)[i] = f32_to_bf16(v); }
    static SDOD_DEVICE void ld8(const void* p, long long i, float (&v)[8]) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p) + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __builtin_bit_cast(float, w[e] << 16);
            v[2 * e + 1] = __builtin_bit_cast(float, w[e] & 0xFFFF0000u);
        }
    }
    static SDOD_DEVICE void st8(void* p, long long i, const float (&v)[8]) {
        u32x4 w;
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (uint32_t)f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16);
        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p) + i) = w;
    }
};

SDOD_DEVICE float block_sum_512(float v, float* red) { // red: >= 8 floats of LDS; every thread gets the total
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

// affine + SiLU + store of 8 consecutive slab elements starting at slab index i (the channel changes every S elements)
template <typename T>
SDOD_DEVICE void gn_nchw_finish8(const GnNchwP& p, long long base, long long i, int g, float (&v)[8], float mean, float rstd) {
    int c = g * p.Cg + (int)(i / p.S);
    int left = p.S - (int)(i % p.S); // elements of channel c from i on
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (left == 0) { ++c; left = p.S; }
        --left;
        float f = (v[e] - mean) * rstd;
        if (p.w) f = f * p.w[c] + p.b[c];
        if (p.silu) f = silu_f(f);
        v[e] = f;
    }
    NchwIo<T>::st8(p.y, base + i, v);
}

// VEC = the slab is a whole number of 16-byte (fp32: 32-byte) vectors and so aligned: 8 elements per thread and step
template <typename T, int KMAX, bool VEC>
__global__ __launch_bounds__(512) void gn_nchw_one_kernel(const GnNchwP p) {
    __shared__ float red[8];
    const long long slab = blockIdx.x; // n * G + g
    const int g = (int)(slab % p.G);
    const long long base = slab * p.L;
    constexpr int W = VEC ? 8 : 1;
    float v[KMAX][W];
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const long long i = ((long long)k * 512 + threadIdx.x) * W;
        if constexpr (VEC) {
            if (i < p.L) NchwIo<T>::ld8(p.x, base + i, v[k]);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) v[k][e] = 0.f;
        } else {
            v[k][0] = i < p.L ? NchwIo<T>::ld(p.x, base + i) : 0.f;
        }
#pragma unroll
        for (int e = 0; e < W; ++e) sum += v[k][e];
    }
    const float mean = block_sum_512(sum, red) / (float)p.L;
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const long long i = ((long long)k * 512 + threadIdx.x) * W;
        if (i < p.L) {
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const float d = v[k][e] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrt_fast(block_sum_512(sq, red) / (float)p.L + p.eps);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const long long i = ((long long)k * 512 + threadIdx.x) * W;
        if (i < p.L) {
            if constexpr (VEC) {
                gn_nchw_finish8<T>(p, base, i, g, v[k], mean, rstd);
            } else {
                const int c = g * p.Cg + (int)(i / p.S);
                float f = (v[k][0] - mean) * rstd;
                if (p.w) f = f * p.w[c] + p.b[c];
                if (p.silu) f = silu_f(f);
                NchwIo<T>::st(p.y, base + i, f);
            }
        }
    }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(512) void gn_nchw_stats_kernel(const GnNchwP p) {
    __shared__ float red[8];
    const long long slab = blockIdx.y;
    const long long base = slab * p.L;
    const long long i0 = (long long)blockIdx.x * p.per_chunk, i1 = min(p.L, i0 + p.per_chunk);
    const float shift = NchwIo<T>::ld(p.x, base); // pilot: the slab's first element (shifted sums keep fp32 accurate)
    float s1 = 0.f, s2 = 0.f;
    if constexpr (VEC) { // per_chunk is a multiple of 8
        for (long long i = i0 + (long long)threadIdx.x * 8; i < i1; i += 512 * 8) {
            float v[8];
            NchwIo<T>::ld8(p.x, base + i, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = v[e] - shift;
                s1 += d;
                s2 += d * d;
            }
        }
    } else {
        for (long long i = i0 + threadIdx.x; i < i1; i += 512) {
            const float d = NchwIo<T>::ld(p.x, base + i) - shift;
            s1 += d;
            s2 += d * d;
        }
    }
    s1 = block_sum_512(s1, red);
    s2 = block_sum_512(s2, red);
    if (threadIdx.x == 0) {
        float* dst = p.partial + (slab * p.chunks + blockIdx.x) * 2;
        dst[0] = s1;
        dst[1] = s2;
    }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(512) void gn_nchw_apply_kernel(const GnNchwP p) {
    const long long slab = blockIdx.y;
    const int g = (int)(slab % p.G);
    const long long base = slab * p.L;
    float s1 = 0.f, s2 = 0.f;
    for (int c = 0; c < p.chunks; ++c) { // every thread, same order: bit-reproducible
        const float* src = p.partial + (slab * p.chunks + c) * 2;
        s1 += src[0];
        s2 += src[1];
    }
    const float shift = NchwIo<T>::ld(p.x, base);
    const float md = s1 / (float)p.L;
    float var = s2 / (float)p.L - md * md;
    var = var < 0.f ? 0.f : var;
    const float mean = shift + md, rstd = rsqrt_fast(var + p.eps);
    const long long i0 = (long long)blockIdx.x * p.per_chunk, i1 = min(p.L, i0 + p.per_chunk);
    if constexpr (VEC) {
        for (long long i = i0 + (long long)threadIdx.x * 8; i < i1; i += 512 * 8) {
            float v[8];
            NchwIo<T>::ld8(p.x, base + i, v);
            gn_nchw_finish8<T>(p, base, i, g, v, mean, rstd);
        }
    } else {
        for (long long i = i0 + threadIdx.x; i < i1; i += 512) {
            const int c = g * p.Cg + (int)(i / p.S);
            float f = (NchwIo<T>::ld(p.x, base + i) - mean) * rstd;
            if (p.w) f = f * p.w[c] + p.b[c];
            if (p.silu) f = silu_f(f);
            NchwIo<T>::st(p.y, base + i, f);
        }
    }
}

template <typename T, bool VEC>
static void gn_nchw_launch_v(const GnNchwP& p, long long slabs, hipStream_t st) {
    constexpr int W = VEC ? 8 : 1;
    if (p.L <= 512 * 64) {
        const int k = (int)((p.L + 512 * W - 1) / (512 * W));
        if (k <= 1) SDOD_LAUNCH((gn_nchw_one_kernel<T, 1, VEC>), dim3((unsigned)slabs), dim3(512), 0, st, p);
        else if (k <= 2) SDOD_LAUNCH((gn_nchw_one_kernel<T, 2, VEC>), dim3((unsigned)slabs), dim3(512), 0, st, p);
        else if (k <= 4) SDOD_LAUNCH((gn_nchw_one_kernel<T, 4, VEC>), dim3((unsigned)slabs), dim3(512), 0, st, p);
        else if (k <= 8) SDOD_LAUNCH((gn_nchw_one_kernel<T, 8, VEC>), dim3((unsigned)slabs), dim3(512), 0, st, p);
        else if constexpr (!VEC) {
            if (k <= 16) SDOD_LAUNCH((gn_nchw_one_kernel<T, 16, VEC>), dim3((unsigned)slabs), dim3(512), 0, st, p);
            else if (k <= 32) SDOD_LAUNCH((gn_nchw_one_kernel<T, 32, VEC>), dim3((unsigned)slabs), dim3(512), 0, st, p);
            else SDOD_LAUNCH((gn_nchw_one_kernel<T, 64, VEC>), dim3((unsigned)slabs), dim3(512), 0, st, p);
        }
        SDOD_HIP_CHECK(hipGetLastError());
        return;
    }
    SDOD_LAUNCH((gn_nchw_stats_kernel<T, VEC>), dim3(p.chunks, (unsigned)slabs), dim3(512), 0, st, p);
    SDOD_HIP_CHECK(hipGetLastError());
    SDOD_LAUNCH((gn_nchw_apply_kernel<T, VEC>), dim3(p.chunks, (unsigned)slabs), dim3(512), 0, st, p);
    SDOD_HIP_CHECK(hipGetLastError());
}

template <typename T>
static void gn_nchw_launch(const GnNchwP& p, long long slabs, bool vec, hipStream_t st) {
    if (vec) gn_nchw_launch_v<T, true>(p, slabs, st);
    else gn_nchw_launch_v<T, false>(p, slabs, st);
}

} // namespace

extern "C" int sdod_compose_linear_f16(const void* p, int ldp, const void* w, int ldw, void* c, int ldc, int n_out, int n_mid, int k,
                                       const float* bias_w, const float* bias_p, float* bias_out, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(p && w && c && n_out > 0 && n_mid > 0 && k > 0 && ldp >= n_mid && ldw >= k && ldc >= k, "bad argument");
    SDOD_REQUIRE(c != w && c != p, "the composed matrix must not alias its factors");
    SDOD_LAUNCH(compose_linear_kernel, dim3((k + 255) / 256, n_out), dim3(256), 0, (hipStream_t)stream, (const f16*)p, ldp, (const f16*)w, ldw,
                (f16*)c, ldc, n_out, n_mid, k, bias_w, bias_p, bias_out);
    SDOD_HIP_CHECK(hipGetLastError());
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_ln_fold_f16(void* w, int n, int k, int ldw, const float* gamma, const float* beta, const float* bias_in,
                                float* s_out, float* t_out, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(w && gamma && beta && s_out && t_out && n > 0 && k > 0 && ldw >= k, "bad argument");
    SDOD_LAUNCH(ln_fold_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (f16*)w, n, k, ldw, gamma, beta, bias_in, s_out, t_out);
    SDOD_HIP_CHECK(hipGetLastError());
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_group_norm_launches(int hw, int c, int groups, int dtype) {
    if (hw <= 0 || c <= 0 || groups <= 0 || c % groups) return 0;
    if (dtype == SDOD_F16 && groups <= 32 && gn2_fits(c, 0, groups) && gn_path_override() == 2) return 2;
    if (dtype == SDOD_F16 && c % 8 == 0 && gn_group_plan(hw, c, 0, c / groups, false).v && gn_path_override() != 2) return 1;
    if (dtype == SDOD_F16 && gn_path_override() == 0 && gn_grid_plan(2, hw, c, 0, groups, nullptr)) return 1;
    return gn_small_fits(hw, c, c / groups, dtype == SDOD_F16 ? 2 : 4) ? 1 : 2; // (+1 collapse launch on very large maps)
}

// which kernel sdod_group_norm_nhwc launches for this shape: 0 = the one-launch grid-barrier kernel (big maps), 1 = the (image,
// group) one-launch kernel, 2 = the small-map LDS kernel, 3 = statistics + apply (two or three launches)
extern "C" int sdod_group_norm_path(int n, int hw, int c0, int c1, int groups, int dtype) {
    const int c = c0 + c1;
    if (n <= 0 || hw <= 0 || c <= 0 || groups <= 0 || c % groups) return -1;
    if (dtype == SDOD_F16 && groups <= 32 && gn2_fits(c0, c1, groups) && gn_path_override() == 2) return 3;
    if (dtype == SDOD_F16 && gn_path_override() == 0 && gn_grid_plan(n, hw, c0, c1, groups, nullptr)) return 0;
    if (dtype == SDOD_F16 && c0 % 8 == 0 && c1 % 8 == 0 && gn_group_plan(hw, c0, c1, c / groups, false).v && gn_path_override() != 2) return 1;
    return gn_small_fits(hw, c, c / groups, dtype == SDOD_F16 ? 2 : 4) ? 2 : 3;
}

extern "C" int sdod_group_norm_reduce_ok(int hw, int c0, int c1, int groups) {
    if (hw <= 0 || c0 <= 0 || c1 < 0 || groups <= 0 || (c0 + c1) % groups || c0 % 8 || c1 % 8) return 0;
    return gn_group_plan(hw, c0, c1, (c0 + c1) / groups, true).v ? 1 : 0;
}

extern "C" int sdod_group_norm_reduce_nhwc(const sdod_gn_reduce* red, const void* x2, void* y, const float* weight,
                                           const float* bias, int n, int hw, int c0, int c1, int groups, float eps, int silu,
                                           void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(red && y && red->partial && red->x_out && red->splits > 1, "null pointer / not a split-K descriptor");
    SDOD_REQUIRE(n > 0 && hw > 0 && c0 > 0 && c1 >= 0 && groups > 0 && (c0 + c1) % groups == 0, "bad shape");
    SDOD_REQUIRE(c1 == 0 || x2 != nullptr, "c1 > 0 needs x2");
    SDOD_REQUIRE((weight == nullptr) == (bias == nullptr), "weight and bias must both be given or both be null");
    SDOD_REQUIRE(red->M == n * hw && red->N == c0 && red->slab_floats == (size_t)red->M * red->N, "reduce descriptor does not match the GroupNorm shape");
    SDOD_REQUIRE(!red->residual || red->ldr % 8 == 0, "residual row stride must keep 16-byte alignment");
    SDOD_REQUIRE(!red->row_bias || red->ld_row_bias >= c0, "row_bias stride < c0");
    SDOD_REQUIRE(sdod_group_norm_reduce_ok(hw, c0, c1, groups), "shape not eligible for the fused reduce (run the GEMM's phase 2 instead)");
    GnP p{};
    p.x0 = red->x_out; p.x1 = x2; p.y = y; p.w = weight; p.b = bias;
    p.N = n; p.HW = hw; p.C0 = c0; p.C1 = c1; p.C = c0 + c1; p.G = groups; p.Cg = (c0 + c1) / groups;
    p.eps = eps; p.silu = silu;
    GnRed r{};
    r.partial = (const float*)red->partial; r.splits = red->splits; r.slab = red->slab_floats;
    r.bias = (const float*)red->bias; r.bias2 = (const float*)red->bias2;
    r.row_bias = (const f16*)red->row_bias; r.ldrb = red->ld_row_bias;
    r.residual = (const f16*)red->residual; r.ldr = red->ldr;
    r.xout = (f16*)red->x_out; r.alpha = red->alpha; r.act = red->act;
    SDOD_REQUIRE(gn_group_try(p, &r, (hipStream_t)stream), "shape not eligible");
    return 0;
    SDOD_CATCH
}

// Workspace layout (floats): [0, GN_SYNC_FLOATS) the grid barrier's 25 counter lines + the sticky timeout word (line 25), at a
// FIXED offset in front of everything else, so that no (n, groups) layout of the partials of one call can reach the barrier
// words another call's layout uses (one grow-only workspace serves every shape of a process); then partial
// [n][GN_MAX_CHUNKS][groups][2], stats [n][groups][2], shift [n][groups].
constexpr size_t GN_SYNC_FLOATS = 1024;
extern "C" size_t sdod_group_norm_workspace_bytes(int n, int groups) {
    if (n <= 0 || groups <= 0) return 0;
    (void)gn_error_word(true);
    return (GN_SYNC_FLOATS + (size_t)n * GN_MAX_CHUNKS * groups * 2 + (size_t)n * groups * 3) * sizeof(float);
}

extern "C" int sdod_group_norm_layout(int n, int groups, size_t* sync_off, size_t* partial_off, size_t* stats_off, size_t* shift_off,
                                      size_t* end_off) {
    if (n <= 0 || groups <= 0) return sdod::INVALID_ARGUMENT;
    const size_t part = GN_SYNC_FLOATS * sizeof(float), st = part + (size_t)n * GN_MAX_CHUNKS * groups * 2 * sizeof(float);
    const size_t sh = st + (size_t)n * groups * 2 * sizeof(float);
    if (sync_off) *sync_off = 0;
    if (partial_off) *partial_off = part;
    if (stats_off) *stats_off = st;
    if (shift_off) *shift_off = sh;
    if (end_off) *end_off = sh + (size_t)n * groups * sizeof(float);
    return 0;
}

extern "C" int sdod_group_norm_status(void) {
    const unsigned* w = gn_error_word(false);
    return (w && *reinterpret_cast<const volatile unsigned*>(w) != 0u) ? sdod::RUNTIME_ERROR : 0;
}

extern "C" int sdod_group_norm_clear_error(void) {
    unsigned* w = gn_error_word(false);
    if (w) *reinterpret_cast<volatile unsigned*>(w) = 0u;
    return 0;
}

extern "C" int sdod_group_norm_nhwc(const void* x, const void* x2, void* y, const float* weight, const float* bias, int n,
                                    int hw, int c0, int c1, int groups, float eps, int silu, int dtype, void* workspace,
                                    void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && y && workspace, "null pointer");
    SDOD_REQUIRE(n > 0 && hw > 0 && c0 > 0 && c1 >= 0 && groups > 0, "bad shape");
    SDOD_REQUIRE(c1 == 0 || x2 != nullptr, "c1 > 0 needs x2");
    const int c = c0 + c1;
    SDOD_REQUIRE(c % groups == 0, "num_channels must be divisible by num_groups");
    SDOD_REQUIRE(c0 % 8 == 0 && c1 % 8 == 0, "channel counts must be multiples of 8");
    SDOD_REQUIRE((weight == nullptr) == (bias == nullptr), "weight and bias must both be given or both be null");
    SDOD_REQUIRE(dtype == SDOD_F16 || dtype == SDOD_F32, "dtype must be SDOD_F16 or SDOD_F32");
    GnP p{};
    p.x0 = x; p.x1 = x2; p.y = y; p.w = weight; p.b = bias;
    p.N = n; p.HW = hw; p.C0 = c0; p.C1 = c1; p.C = c; p.G = groups; p.Cg = c / groups;
    p.eps = eps; p.silu = silu;
    // enough statistics workgroups to cover the chip: ~512 in total, more (>= 32 KB of input each) for the VAE's big maps
    int nchunks = (512 + n - 1) / n;
    const size_t bytes_per_img = (size_t)hw * c * (dtype == SDOD_F16 ? 2 : 4);
    const int by_size = (int)std::min<size_t>(GN_MAX_CHUNKS, bytes_per_img / (64 * 1024));
    if (nchunks > GN_INLINE_CHUNKS) nchunks = GN_INLINE_CHUNKS;
    if (by_size > nchunks && n * nchunks < 512) nchunks = std::min(by_size, (1024 + n - 1) / n);
    if (nchunks > GN_MAX_CHUNKS) nchunks = GN_MAX_CHUNKS;
    if (nchunks > hw) nchunks = hw;
    p.pix_per_chunk = (hw + nchunks - 1) / nchunks;
    p.nchunks = (hw + p.pix_per_chunk - 1) / p.pix_per_chunk;
    SDOD_REQUIRE(((uintptr_t)workspace & 127) == 0, "workspace must be 128-byte aligned");
    if (sdod_group_norm_status() != 0)
        throw sdod::Error(sdod::RUNTIME_ERROR, std::string(__func__) +
                          ": an earlier one-launch GroupNorm timed out at its grid barrier (workspace clobbered, or shared by concurrent "
                          "launches): its output is invalid; re-zero the workspace and call sdod_group_norm_clear_error()");
    p.sync = workspace; // 26 lines of 128 bytes at the fixed front of the workspace (sdod_group_norm_workspace_bytes)
    p.partial = (float*)workspace + GN_SYNC_FLOATS;
    p.stats = p.partial + (size_t)n * GN_MAX_CHUNKS * groups * 2;
    p.shift = p.stats + (size_t)n * groups * 2;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SDOD_F16) gn_launch<f16>(p, st);
    else gn_launch<float>(p, st);
    return 0;
    SDOD_CATCH
}

extern "C" size_t sdod_group_norm_nchw_workspace_bytes(int n, int groups) {
    if (n <= 0 || groups <= 0) return 0;
    return (size_t)n * groups * 64 * 2 * sizeof(float);
}

extern "C" int sdod_group_norm_nchw(const void* x, void* y, const float* weight, const float* bias, int n, int c, long long spatial,
                                    int groups, float eps, int silu, int dtype, void* workspace, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && y, "null pointer");
    SDOD_REQUIRE(n > 0 && c > 0 && spatial > 0 && groups > 0, "bad shape");
    SDOD_REQUIRE(c % groups == 0, "num_channels must be divisible by num_groups");
    SDOD_REQUIRE((weight == nullptr) == (bias == nullptr), "weight and bias must both be given or both be null");
    SDOD_REQUIRE(dtype == SDOD_F16 || dtype == SDOD_F32 || dtype == SDOD_BF16, "dtype must be SDOD_F16, SDOD_F32 or SDOD_BF16");
    SDOD_REQUIRE((long long)n * groups < (1ll << 31) / 64, "too many (image, group) slabs");
    GnNchwP p{};
    p.x = x; p.y = y; p.w = weight; p.b = bias;
    p.G = groups; p.Cg = c / groups;
    SDOD_REQUIRE(spatial < (1ll << 31), "spatial size must fit 31 bits");
    p.S = (int)spatial;
    p.L = (long long)p.Cg * spatial;
    p.eps = eps; p.silu = silu;
    const long long slabs = (long long)n * groups;
    if (p.L > 512 * 64) {
        SDOD_REQUIRE(workspace != nullptr, "slabs above 32 Ki elements need the workspace (sdod_group_norm_nchw_workspace_bytes)");
        // enough workgroups to cover the chip (~1024), at least 16 Ki elements each, at most 64 chunks per slab
        long long chunks = std::min<long long>(64, std::max<long long>(1, (1024 + slabs - 1) / slabs));
        chunks = std::min<long long>(chunks, std::max<long long>(1, p.L / 16384));
        p.per_chunk = ((p.L + chunks - 1) / chunks + 7) / 8 * 8; // whole vectors per chunk
        p.chunks = (int)((p.L + p.per_chunk - 1) / p.per_chunk);
        p.partial = (float*)workspace;
    }
    // whole, aligned 8-element vectors: every slab then starts on a 16-byte (fp32: 32-byte) boundary
    const size_t elem = dtype == SDOD_F32 ? 4 : 2;
    const bool vec = p.L % 8 == 0 && ((uintptr_t)x % (8 * elem)) == 0 && ((uintptr_t)y % (8 * elem)) == 0;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SDOD_F16) gn_nchw_launch<f16>(p, slabs, vec, st);
    else if (dtype == SDOD_F32) gn_nchw_launch<float>(p, slabs, vec, st);
    else gn_nchw_launch<bf16_tag>(p, slabs, vec, st);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_layer_norm_f16(const void* x, void* y, const float* weight, const float* bias, int m, int c, float eps,
                                   void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && y, "null pointer");
    SDOD_REQUIRE(m > 0 && c > 0 && c % 8 == 0 && c <= 3072, "LayerNorm needs C % 8 == 0 and C <= 3072");
    const int cp = c / 8;
    const int lpr = cp <= 48 ? 8 : cp <= 96 ? 16 : cp <= 192 ? 32 : 64; // lanes per row: at most 6 chunks per lane
    const int rows_per_wg = 4 * (64 / lpr);
    const int grid = std::min((m + rows_per_wg - 1) / rows_per_wg, 2048);
    const size_t lds = (size_t)2 * c * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const f16* xs = (const f16*)x;
    f16* yd = (f16*)y;
    switch (lpr) {
    case 8: SDOD_LAUNCH(layer_norm_kernel<8>, dim3(grid), dim3(256), lds, st, xs, yd, weight, bias, m, c, eps); break;
    case 16: SDOD_LAUNCH(layer_norm_kernel<16>, dim3(grid), dim3(256), lds, st, xs, yd, weight, bias, m, c, eps); break;
    case 32: SDOD_LAUNCH(layer_norm_kernel<32>, dim3(grid), dim3(256), lds, st, xs, yd, weight, bias, m, c, eps); break;
    default: SDOD_LAUNCH(layer_norm_kernel<64>, dim3(grid), dim3(256), lds, st, xs, yd, weight, bias, m, c, eps); break;
    }
    SDOD_HIP_CHECK(hipGetLastError());
    return 0;
    SDOD_CATCH
}

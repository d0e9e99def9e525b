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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
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
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                a += __shfl_xor(a, o);
                b += __shfl_xor(b, o);
            }
            if (lane == 0) {
                const float cnt = (float)p.HW * (float)p.Cg;
                const float shift = (float)*gn_src<T>(p, n, 0, g * p.Cg);
                const float md = a / cnt;
                float var = b / cnt - md * md;
                var = var < 0.f ? 0.f : var;
                gm[g] = shift + md;
                gr[g] = 1.0f / sqrtf(var + p.eps);
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
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1[gi] += __shfl_xor(s1[gi], o);
            s2[gi] += __shfl_xor(s2[gi], o);
        }
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
        const float rstd = 1.0f / sqrtf(var + p.eps);
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
    static bool attr_set = false;
    if (!attr_set) {
        SDOD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_small_kernel<T, NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           96 * 1024 + 4096));
        attr_set = true;
    }
    hipLaunchKernelGGL((gn_small_kernel<T, NT>), dim3(p.C / sw, p.N), dim3(NT), smem, st, p, sw);
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

template <typename T>
void gn_launch(GnP& p, hipStream_t st) {
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
        hipLaunchKernelGGL((gn_stats_kernel<T, 1>), sgrid, sblock, smem_stats, st, p);
    else
        hipLaunchKernelGGL((gn_stats_kernel<T, 2>), sgrid, sblock, smem_stats, st, p);
    SDOD_HIP_CHECK(hipGetLastError());
    if (p.nchunks > GN_INLINE_CHUNKS) {
        hipLaunchKernelGGL(gn_collapse_kernel, dim3((p.G + 3) / 4, p.N), dim3(256), 0, st, p.partial, p.stats, p.nchunks, p.G);
        SDOD_HIP_CHECK(hipGetLastError());
        p.partial = p.stats; // [N][1][G][2]
        p.nchunks = 1;
    }
    const size_t total = (size_t)p.HW * cp;
    int bx = (int)((total + 255) / 256);
    const int cap = 1024 / (p.N > 0 ? p.N : 1) + 1;
    if (bx > cap) bx = cap;
    hipLaunchKernelGGL((gn_apply_kernel<T>), dim3(bx, p.N), dim3(256), ((size_t)p.C * 2 + (size_t)p.G * 2) * sizeof(float), st, p);
    SDOD_HIP_CHECK(hipGetLastError());
}

constexpr int GN_MAX_CHUNKS = 1024;

// ---------------------------------------------------------------- LayerNorm: one wave per row
__global__ __launch_bounds__(256) void layer_norm_kernel(const f16* x, f16* y, const float* w, const float* b, int M,
                                                         int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int cp = C / 8;
    const f16* xr = x + (size_t)row * C;
    f16x8 v[4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = lane + 64 * i;
        if (ch < cp) {
            v[i] = ldg8(xr + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += (float)v[i][e];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = lane + 64 * i;
        if (ch < cp) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (float)v[i][e] - mean;
                sq += d * d;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)C + eps);
    f16* yr = y + (size_t)row * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = lane + 64 * i;
        if (ch < cp) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = ch * 8 + e;
                const float ww = w ? w[c] : 1.0f, bb = b ? b[c] : 0.0f;
                o[e] = (f16)(((float)v[i][e] - mean) * rstd * ww + bb);
            }
            stg8(yr + ch * 8, o);
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        t += __shfl_xor(t, o);
        sacc += __shfl_xor(sacc, o);
    }
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

} // namespace

extern "C" int sdod_ln_fold_f16(void* w, int n, int k, int ldw, const float* gamma, const float* beta, const float* bias_in,
                                float* s_out, float* t_out, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(w && gamma && beta && s_out && t_out && n > 0 && k > 0 && ldw >= k, "bad argument");
    hipLaunchKernelGGL(ln_fold_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (f16*)w, n, k, ldw, gamma, beta, bias_in, s_out, t_out);
    SDOD_HIP_CHECK(hipGetLastError());
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_group_norm_launches(int hw, int c, int groups, int dtype) {
    if (hw <= 0 || c <= 0 || groups <= 0 || c % groups) return 0;
    return gn_small_fits(hw, c, c / groups, dtype == SDOD_F16 ? 2 : 4) ? 1 : 2; // (+1 collapse launch on very large maps)
}

extern "C" size_t sdod_group_norm_workspace_bytes(int n, int groups) {
    if (n <= 0 || groups <= 0) return 0;
    return ((size_t)n * GN_MAX_CHUNKS * groups * 2 + (size_t)n * groups * 2) * sizeof(float);
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
    p.partial = (float*)workspace;
    p.stats = p.partial + (size_t)n * GN_MAX_CHUNKS * groups * 2;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == SDOD_F16) gn_launch<f16>(p, st);
    else gn_launch<float>(p, st);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_layer_norm_f16(const void* x, void* y, const float* weight, const float* bias, int m, int c, float eps,
                                   void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && y, "null pointer");
    SDOD_REQUIRE(m > 0 && c > 0 && c % 8 == 0 && c <= 2048, "LayerNorm needs C % 8 == 0 and C <= 2048");
    hipLaunchKernelGGL(layer_norm_kernel, dim3((m + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y,
                       weight, bias, m, c, eps);
    SDOD_HIP_CHECK(hipGetLastError());
    return 0;
    SDOD_CATCH
}

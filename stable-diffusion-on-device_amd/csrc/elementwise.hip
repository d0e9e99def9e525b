// elementwise.hip -- HBM-bound glue kernels of the txt2img path (gfx950).  All fp16 traffic moves as
// 16-byte lanes (8 halves), grid-stride over <= 2048 workgroups.  The sampler-side kernels restate, on
// device and with IEEE (non-contracted) fp32 arithmetic, what the reference does on the host:
//   cfg_combine         context.cpp:359-373 (+ qnn_context.cpp:1065-1081 simple_cast<Accum,Scale>)
//   dpm_update          dpm_solver.cpp:136-181
//   timestep_features   context.cpp:257-274
//   image_to_u8         context.cpp:392-395
// and the PLMS/DDIM arithmetic of config 1's CPU reference (ldm PLMSSampler; not in /root/reference).
#include "common.h"
#include <atomic>
#include "sdod_hip.h"
#include "host_util.h"

// The sampler kernels must reproduce the host's fp32 arithmetic bit for bit: hipcc contracts a*b+c into an
// FMA by default (HIP's __fmul_rn/__fadd_rn are header inlines compiled with contraction allowed, so they do not prevent it).
#pragma clang fp contract(off)

namespace {

// defined here (under contract(off)) so that no `contract` flag rides along when they are inlined
SDOD_DEVICE float mul_rn(float a, float b) { return a * b; }
SDOD_DEVICE float add_rn(float a, float b) { return a + b; }
SDOD_DEVICE float sub_rn(float a, float b) { return a - b; }
SDOD_DEVICE float div_rn(float a, float b) { return a / b; }

inline int grid_for(size_t work, int block = 256) {
    size_t b = (work + block - 1) / block;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

#define GRID_STRIDE(i, n) \
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (size_t)gridDim.x * blockDim.x)

__global__ void geglu_kernel(const f16* x, f16* y, int M, int C) {
    const int cp = C / 8;
    const size_t total = (size_t)M * cp;
    GRID_STRIDE(i, total) {
        const size_t m = i / cp;
        const int c0 = (int)(i - m * cp) * 8;
        const f16x8 a = ldg8(x + m * 2 * C + c0);
        const f16x8 gt = ldg8(x + m * 2 * C + C + c0);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((float)a[e] * gelu_erf_f((float)gt[e]));
        stg8(y + m * C + c0, o);
    }
}

__global__ void act_kernel(const f16* x, f16* y, size_t n8, int act) {
    GRID_STRIDE(i, n8) {
        const f16x8 a = ldg8(x + i * 8);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)apply_act((float)a[e], act);
        stg8(y + i * 8, o);
    }
}

__global__ void add_kernel(const f16* a, const f16* b, f16* y, size_t n8) {
    GRID_STRIDE(i, n8) {
        const f16x8 u = ldg8(a + i * 8), v = ldg8(b + i * 8);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((float)u[e] + (float)v[e]);
        stg8(y + i * 8, o);
    }
}

__global__ void concat_kernel(const f16* a, const f16* b, f16* y, size_t rows, int c0, int c1) {
    const int cp = (c0 + c1) / 8, cp0 = c0 / 8;
    const size_t total = rows * cp;
    GRID_STRIDE(i, total) {
        const size_t r = i / cp;
        const int ch = (int)(i - r * cp);
        const f16x8 v = ch < cp0 ? ldg8(a + r * c0 + ch * 8) : ldg8(b + r * c1 + (ch - cp0) * 8);
        stg8(y + r * (c0 + c1) + ch * 8, v);
    }
}

// 3x3 pad-1 im2col for tiny Cin (the 4-channel latent): y[row][k], k=(r*3+s)*c+ch, zero-padded to kpad
__global__ void im2col_small_kernel(const f16* x, f16* y, int n_img, int h, int w, int c, int kpad) {
    const size_t total = (size_t)n_img * h * w * kpad;
    GRID_STRIDE(i, total) {
        const size_t row = i / kpad;
        const int k = (int)(i - row * kpad);
        f16 v = (f16)0.f;
        if (k < 9 * c) {
            const int tap = k / c, ch = k - tap * c;
            const int r = tap / 3, s = tap - r * 3;
            const int img = (int)(row / ((size_t)h * w));
            const int rem = (int)(row - (size_t)img * h * w);
            const int oy = rem / w, ox = rem - oy * w;
            const int yy = oy + r - 1, xx = ox + s - 1;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) v = x[(((size_t)img * h + yy) * w + xx) * c + ch];
        }
        y[i] = v;
    }
}

// latent_prep (NCHW fp32 -> fp16, scaled) and im2col_small in one pass: the UNet's input convolution (Cin = 4) as a K = 64 GEMM
// needs only the im2col matrix, so the NHWC copy in between is never written.  8 consecutive k per thread (one 16-byte store).
__global__ void latent_im2col_kernel(const float* x, f16* y, int n_img, int h, int w, int c, int kpad, float scale) {
    const int kp8 = kpad / 8;
    const size_t total = (size_t)n_img * h * w * kp8;
    GRID_STRIDE(i, total) {
        const size_t row = i / kp8;
        const int k0 = (int)(i - row * kp8) * 8;
        const int img = (int)(row / ((size_t)h * w));
        const int rem = (int)(row - (size_t)img * h * w);
        const int oy = rem / w, ox = rem - oy * w;
        f16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            float f = 0.f;
            if (k < 9 * c) {
                const int tap = k / c, ch = k - tap * c;
                const int r = tap / 3, s2 = tap - r * 3;
                const int yy = oy + r - 1, xx = ox + s2 - 1;
                if (yy >= 0 && yy < h && xx >= 0 && xx < w) f = 0.f + x[(((size_t)img * c + ch) * h + yy) * w + xx] * scale;
            }
            v[e] = (f16)f;
        }
        *reinterpret_cast<f16x8*>(y + row * kpad + k0) = v;
    }
}

// The UNet's input convolution (3x3 pad 1, Cin = 4 -> Cout = 320) in ONE launch: latent NCHW fp32 -> im2col rows (K = 9 Cin padded
// to 64, the values latent_im2col_kernel writes) built straight in LDS, the whole [Cout][64] weight matrix next to them, one
// v_mfma_f32_16x16x32_f16 pair per 16 x 16 output block, bias, NHWC fp16 out.  A workgroup owns 32 pixels x all Cout columns
// (wave w: column blocks w, w + 4, ...): until now this was an im2col launch plus a K = 64 GEMM launch that is all prologue and
// epilogue (7 + 10 us at 64x64).  Same products in the same order as the GEMM (two K steps), bias added in fp32.
template <int NB> // 16-column blocks per wave: Cout = 64 * NB
__global__ __launch_bounds__(256) void conv_in_kernel(const float* x, const f16* w, const float* bias, f16* y, int n_img, int h, int wd, int c,
                                                      float scale) {
    __shared__ __attribute__((aligned(16))) f16 sa[32][64 + 8];
    __shared__ __attribute__((aligned(16))) f16 sw[64 * NB][64 + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cout = 64 * NB, hw = h * wd;
    const long long m0 = (long long)blockIdx.x * 32, M = (long long)n_img * hw;
    // weights: 8 pieces of 16 bytes per row, 2 NB pieces per thread -- all requested before the first is stored (one memory round
    // trip, not 2 NB of them)
    f16x8 wv[2 * NB];
#pragma unroll
    for (int i = 0; i < 2 * NB; ++i) wv[i] = ldg8(w + (size_t)(tid + 256 * i) * 8);
    // im2col rows: thread = (pixel, 8 consecutive k)
    {
        const int pl = tid >> 3, k0 = (tid & 7) * 8;
        const long long m = m0 + pl;
        f16x8 v = zero8();
        if (m < M) {
            const int img = (int)(m / hw), rem = (int)(m - (long long)img * hw);
            const int oy = rem / wd, ox = rem - oy * wd;
            int tap = k0 / c, ch = k0 - tap * c; // one division per thread; (tap, channel) then advance by increments
            const float* xi = x + (size_t)img * c * hw;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = 0.f;
                if (tap < 9) {
                    const int r = (tap * 11) >> 5, s2 = tap - r * 3; // tap / 3 for tap < 9
                    const int yy = oy + r - 1, xx = ox + s2 - 1;
                    if (yy >= 0 && yy < h && xx >= 0 && xx < wd) f = 0.f + xi[(size_t)ch * hw + yy * wd + xx] * scale;
                }
                v[e] = (f16)f;
                if (++ch == c) { ch = 0; ++tap; }
            }
        }
        *reinterpret_cast<f16x8*>(&sa[pl][k0]) = v;
    }
#pragma unroll
    for (int i = 0; i < 2 * NB; ++i) {
        const int idx = tid + 256 * i;
        *reinterpret_cast<f16x8*>(&sw[idx >> 3][(idx & 7) * 8]) = wv[i];
    }
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    f32x4 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        f16x8 fa[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f16x8*>(&sa[i * 16 + fr][ks * 32 + fg * 8]);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const f16x8 fb = *reinterpret_cast<const f16x8*>(&sw[(j * 4 + wave) * 16 + fr][ks * 32 + fg * 8]);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i][j] = mfma16(fb, fa[i], acc[i][j]); // lane: row i*16 + fr, columns (j*4+wave)*16 + 4 fg + r
        }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int n = (j * 4 + wave) * 16 + fg * 4;
        const f32x4 b = bias ? *reinterpret_cast<const f32x4*>(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long m = m0 + i * 16 + fr;
            if (m < M) {
                f16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (f16)(acc[i][j][r] + b[r]);
                *reinterpret_cast<f16x4*>(y + (size_t)m * cout + n) = o;
            }
        }
    }
}

__global__ void nchw_to_nhwc_kernel(const float* x, f16* y, int n, int c, int hw, float scale) {
    const size_t total = (size_t)n * c * hw;
    GRID_STRIDE(i, total) { // i indexes the NHWC output
        const int ch = (int)(i % c);
        const size_t t = i / c;
        const int pix = (int)(t % hw);
        const int img = (int)(t / hw);
        y[i] = (f16)(x[((size_t)img * c + ch) * hw + pix] * scale);
    }
}

// y[img][pix][o] = sum_c w[o][c] * (scale * x[img][c][pix]) + b[o]   (tiny channel counts: the 4-channel latent).
// Folds ldm's `z / 0.18215` and the VAE's 1x1 post_quant_conv into the layout change.
__global__ void latent_prep_kernel(const float* x, const float* w, const float* b, f16* y, int n, int c, int hw, float scale) {
    const size_t total = (size_t)n * c * hw;
    GRID_STRIDE(i, total) { // i indexes the NHWC output
        const int o = (int)(i % c);
        const size_t t = i / c;
        const int pix = (int)(t % hw);
        const int img = (int)(t / hw);
        float acc = b ? b[o] : 0.f;
        if (w) {
            for (int ch = 0; ch < c; ++ch) acc += w[o * c + ch] * (x[((size_t)img * c + ch) * hw + pix] * scale);
        } else {
            acc += x[((size_t)img * c + o) * hw + pix] * scale;
        }
        y[i] = (f16)acc;
    }
}

__global__ void nhwc_to_nchw_kernel(const f16* x, float* y, int n, int c, int hw) {
    const size_t total = (size_t)n * c * hw;
    GRID_STRIDE(i, total) { // i indexes the NCHW output
        const int pix = (int)(i % hw);
        const size_t t = i / hw;
        const int ch = (int)(t % c);
        const int img = (int)(t / c);
        y[i] = (float)x[((size_t)img * hw + pix) * c + ch];
    }
}

__global__ void embedding_kernel(const int32_t* ids, const f16* table, const f16* pos, f16* y, int rows, int seq, int c) {
    const int cp = c / 8;
    const size_t total = (size_t)rows * cp;
    GRID_STRIDE(i, total) {
        const int row = (int)(i / cp);
        const int c0 = (int)(i - (size_t)row * cp) * 8;
        const f16x8 t = ldg8(table + (size_t)ids[row] * c + c0);
        const f16x8 q = ldg8(pos + (size_t)(row % seq) * c + c0);
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f16)((float)t[e] + (float)q[e]);
        stg8(y + (size_t)row * c + c0, o);
    }
}

__global__ void timestep_features_kernel(const float* t, f16* y, int n, int dim) {
    const int half = dim / 2;
    const size_t total = (size_t)n * half;
    const float log_period = -logf(10000.0f);
    GRID_STRIDE(i, total) {
        const int row = (int)(i / half), j = (int)(i - (size_t)row * half);
        const float arg = t[row] * expf(log_period * j / half);
        y[(size_t)row * dim + j] = (f16)cosf(arg);
        y[(size_t)row * dim + half + j] = (f16)sinf(arg);
    }
}

// fp16 rows, fp32 math; each thread keeps <= NCH chunks of its row in registers (NCH = 4: rows up to 8192 columns, the
// SD v1 VAE attention at 64x64; NCH = 8: up to 16384, the 96x96 latent of SD v2.1-768)
template <int NCH>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const f16* x, f16* y, int M, int N) {
    __shared__ float red[8];
    const int row = blockIdx.x;
    const int cp = N / 8;
    const f16* xr = x + (size_t)row * N;
    f16x8 v[NCH];
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = threadIdx.x + 256 * i;
        if (ch < cp) {
            v[i] = ldg8(xr + ch * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)v[i][e]);
        }
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    float ev[NCH][8];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = threadIdx.x + 256 * i;
        if (ch < cp) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                ev[i][e] = __expf((float)v[i][e] - mx);
                sum += ev[i][e];
            }
        }
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    f16* yr = y + (size_t)row * N;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int ch = threadIdx.x + 256 * i;
        if (ch < cp) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (f16)(ev[i][e] * inv);
            stg8(yr + ch * 8, o);
        }
    }
}

// e = g*e_cond + (1-g)*e_uncond   (mode 0, the reference driver: scale, then accumulate)
// e = e_uncond + g*(e_cond-e_uncond)  (mode 1, ldm's PLMS/DDIM samplers)
__global__ void cfg_kernel(const f16* eps, float* out, int n, int c, int hw, float g, int uncond_first, int mode) {
    const size_t total = (size_t)n * c * hw;
    GRID_STRIDE(i, total) { // NCHW output index
        const int pix = (int)(i % hw);
        const size_t t = i / hw;
        const int ch = (int)(t % c);
        const int img = (int)(t / c);
        const int iu = uncond_first ? img : img + n;
        const int ic = uncond_first ? img + n : img;
        const float eu = (float)eps[((size_t)iu * hw + pix) * c + ch];
        const float ec = (float)eps[((size_t)ic * hw + pix) * c + ch];
        float e;
        if (mode == 0) {
            e = mul_rn(ec, g);
            e = add_rn(e, mul_rn(eu, sub_rn(1.0f, g)));
        } else {
            e = add_rn(eu, mul_rn(g, sub_rn(ec, eu)));
        }
        out[i] = e;
    }
}

// dpm_solver.cpp:136-181, same operation order, no FMA contraction
__global__ void dpm_update_kernel(float* x, const float* eps, float* y_prev, size_t count, int order, float sigma_s,
                                  float alpha_s, float sigma_ratio, float c_prev, float c_cur) {
    GRID_STRIDE(i, count) {
        const float xv = x[i];
        const float y = div_rn(add_rn(xv, mul_rn(-sigma_s, eps[i])), alpha_s); // :139
        float xn = mul_rn(xv, sigma_ratio);                                            // :153 / :168
        if (order == 2) xn = add_rn(xn, mul_rn(c_prev, y_prev[i]));                 // :169
        xn = add_rn(xn, mul_rn(c_cur, y));                                          // :154 / :170
        x[i] = xn;
        y_prev[i] = y;                                                                    // :177-180
    }
}

// ldm DDIM/PLMS step (eta = 0): pred_x0 = (x - s1m_at*e)/s_at ; x = s_aprev*pred_x0 + dir*e
__global__ void ddim_step_kernel(float* x, const float* e, size_t count, float s1m_at, float s_at, float s_aprev, float dir) {
    GRID_STRIDE(i, count) {
        const float ev = e[i];
        const float x0 = div_rn(sub_rn(x[i], mul_rn(s1m_at, ev)), s_at);
        x[i] = add_rn(mul_rn(s_aprev, x0), mul_rn(dir, ev));
    }
}

__global__ void lincomb4_kernel(float* out, const float* e0, const float* e1, const float* e2, const float* e3, float c0,
                                float c1, float c2, float c3, float div, size_t count) {
    GRID_STRIDE(i, count) {
        float v = mul_rn(c0, e0[i]);
        if (e1) v = add_rn(v, mul_rn(c1, e1[i]));
        if (e2) v = add_rn(v, mul_rn(c2, e2[i]));
        if (e3) v = add_rn(v, mul_rn(c3, e3[i]));
        out[i] = div_rn(v, div);
    }
}

__global__ void to_u8_kernel(const f16* img, uint8_t* out, size_t count, float a, float b, int mode) {
    GRID_STRIDE(i, count) {
        float f = add_rn(mul_rn(a, (float)img[i]), b);
        if (mode == 0) { // context.cpp:392-395: clamp(255*f, 0, 255), truncating cast
            f = mul_rn(255.0f, f);
            f = fminf(fmaxf(f, 0.0f), 255.0f);
        } else {         // ldm txt2img: 255 * clamp(f, 0, 1), truncating cast
            f = fminf(fmaxf(f, 0.0f), 1.0f);
            f = mul_rn(255.0f, f);
        }
        out[i] = (uint8_t)f;
    }
}

} // namespace


// ---- sampler-loop staging: the UNet graph's inputs for one guided evaluation in ONE launch.  The latent x (fp32 NCHW, n
// images) is written `reps` times back to back (uncond rows, then cond rows) and the projected time-conditioning row is
// broadcast to every batch row -- the three strided copies the host loop used to issue per evaluation.
// One PLMS step behind a UNet evaluation in ONE launch: classifier-free guidance (cfg_kernel's arithmetic), the optional
// v -> eps conversion, the multistep combination (lincomb4_kernel), the DDIM update (ddim_step_kernel) and the staging of the
// next evaluation's inputs (stage_unet_inputs_kernel) -- the same fp32 operations in the same order, so the results are the
// bits the four launches produce (tests/test_kernels_gpu.py::test_plms_update_equals_the_four_launches).
__global__ void plms_update_kernel(const sdod_plms_update_args a) {
    const size_t lat = (size_t)a.n * a.c * a.hw;
    const size_t nt = a.temb_row ? (size_t)a.temb_width * a.temb_reps : 0;
    const f16* eps = (const f16*)a.eps_nhwc;
    GRID_STRIDE(i, lat + nt) {
        if (i >= lat) {
            ((f16*)a.temb_dst)[i - lat] = ((const f16*)a.temb_row)[(i - lat) % a.temb_width];
            continue;
        }
        const int pix = (int)(i % a.hw);
        const size_t t = i / a.hw;
        const int ch = (int)(t % a.c);
        const int img = (int)(t / a.c);
        const int iu = a.uncond_first ? img : img + a.n;
        const int ic = a.uncond_first ? img + a.n : img;
        const float eu = (float)eps[((size_t)iu * a.hw + pix) * a.c + ch];
        const float ec = (float)eps[((size_t)ic * a.hw + pix) * a.c + ch];
        float e;
        if (a.mode == 0) {
            e = mul_rn(ec, a.guidance);
            e = add_rn(e, mul_rn(eu, sub_rn(1.0f, a.guidance)));
        } else {
            e = add_rn(eu, mul_rn(a.guidance, sub_rn(ec, eu)));
        }
        const float xv = a.x[i];
        if (a.v_pred) e = div_rn(add_rn(mul_rn(a.vc0, e), mul_rn(a.vc1, xv)), 1.0f); // lincomb4([e, x], [vc0, vc1], 1)
        a.e_out[i] = e;
        float v = mul_rn(a.c0, e);
        if (a.old1) v = add_rn(v, mul_rn(a.c1, a.old1[i]));
        if (a.old2) v = add_rn(v, mul_rn(a.c2, a.old2[i]));
        if (a.old3) v = add_rn(v, mul_rn(a.c3, a.old3[i]));
        const float ep = div_rn(v, a.div);
        const float x0 = div_rn(sub_rn(xv, mul_rn(a.sqrt_one_minus_at, ep)), a.sqrt_at);
        const float xn = add_rn(mul_rn(a.sqrt_a_prev, x0), mul_rn(a.dir_coef, ep));
        a.x[i] = xn;
        if (a.x_stage)
            for (int r = 0; r < a.stage_reps; ++r) a.x_stage[(size_t)r * lat + i] = xn;
    }
}

// The reference driver's per-step arithmetic behind a UNet evaluation in ONE launch (context.cpp:359-373 + dpm_solver.cpp:139-180 +
// the next step's input staging, :348-352): cfg_kernel (either mode), dpm_update_kernel and stage_unet_inputs_kernel, the same
// fp32 operations in the same order -- bit-identical to the three launches (test_dpm_step_equals_the_three_launches).
__global__ void dpm_step_kernel(const sdod_dpm_step_args a) {
    const size_t lat = (size_t)a.n * a.c * a.hw;
    const size_t nt = a.temb_row ? (size_t)a.temb_width * a.temb_reps : 0;
    const f16* eps = (const f16*)a.eps_nhwc;
    GRID_STRIDE(i, lat + nt) {
        if (i >= lat) {
            ((f16*)a.temb_dst)[i - lat] = ((const f16*)a.temb_row)[(i - lat) % a.temb_width];
            continue;
        }
        const int pix = (int)(i % a.hw);
        const size_t t = i / a.hw;
        const int ch = (int)(t % a.c);
        const int img = (int)(t / a.c);
        const int iu = a.uncond_first ? img : img + a.n;
        const int ic = a.uncond_first ? img + a.n : img;
        const float eu = (float)eps[((size_t)iu * a.hw + pix) * a.c + ch];
        const float ec = (float)eps[((size_t)ic * a.hw + pix) * a.c + ch];
        float e;
        if (a.mode == 0) {
            e = mul_rn(ec, a.guidance);
            e = add_rn(e, mul_rn(eu, sub_rn(1.0f, a.guidance)));
        } else {
            e = add_rn(eu, mul_rn(a.guidance, sub_rn(ec, eu)));
        }
        if (a.e_out) a.e_out[i] = e;
        const float xv = a.x[i];
        const float y = div_rn(add_rn(xv, mul_rn(-a.sigma_s, e)), a.alpha_s);
        float xn = mul_rn(xv, a.sigma_ratio);
        if (a.order == 2) xn = add_rn(xn, mul_rn(a.c_prev, a.y_prev[i]));
        xn = add_rn(xn, mul_rn(a.c_cur, y));
        a.x[i] = xn;
        a.y_prev[i] = y;
        if (a.x_stage)
            for (int r = 0; r < a.stage_reps; ++r) a.x_stage[(size_t)r * lat + i] = xn;
    }
}

__global__ void stage_unet_inputs_kernel(const float* x, float* x_dst, size_t lat, int reps, const f16* temb_row, f16* temb_dst,
                                         size_t temb_w, int temb_reps) {
    const size_t nx = lat * (size_t)reps, nt = temb_w * (size_t)temb_reps;
    GRID_STRIDE(i, nx + nt) {
        if (i < nx) x_dst[i] = x[i % lat];
        else temb_dst[i - nx] = temb_row[(i - nx) % temb_w];
    }
}

// ---- on-device x_T for throughput runs (SURVEY 7.2 "RNG"): Philox4x32-10 (Salmon et al., SC'11; known-answer vectors in
// tests/test_kernels_gpu.py via oracle/philox_oracle.py) keyed by `seed`, counter = (block index, stream); the four words
// of a block give two Box-Muller pairs.  Element i of stream s is a pure function of (seed, s, i): any sharding of the
// images over ranks draws the same latents.  The reference draws on the host (context.cpp:333-334, std::mt19937).
SDOD_DEVICE void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}
__global__ void randn_kernel(float* out, uint32_t* words, size_t count, uint64_t seed, uint64_t stream) {
    const size_t nblk = (count + 3) / 4;
    GRID_STRIDE(j, nblk) {
        uint32_t c[4] = {(uint32_t)j, (uint32_t)((uint64_t)j >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        float z[4];
#pragma unroll
        for (int a = 0; a < 4; a += 2) {
            const float u1 = ((float)(c[a] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float u2 = ((float)(c[a + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float rad = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.283185307179586f * u2, &sn, &cs);
            z[a] = rad * cs;
            z[a + 1] = rad * sn;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (4 * j + q < count) {
                out[4 * j + q] = z[q];
                if (words) words[4 * j + q] = c[q];
            }
    }
}

#define LAUNCH(kernel, work, st, ...)                                                            \
    do {                                                                                         \
        SDOD_LAUNCH(kernel, dim3(grid_for(work)), dim3(256), 0, (hipStream_t)(st), __VA_ARGS__); \
        SDOD_HIP_CHECK(hipGetLastError());                                                       \
    } while (0)

extern "C" int sdod_geglu_f16(const void* x, void* y, int m, int c, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && y && m > 0 && c > 0 && c % 8 == 0, "bad argument");
    LAUNCH(geglu_kernel, (size_t)m * (c / 8), stream, (const f16*)x, (f16*)y, m, c);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_act_f16(const void* x, void* y, size_t n, int act, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && y && n > 0 && n % 8 == 0, "bad argument (count must be a multiple of 8)");
    LAUNCH(act_kernel, n / 8, stream, (const f16*)x, (f16*)y, n / 8, act);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_add_f16(const void* a, const void* b, void* y, size_t n, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(a && b && y && n > 0 && n % 8 == 0, "bad argument (count must be a multiple of 8)");
    LAUNCH(add_kernel, n / 8, stream, (const f16*)a, (const f16*)b, (f16*)y, n / 8);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_concat_channels_f16(const void* a, const void* b, void* y, size_t rows, int c0, int c1, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(a && b && y && rows > 0 && c0 > 0 && c1 > 0 && c0 % 8 == 0 && c1 % 8 == 0, "bad argument");
    LAUNCH(concat_kernel, rows * ((c0 + c1) / 8), stream, (const f16*)a, (const f16*)b, (f16*)y, rows, c0, c1);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_im2col3x3_small_f16(const void* x, void* y, int n_img, int h, int w, int c, int kpad, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && y && n_img > 0 && h > 0 && w > 0 && c > 0 && kpad >= 9 * c, "bad argument");
    LAUNCH(im2col_small_kernel, (size_t)n_img * h * w * kpad, stream, (const f16*)x, (f16*)y, n_img, h, w, c, kpad);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_latent_im2col_f16(const float* x, void* y, int n_img, int h, int w, int c, int kpad, float scale, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && y && n_img > 0 && h > 0 && w > 0 && c > 0 && kpad >= 9 * c && kpad % 8 == 0 && ((uintptr_t)y & 15) == 0, "bad argument");
    LAUNCH(latent_im2col_kernel, (size_t)n_img * h * w * (kpad / 8), stream, x, (f16*)y, n_img, h, w, c, kpad, scale);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_conv_in_f16(const float* x, const void* w, const float* bias, void* y, int n_img, int h, int wd, int c, int cout,
                                float scale, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && w && y && n_img > 0 && h > 0 && wd > 0 && c > 0 && 9 * c <= 64, "bad argument (9 * Cin must fit the 64-deep K slab)");
    SDOD_REQUIRE(cout == 320 || cout == 256 || cout == 128 || cout == 64, "Cout must be 64, 128, 256 or 320");
    SDOD_REQUIRE((((uintptr_t)w | (uintptr_t)bias) & 15) == 0 && ((uintptr_t)y & 7) == 0, "misaligned pointer");
    const long long M = (long long)n_img * h * wd;
    const dim3 grid((unsigned)((M + 31) / 32));
    hipStream_t st = (hipStream_t)stream;
    switch (cout / 64) {
    case 1: SDOD_LAUNCH(conv_in_kernel<1>, grid, dim3(256), 0, st, x, (const f16*)w, bias, (f16*)y, n_img, h, wd, c, scale); break;
    case 2: SDOD_LAUNCH(conv_in_kernel<2>, grid, dim3(256), 0, st, x, (const f16*)w, bias, (f16*)y, n_img, h, wd, c, scale); break;
    case 4: SDOD_LAUNCH(conv_in_kernel<4>, grid, dim3(256), 0, st, x, (const f16*)w, bias, (f16*)y, n_img, h, wd, c, scale); break;
    default: SDOD_LAUNCH(conv_in_kernel<5>, grid, dim3(256), 0, st, x, (const f16*)w, bias, (f16*)y, n_img, h, wd, c, scale); break;
    }
    SDOD_HIP_CHECK(hipGetLastError());
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_nchw_f32_to_nhwc_f16(const float* x, void* y, int n, int c, int hw, float scale, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && y && n > 0 && c > 0 && hw > 0, "bad argument");
    LAUNCH(nchw_to_nhwc_kernel, (size_t)n * c * hw, stream, x, (f16*)y, n, c, hw, scale);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_latent_prep_f16(const float* x, const float* w, const float* b, void* y, int n, int c, int hw, float scale,
                                    void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && y && n > 0 && c > 0 && c <= 16 && hw > 0, "bad argument");
    LAUNCH(latent_prep_kernel, (size_t)n * c * hw, stream, x, w, b, (f16*)y, n, c, hw, scale);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_nhwc_f16_to_nchw_f32(const void* x, float* y, int n, int c, int hw, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && y && n > 0 && c > 0 && hw > 0, "bad argument");
    LAUNCH(nhwc_to_nchw_kernel, (size_t)n * c * hw, stream, (const f16*)x, y, n, c, hw);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_embedding_f16(const int32_t* ids, const void* table, const void* pos, void* y, int rows, int seq, int c,
                                  void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(ids && table && pos && y && rows > 0 && seq > 0 && c % 8 == 0, "bad argument");
    LAUNCH(embedding_kernel, (size_t)rows * (c / 8), stream, ids, (const f16*)table, (const f16*)pos, (f16*)y, rows, seq, c);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_timestep_features_f16(const float* t, void* y, int n, int dim, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(t && y && n > 0 && dim > 0 && dim % 2 == 0, "bad argument");
    LAUNCH(timestep_features_kernel, (size_t)n * (dim / 2), stream, t, (f16*)y, n, dim);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_softmax_rows_f16(const void* x, void* y, int m, int n, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && y && m > 0 && n > 0 && n % 8 == 0 && n <= 16384, "softmax rows need N % 8 == 0 and N <= 16384");
    if (n <= 8192) SDOD_LAUNCH(softmax_rows_kernel<4>, dim3(m), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, m, n);
    else SDOD_LAUNCH(softmax_rows_kernel<8>, dim3(m), dim3(256), 0, (hipStream_t)stream, (const f16*)x, (f16*)y, m, n);
    SDOD_HIP_CHECK(hipGetLastError());
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_stage_unet_inputs(const float* x, float* x_dst, size_t lat_count, int reps, const void* temb_row, void* temb_dst,
                                      size_t temb_width, int temb_reps, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && x_dst && lat_count > 0 && reps > 0, "bad latent argument");
    SDOD_REQUIRE((temb_row && temb_dst && temb_width > 0 && temb_reps > 0) || temb_reps == 0, "bad time-conditioning argument");
    LAUNCH(stage_unet_inputs_kernel, lat_count * (size_t)reps + temb_width * (size_t)temb_reps, stream, x, x_dst, lat_count, reps,
           (const f16*)temb_row, (f16*)temb_dst, temb_width, temb_reps);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_plms_update(const sdod_plms_update_args* a, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(a && a->eps_nhwc && a->e_out && a->x && a->n > 0 && a->c > 0 && a->hw > 0 && (a->mode == 0 || a->mode == 1) && a->div != 0.0f,
                 "bad argument");
    SDOD_REQUIRE(!a->x_stage || a->stage_reps > 0, "x_stage needs stage_reps");
    SDOD_REQUIRE(!a->temb_row || (a->temb_dst && a->temb_width > 0 && a->temb_reps > 0), "bad time-conditioning argument");
    const size_t work = (size_t)a->n * a->c * a->hw + (a->temb_row ? (size_t)a->temb_width * a->temb_reps : 0);
    LAUNCH(plms_update_kernel, work, stream, *a);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_dpm_step(const sdod_dpm_step_args* a, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(a && a->eps_nhwc && a->x && a->y_prev && a->n > 0 && a->c > 0 && a->hw > 0 && (a->mode == 0 || a->mode == 1) &&
                     (a->order == 1 || a->order == 2), "bad argument");
    SDOD_REQUIRE(!a->x_stage || a->stage_reps > 0, "x_stage needs stage_reps");
    SDOD_REQUIRE(!a->temb_row || (a->temb_dst && a->temb_width > 0 && a->temb_reps > 0), "bad time-conditioning argument");
    const size_t work = (size_t)a->n * a->c * a->hw + (a->temb_row ? (size_t)a->temb_width * a->temb_reps : 0);
    LAUNCH(dpm_step_kernel, work, stream, *a);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_randn_f32(float* out, uint32_t* words_out, size_t count, uint64_t seed, uint64_t stream_id, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(out && count > 0, "bad argument");
    LAUNCH(randn_kernel, (count + 3) / 4, stream, out, words_out, count, seed, stream_id);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_cfg_combine(const void* eps_nhwc, float* e_out, int n, int c, int hw, float guidance, int uncond_first,
                                int mode, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(eps_nhwc && e_out && n > 0 && c > 0 && hw > 0 && (mode == 0 || mode == 1), "bad argument");
    LAUNCH(cfg_kernel, (size_t)n * c * hw, stream, (const f16*)eps_nhwc, e_out, n, c, hw, guidance, uncond_first, mode);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_dpm_update(float* x, const float* eps, float* y_prev, size_t count, int order, float sigma_s,
                               float alpha_s, float sigma_ratio, float c_prev, float c_cur, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && eps && y_prev && count > 0 && (order == 1 || order == 2), "bad argument");
    LAUNCH(dpm_update_kernel, count, stream, x, eps, y_prev, count, order, sigma_s, alpha_s, sigma_ratio, c_prev, c_cur);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_ddim_step_f32(float* x, const float* e, size_t count, float sqrt_one_minus_at, float sqrt_at,
                                  float sqrt_a_prev, float dir_coef, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(x && e && count > 0, "bad argument");
    LAUNCH(ddim_step_kernel, count, stream, x, e, count, sqrt_one_minus_at, sqrt_at, sqrt_a_prev, dir_coef);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_lincomb4_f32(float* out, const float* e0, const float* e1, const float* e2, const float* e3, float c0,
                                 float c1, float c2, float c3, float div, size_t count, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(out && e0 && count > 0 && div != 0.0f, "bad argument");
    LAUNCH(lincomb4_kernel, count, stream, out, e0, e1, e2, e3, c0, c1, c2, c3, div, count);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_image_to_u8(const void* img, uint8_t* out, size_t count, float a, float b, int mode, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(img && out && count > 0 && (mode == 0 || mode == 1), "bad argument");
    LAUNCH(to_u8_kernel, count, stream, (const f16*)img, out, count, a, b, mode);
    return 0;
    SDOD_CATCH
}

// ---- pull a buffer towards the caches: every 128-byte line touched once (Graph::run_ops: weight prefetch on a side stream)
namespace {
__global__ __launch_bounds__(256) void l2_prefetch_kernel(const unsigned* __restrict__ p, size_t lines, unsigned* sink) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < lines; i += stride) acc += p[i * 32];
    if (acc == 0x9e3779b9u) *sink = acc; // (keeps the loads alive; practically never true)
}
} // namespace

extern "C" int sdod_l2_prefetch(const void* ptr, size_t bytes, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(ptr != nullptr && ((uintptr_t)ptr & 3) == 0, "null / unaligned pointer");
    const size_t lines = bytes / 128;
    if (lines == 0) return 0;
    static std::atomic<unsigned*> sink[64]; // one dump word per device (first use may race between threads: CAS, loser frees)
    int dev = 0;
    SDOD_HIP_CHECK(hipGetDevice(&dev));
    SDOD_REQUIRE(dev >= 0 && dev < 64, "device index");
    unsigned* cur = sink[dev].load(std::memory_order_acquire);
    if (!cur) {
        unsigned* fresh = nullptr;
        SDOD_HIP_CHECK(hipMalloc((void**)&fresh, 256));
        if (sink[dev].compare_exchange_strong(cur, fresh, std::memory_order_acq_rel)) cur = fresh;
        else (void)hipFree(fresh);
    }
    // few workgroups on purpose: the point is bytes in flight on an otherwise idle HBM, not CUs taken from the main chain
    const int blocks = (int)std::min<size_t>(48, (lines + 255) / 256);
    SDOD_LAUNCH(l2_prefetch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned*)ptr, lines, cur);
    SDOD_HIP_CHECK(hipGetLastError());
    return 0;
    SDOD_CATCH
}

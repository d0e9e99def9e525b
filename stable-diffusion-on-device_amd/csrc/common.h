// common.h -- shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wave = 64 lanes; MFMA 16x16x32 f16 fragments; LDS tiles addressed in 16-byte chunks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define SDOD_DEVICE __device__ __forceinline__

// D[i][j] += sum_k A[i][k] B[k][j];  lane l holds A[l&15][8(l>>4)+e], B[8(l>>4)+e][l&15], e=0..7
// and D[4(l>>4)+r][l&15], r=0..3   (cdna_hip_programming.md section 3)
SDOD_DEVICE f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// Reductions over the four lanes l, l ^ 16, l ^ 32, l ^ 48 of a wave (the lanes that share a row in the MFMA accumulator layout)
// with gfx950's row swaps instead of two ds_bpermute round trips: v_permlane16_swap(x, x) = {rows 0 0 2 2 | rows 1 1 3 3},
// v_permlane32_swap(x, x) = {rows 0 1 0 1 | rows 2 3 2 3}; combining the two halves is the xor-16 / xor-32 step.
// (The results are copied out as integers first: __builtin_bit_cast applied to an ELEMENT of the builtin's vector result reads
// element 0 for every index with this compiler.)
SDOD_DEVICE float quad_rows_max(float x) {
    const unsigned a16 = __builtin_bit_cast(unsigned, x);
    const auto s16 = __builtin_amdgcn_permlane16_swap(a16, a16, false, false);
    const unsigned p0 = s16[0], p1 = s16[1];
    x = fmaxf(__builtin_bit_cast(float, p0), __builtin_bit_cast(float, p1));
    const unsigned a32 = __builtin_bit_cast(unsigned, x);
    const auto s32 = __builtin_amdgcn_permlane32_swap(a32, a32, false, false);
    const unsigned q0 = s32[0], q1 = s32[1];
    return fmaxf(__builtin_bit_cast(float, q0), __builtin_bit_cast(float, q1));
}
SDOD_DEVICE float quad_rows_sum(float x) {
    const unsigned a16 = __builtin_bit_cast(unsigned, x);
    const auto s16 = __builtin_amdgcn_permlane16_swap(a16, a16, false, false);
    const unsigned p0 = s16[0], p1 = s16[1];
    x = __builtin_bit_cast(float, p0) + __builtin_bit_cast(float, p1);
    const unsigned a32 = __builtin_bit_cast(unsigned, x);
    const auto s32 = __builtin_amdgcn_permlane32_swap(a32, a32, false, false);
    const unsigned q0 = s32[0], q1 = s32[1];
    return __builtin_bit_cast(float, q0) + __builtin_bit_cast(float, q1);
}

// Sum / maximum over all 64 lanes, every lane gets the result: a butterfly of data-parallel-primitive moves inside the 16-lane rows
// (quad permutes for the 1- and 2-lane steps, half-row and row mirrors for the 4- and 8-lane steps: a symmetric pairing is all a
// reduction needs) and the two row swaps above, instead of six ds_bpermute round trips (~100 cycles each on a dependent chain).
template <int CTRL>
SDOD_DEVICE float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
SDOD_DEVICE float wave_sum(float v) {
    v += dpp_move<0xB1>(v);  // quad_perm [1, 0, 3, 2]
    v += dpp_move<0x4E>(v);  // quad_perm [2, 3, 0, 1]
    v += dpp_move<0x141>(v); // row_half_mirror
    v += dpp_move<0x140>(v); // row_mirror
    return quad_rows_sum(v);
}
SDOD_DEVICE float oct_sum(float v) { // sum over the aligned group of 8 lanes a lane belongs to, in every lane of the group
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    return v;
}
template <int N>
SDOD_DEVICE float group_sum(float v) { // sum over the aligned group of N = 2 .. 64 lanes a lane belongs to, in every lane of the group
    static_assert(N == 2 || N == 4 || N == 8 || N == 16 || N == 32 || N == 64, "power of two");
    v += dpp_move<0xB1>(v);
    if constexpr (N >= 4) v += dpp_move<0x4E>(v);
    if constexpr (N >= 8) v += dpp_move<0x141>(v);
    if constexpr (N >= 16) v += dpp_move<0x140>(v);
    if constexpr (N >= 32) {
        const unsigned a16 = __builtin_bit_cast(unsigned, v);
        const auto s16 = __builtin_amdgcn_permlane16_swap(a16, a16, false, false);
        const unsigned p0 = s16[0], p1 = s16[1];
        v = __builtin_bit_cast(float, p0) + __builtin_bit_cast(float, p1);
    }
    if constexpr (N >= 64) {
        const unsigned a32 = __builtin_bit_cast(unsigned, v);
        const auto s32 = __builtin_amdgcn_permlane32_swap(a32, a32, false, false);
        const unsigned q0 = s32[0], q1 = s32[1];
        v = __builtin_bit_cast(float, q0) + __builtin_bit_cast(float, q1);
    }
    return v;
}
SDOD_DEVICE float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    return quad_rows_max(v);
}

// 1 / sqrt(x) for the statistics of the normalisation kernels: v_rsq_f32 (1 ulp) instead of an IEEE square root followed by an IEEE
// division -- ~25 dependent instructions on the one lane every other lane of the group waits for
SDOD_DEVICE float rsqrt_fast(float x) { return __builtin_amdgcn_rsqf(x); }

SDOD_DEVICE f16x8 zero8() {
    f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return z;
}

// 16-byte global load / store of 8 halves
SDOD_DEVICE f16x8 ldg8(const f16* p) { return *reinterpret_cast<const f16x8*>(p); }
SDOD_DEVICE void stg8(f16* p, f16x8 v) { *reinterpret_cast<f16x8*>(p) = v; }

// x * sigmoid(x) with v_rcp_f32 (1 ulp) instead of an IEEE division: `x / (1 + e)` compiles to the ten-instruction
// div_scale / rcp / fma / div_fmas / div_fixup sequence, per element, in every GroupNorm + SiLU kernel
SDOD_DEVICE float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// GELU(x) = x * Phi(x) = max(x, 0) - |x| * Q(|x|) with Q(t) = erfc(t / sqrt 2) / 2 = 2^-P(t): -log2 Q is smooth (~ t^2 / (2 ln 2)
// + log2 t), so a degree-5 polynomial with P(0) = 1 (weighted minimax fit of the error of |x| * Q on [0, 6], monotone beyond)
// gives |error| <= 5.4e-7 over all x, three orders below fp16 resolution -- five fma, one v_exp_f32, one max and one fma.
// (Until round 3: erf by Abramowitz & Stegun 7.1.26, |error| <= 2.2e-7, with an rcp AND an exp: 23 issue slots against these
// 11; the GEGLU epilogue evaluates it 10.5 M times per ff GEMM and was VALU-bound on it.  The library erff is ~40 instructions.)
SDOD_DEVICE float gelu_erf_f(float x) {
    const float a = fabsf(x);
    float q = 0.000488102092f;
    q = fmaf(q, a, -0.0071987181f); // (the library is built with -ffp-contract=off: fma where fma is meant)
    q = fmaf(q, a, 0.052146631f);
    q = fmaf(q, a, 0.459595845f);
    q = fmaf(q, a, 1.15100054f);
    q = fmaf(q, a, 1.0f);
    return fmaf(-a, __builtin_amdgcn_exp2f(-q), fmaxf(x, 0.0f));
}
SDOD_DEVICE float quick_gelu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }

enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2, ACT_QUICK_GELU = 3 };

SDOD_DEVICE float apply_act(float x, int act) {
    switch (act) {
    case ACT_SILU: return silu_f(x);
    case ACT_GELU: return gelu_erf_f(x);
    case ACT_QUICK_GELU: return quick_gelu_f(x);
    default: return x;
    }
}

// XCD-aware block remap (bijective form, cdna_hip_programming.md T1): blocks that share the
// low 3 bits of their id share an XCD/L2; give each XCD a contiguous range of logical tiles.
SDOD_DEVICE int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

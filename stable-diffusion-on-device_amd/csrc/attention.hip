// attention.hip -- fused softmax(scale * Q K^T) V for the UNet's self/cross attention and CLIP's causal
// attention on gfx950.  In the reference these are the `/attn*/to_*`, `/smax/`, `/MatMul` ops inside the
// opaque UNet / text-encoder QNN graphs (analyze_results.py:69-79; executed at qnn_context.cpp:711-713).
//
// The score matrix (4096x4096 per head-batch at the 64x64 level) never touches HBM: a workgroup owns
// 64*QT query rows of one (batch, head), walks K/V in 64-key tiles staged through LDS (register
// prefetch, double-buffered, one barrier per tile) and keeps an online softmax in registers.
//
// wave64 / MFMA mapping (v_mfma_f32_16x16x32_f16), chosen so that NO cross-lane data movement is needed
// between the two matrix products:
//   * S^T = K . Q^T  (K rows are the MFMA "A" operand, Q rows the "B" operand): lane l ends up with the
//     scores of ONE query (column l&15) against keys 16*t + 4*(l>>4) + r  -> the row max / row sum are
//     in-lane reductions plus two xor-shuffles (16, 32);
//   * O^T = V^T . P^T: the P registers a lane already holds ARE its B-operand fragment (keys on the
//     contraction axis), and V^T fragments come from the row-major V tile through the gfx950
//     transposed LDS read ds_read_b64_tr_b16 (4 keys x 16 columns per 16-lane group);
//   * O^T keeps the query on the lane, so the online-softmax rescale is a per-lane scalar multiply.
// Head dims 40/80/160 (SD v1) and 64 (CLIP, SD v2) are zero-padded in LDS to the MFMA granularity.
#include "common.h"
#include "sdod_hip.h"
#include "host_util.h"

#include <cstdlib>
#include <type_traits>

namespace {

struct AttnP {
    const f16* q;
    const f16* k;
    const f16* v;
    f16* out;
    int B, H, Lq, Lk;
    int ldq, ldk, ldv, ldo;
    float scale_log2;
    int causal;
};

typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) short4v* lds_short4_ptr;

constexpr int odd16(int dv) { // smallest 16*odd >= dv  (V row stride in halves: 32*odd bytes, conflict-free tr reads)
    int s = (dv + 15) / 16;
    if ((s & 1) == 0) s += 1;
    return s * 16;
}

// waves per SIMD the register allocation must leave room for: the d = 40 two-query-tile kernel needs 170 VGPRs left alone --
// two over the 168 that let THREE workgroups share a CU (the kernel is issue-bound with stalls a third wave can fill:
// SQ counters, profiles/r02_attention_sq_counters.txt); same step for d = 80 with one query tile (176)
template <int D, int QT>
constexpr int attn_min_waves() { return (D == 40 || (D == 80 && QT == 1)) ? 3 : 1; }

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <int N, int I = 0, class F>
SDOD_DEVICE void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// KVS = key/value split INSIDE the workgroup: KVS groups of four waves own the SAME 64*QT query rows and every KVS-th key tile
// (own LDS stages), and merge their (max, sum, O) through LDS at the end.  For the launches whose grid is one workgroup per CU
// (d = 80 at 32x32: 256 workgroups) the tile loop is a latency-bound chain on one wave per SIMD; a second group halves the
// chain and doubles the waves that hide it, at the same K/V traffic.
template <int D, int QT, bool TR, int KVS = 1>
__global__ __launch_bounds__(256 * KVS, (attn_min_waves<D, QT>())) void attn_kernel(const AttnP p) {
    constexpr int DP = ((D + 31) / 32) * 32; // QK^T contraction, padded to MFMA K=32
    constexpr int KSTEPS = DP / 32;
    constexpr int DV = ((D + 15) / 16) * 16; // PV output columns, padded to 16
    constexpr int NDT = DV / 16;
    constexpr int KSTR = DP + 8;             // K tile row stride (halves)
    constexpr int VSTR = odd16(DV);          // V tile row stride (halves)
    constexpr int KT = 64;                   // keys per tile
    constexpr int DC = D / 8;                // 16-byte chunks per K/V row
    constexpr int LD_IT = (KT * DC + 255) / 256;
    constexpr int STAGE = KT * (KSTR + VSTR);
    // d = 40: V is padded to 48 columns; a column of ones there makes the PV product accumulate the softmax row sum
    // (sum of the SAME fp16-rounded probabilities that weight V) -- 16 packed adds per tile and query tile less on the VALU
    constexpr bool SUM_BY_MFMA = DV > D;
    static_assert(D % 8 == 0, "head dim must be a multiple of 8");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int grp = KVS > 1 ? (int)(threadIdx.x >> 8) : 0; // key/value group of this wave (wave-uniform)
    f16* smem = reinterpret_cast<f16*>(smem_raw) + (size_t)grp * 2 * STAGE;

    const int tid = threadIdx.x & 255; // thread inside its group
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int g = lane >> 4;   // lane group 0..3
    const int li = lane & 15;  // index inside the group
    // 1-D grid, XCD-aware: workgroups b and b + 8 share an XCD and its L2 (round-robin dispatch; speed only), so each XCD is
    // handed a contiguous run of logical ids, and the query blocks of one (batch, head) are consecutive ids: the K/V of a
    // head (655 KB at L = 4096, d = 40) is then streamed through ONE L2 instead of all eight (16 heads x 655 KB would thrash
    // every 4 MiB L2 and every K/V tile load would pay the Infinity-Cache round trip)
    const int nqb = (p.Lq + 64 * QT - 1) / (64 * QT);
    const int lid = xcd_remap(blockIdx.x, nqb * p.H * p.B);
    const int bh = lid / nqb;
    const int b = bh / p.H, h = bh - b * p.H;
    const int q_block = (lid - bh * nqb) * (64 * QT);

    const f16* kbase = p.k + (size_t)b * p.Lk * p.ldk + h * D;
    const f16* vbase = p.v + (size_t)b * p.Lk * p.ldv + h * D;

    // zero the padded columns of both stages once (they are never overwritten by the staging stores)
    {
        constexpr int KPADC = (DP - D) / 8, VPADC = (DV - D) / 8;
        for (int st = 0; st < 2; ++st) {
            f16* sK = smem + st * STAGE;
            f16* sV = sK + KT * KSTR;
            if (KPADC > 0)
                for (int idx = tid; idx < KT * KPADC; idx += 256) {
                    const int row = idx / (KPADC > 0 ? KPADC : 1), ch = idx - row * KPADC;
                    *reinterpret_cast<f16x8*>(sK + row * KSTR + D + ch * 8) = zero8();
                }
            if (VPADC > 0)
                for (int idx = tid; idx < KT * VPADC; idx += 256) {
                    const int row = idx / (VPADC > 0 ? VPADC : 1), ch = idx - row * VPADC;
                    f16x8 pad = zero8();
                    if (ch == 0) pad[0] = (f16)1.0f; // column D: the row-sum column (SUM_BY_MFMA)
                    *reinterpret_cast<f16x8*>(sV + row * VSTR + D + ch * 8) = pad;
                }
        }
    }

    // Q fragments stay in registers for the whole kernel
    f16x8 qf[QT][KSTEPS];
    int qrow[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        qrow[t] = q_block + wave * (16 * QT) + t * 16 + li;
        const f16* qp = p.q + ((size_t)b * p.Lq + qrow[t]) * p.ldq + h * D;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int d0 = ks * 32 + g * 8;
            qf[t][ks] = (qrow[t] < p.Lq && d0 + 8 <= D) ? ldg8(qp + d0) : zero8();
        }
    }

    int kend = p.Lk;
    if (p.causal) kend = min(p.Lk, q_block + 64 * QT);
    const int NT = (kend + KT - 1) / KT;

    f16x8 rk[LD_IT], rv[LD_IT];
    const f16* kptr0[LD_IT];
    const f16* vptr0[LD_IT]; // this thread's 16-byte pieces of key tile 0
#pragma unroll
    for (int i = 0; i < LD_IT; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx / DC, ch = idx - row * DC;
        kptr0[i] = kbase + (size_t)row * p.ldk + ch * 8;
        vptr0[i] = vbase + (size_t)row * p.ldv + ch * 8;
    }
    auto load_tile = [&](int t) {
        // A tile that lies entirely inside the sequence (every tile but a ragged last one) needs no per-key predicate and no
        // zero fill: the lanes past the tile's 16-byte pieces keep whatever their registers hold -- store_tile skips them.
        // (The predicated form costs a compare, an exec dance and eight register clears per piece: ~10 % of the loop's VALU.)
        if (t * KT + KT <= p.Lk) {
            const size_t ko = (size_t)t * KT * p.ldk, vo = (size_t)t * KT * p.ldv; // wave-uniform: one 64-bit add per load
#pragma unroll
            for (int i = 0; i < LD_IT; ++i) {
                if (tid + 256 * i < KT * DC) {
                    rk[i] = ldg8(kptr0[i] + ko);
                    rv[i] = ldg8(vptr0[i] + vo);
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < LD_IT; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / DC, ch = idx - row * DC;
            const int key = t * KT + row;
            const bool ok = (idx < KT * DC) && (key < p.Lk);
            rk[i] = ok ? ldg8(kbase + (size_t)key * p.ldk + ch * 8) : zero8();
            rv[i] = ok ? ldg8(vbase + (size_t)key * p.ldv + ch * 8) : zero8();
        }
    };
    auto store_tile = [&](int st) {
        f16* sK = smem + st * STAGE;
        f16* sV = sK + KT * KSTR;
#pragma unroll
        for (int i = 0; i < LD_IT; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx / DC, ch = idx - row * DC;
            if (idx < KT * DC) {
                *reinterpret_cast<f16x8*>(sK + row * KSTR + ch * 8) = rk[i];
                *reinterpret_cast<f16x8*>(sV + row * VSTR + ch * 8) = rv[i];
            }
        }
    };

    f32x4 o[QT][NDT];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_run[t] = -1e30f;
        l_run[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

#ifdef SDOD_ATTN_ABLATE
    constexpr int abl = SDOD_ATTN_ABLATE; // developer builds only (make lib/libsdod_attnabl<mask>.so): 1 = no K/V loads after tile 0,
#else                                     // 2 = no exp in the softmax; results are wrong, only the timing is of interest
    constexpr int abl = 0;
#endif
    constexpr bool HOLD = D <= 80; // K / V^T fragments of a tile are read once and kept for both query tiles (d = 160: read at the point of use)
    auto read_vt = [&](const f16* sV, int u, int dt) -> f16x8 {
        const f16* a0 = sV + (32 * u + 4 * g + (li >> 2)) * VSTR + dt * 16 + 4 * (li & 3);
        const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_ptr)(a0));
        const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_ptr)(a0 + 16 * VSTR));
        typedef short short8v __attribute__((__vector_size__(8 * sizeof(short))));
        const short8v both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(f16x8, both);
    };
    // ---- online softmax of query tile A against key tile t (scores in s_a) and O^T += V^T . P^T; lane owns query qrow[A],
    // keys t*64 + c*16 + 4g + r.  VALU budget matters here (at d=40 the MFMAs of a tile take ~450 cycles, a naive softmax 3x
    // that): the row max is taken on the RAW scores (scale > 0), exp2(s*c - m*c) is one fma + one v_exp, masking code only
    // runs for tiles that contain masked keys (the last ragged tile / the causal diagonal).
    auto softmax_pv = [&](auto a_c, int t, f32x4 (&s_a)[4], const f16x8 (&vfr)[HOLD ? 2 : 1][HOLD ? NDT : 1], const f16* sV) {
        constexpr int a = decltype(a_c)::value;
        const bool need_mask = (t * KT + KT > p.Lk) || (p.causal && (t * KT + KT - 1 > q_block + wave * (16 * QT)));
        if (need_mask) {
            // (key0 goes through an opaque statement so that the sixteen key indices are computed INSIDE this rarely taken
            // branch: the compiler otherwise hoists them in front of it and every tile pays for them)
            int key0 = t * KT + g * 4;
            asm volatile("" : "+v"(key0));
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = key0 + c * 16 + r;
                    const bool masked = (key >= p.Lk) | (p.causal & (key > qrow[a]));
                    s_a[c][r] = masked ? -1e30f : s_a[c][r];
                }
        }
        // row maximum of this lane's 16 scores: three-input maxima (v_max3_f32: 8 instructions instead of 15) as a TREE of depth 3,
        // not a chain of 8 -- the maximum heads the tile's dependent chain (max -> exp -> PV)
        float mx;
        {
            const float t0 = fmaxf(fmaxf(s_a[0][0], s_a[0][1]), s_a[0][2]);
            const float t1 = fmaxf(fmaxf(s_a[0][3], s_a[1][0]), s_a[1][1]);
            const float t2 = fmaxf(fmaxf(s_a[1][2], s_a[1][3]), s_a[2][0]);
            const float t3 = fmaxf(fmaxf(s_a[2][1], s_a[2][2]), s_a[2][3]);
            const float t4 = fmaxf(fmaxf(s_a[3][0], s_a[3][1]), s_a[3][2]);
            const float u0 = fmaxf(fmaxf(t0, t1), t2);
            const float u1 = fmaxf(fmaxf(t3, t4), s_a[3][3]);
            mx = fmaxf(u0, u1);
        }
        // the four lanes that share a query: row swaps in VALU latency instead of two ds_bpermute round trips on the loop's
        // critical path (common.h: quad_rows_max; profiles/r03_attention_permlane.txt)
        mx = quad_rows_max(mx);
        const float m_new = fmaxf(m_run[a], mx * p.scale_log2); // running max in scaled (log2) units
        // once the running maxima have settled (a few tiles in) no lane of the wave changes its maximum: skip the
        // rescale of the output accumulators (wave-uniform branch)
        const bool rescale = __builtin_amdgcn_ballot_w64(m_new != m_run[a]) != 0;
        const float alpha = rescale ? __builtin_amdgcn_exp2f(m_run[a] - m_new) : 1.0f;
        m_run[a] = m_new;
        // plain fp32 fma / add, NOT the packed forms: beside MFMAs a v_pk_add_f32 / v_pk_fma_f32 costs several times the issue
        // cycles of the two scalar instructions it replaces (MI355X_MICROARCH.md, per-instruction constants: an anti-lever), and
        // this loop is bound by the SIMD's vector issue (-ffp-contract=off: the fma is spelled out)
        float rs[4] = {0.f, 0.f, 0.f, 0.f};
        const float sc = p.scale_log2, nm = -m_new;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = __builtin_fmaf(s_a[c][r], sc, nm);
                if (!(abl & 2)) v = __builtin_amdgcn_exp2f(v);
                if (!SUM_BY_MFMA) rs[r] += v;
                s_a[c][r] = v;
            }
        }
        if (!SUM_BY_MFMA) l_run[a] = __builtin_fmaf(l_run[a], alpha, (rs[0] + rs[1]) + (rs[2] + rs[3]));
        if (rescale) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[a][dt][r] *= alpha;
        }
        f16x8 pf[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f16x8 f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f[r] = (f16)s_a[2 * u][r];
                f[4 + r] = (f16)s_a[2 * u + 1][r];
            }
            pf[u] = f;
        }
        // ---- O^T += V^T . P^T for this query tile; contraction element e of lane group g is key 32u + 16(e>>2) + 4g + (e&3)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                f16x8 vf;
                if (TR) {
                    vf = HOLD ? vfr[u][dt] : read_vt(sV, u, dt);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) vf[e] = sV[(32 * u + 16 * (e >> 2) + 4 * g + (e & 3)) * VSTR + dt * 16 + li];
                }
                o[a][dt] = mfma16(vf, pf[u], o[a][dt]);
            }
        }
    };
    auto read_k = [&](const f16* sK, f16x8 (&kf)[HOLD ? KSTEPS : 1][4]) {
        if constexpr (HOLD) {
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
                for (int c = 0; c < 4; ++c) kf[ks][c] = *reinterpret_cast<const f16x8*>(sK + (c * 16 + li) * KSTR + ks * 32 + g * 8);
        }
    };
    auto read_v = [&](const f16* sV, f16x8 (&vfr)[HOLD ? 2 : 1][HOLD ? NDT : 1]) {
        if constexpr (TR && HOLD) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) vfr[u][dt] = read_vt(sV, u, dt);
        }
    };
    // S^T = K . Q^T of one key tile for every query tile of the wave
    auto qk = [&](const f16* sK, const f16x8 (&kf)[HOLD ? KSTEPS : 1][4], f32x4 (&s)[QT][4]) {
#pragma unroll
        for (int a = 0; a < QT; ++a) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f16x8 k8 = HOLD ? kf[ks][c] : *reinterpret_cast<const f16x8*>(sK + (c * 16 + li) * KSTR + ks * 32 + g * 8);
                    s[a][c] = mfma16(k8, qf[a][ks], s[a][c]);
                }
        }
    };

    // (Round 3 tried a software-pipelined form of this loop -- the scores of tile t + 1 computed under the softmax of tile t,
    // three LDS stages, K fragments read behind the previous barrier: 220 VGPRs, 8 % SLOWER (123 vs 113 us at 4096^2, d = 40:
    // the compiler serialises the longer body no better, and the third LDS stage costs occupancy where B >= 4 would have
    // used it).  Removed; profiles/r03_attention_occupancy.txt keeps the numbers.)
    if (grp < NT) {
        load_tile(grp);
        store_tile(0);
    }
    __syncthreads();
    const int NTG = (NT + KVS - 1) / KVS; // every group runs the same number of iterations (the barriers are workgroup-wide)
    for (int it = 0; it < NTG; ++it) {
        const int t = it * KVS + grp;
        const int cur = it & 1;
        const bool has_next = t + KVS < NT;
        const f16* sK = smem + cur * STAGE;
        const f16* sV = sK + KT * KSTR;
        if (t >= NT) { // (KVS > 1: a ragged last round)
            __syncthreads();
            continue;
        }
        // ---- fragments of the tile first: K for S^T = K . Q^T and (transposed) V for O^T += V^T . P^T are read ONCE and kept
        // in registers for both query tiles; the V reads are issued here so that they land during the softmax
        // (D = 160 would need 160 registers for them and drop to one wave per SIMD: it reads at the point of use)
        f16x8 kf[HOLD ? KSTEPS : 1][4];
        f16x8 vfr[HOLD ? 2 : 1][HOLD ? NDT : 1];
        read_k(sK, kf);
        read_v(sV, vfr);
        // ---- S^T = K . Q^T, query tile by query tile: the softmax of tile a starts (VALU) while the matrix pipe is still
        // busy with tile a + 1, and further down the PV product of tile a runs under the softmax of tile a + 1
        f32x4 s[QT][4];
        qk(sK, kf, s);
        // the next tile's K/V are requested HERE, behind the fragment reads and the QK^T issue: requested at the top of the
        // iteration, the compiler's wait-count pass (it cannot count loads under divergent predicates) put a vmcnt(0) in front
        // of the first MFMA, i.e. the full global-load latency on the critical path of every tile
        if (has_next && !(abl & 1)) load_tile(t + KVS);
        static_for<QT>([&](auto a_c) { softmax_pv(a_c, t, s[decltype(a_c)::value], vfr, sV); });
        if (has_next) store_tile(cur ^ 1);
        __syncthreads();
    }

    if constexpr (KVS > 1) {
        // merge the groups' partial softmaxes: group g > 0 parks (m, l, O) of its lanes in LDS ([value][thread]: conflict-free),
        // group 0 folds them in -- the usual online-softmax combination, once -- and stores
        static_assert(KVS == 2, "the merge below handles two groups");
        float* mrg = reinterpret_cast<float*>(smem_raw);
        constexpr int NV = QT * (2 + NDT * 4);
        static_assert((size_t)NV * 256 * sizeof(float) <= (size_t)KVS * 2 * STAGE * sizeof(f16), "merge buffer fits the staging LDS");
        if (grp == 1) {
#pragma unroll
            for (int a = 0; a < QT; ++a) {
                mrg[(a * (2 + NDT * 4) + 0) * 256 + tid] = m_run[a];
                mrg[(a * (2 + NDT * 4) + 1) * 256 + tid] = l_run[a];
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) mrg[(a * (2 + NDT * 4) + 2 + dt * 4 + r) * 256 + tid] = o[a][dt][r];
            }
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int a = 0; a < QT; ++a) {
            const float m1 = mrg[(a * (2 + NDT * 4) + 0) * 256 + tid], l1 = mrg[(a * (2 + NDT * 4) + 1) * 256 + tid];
            const float m = fmaxf(m_run[a], m1);
            const float a0 = __builtin_amdgcn_exp2f(m_run[a] - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
            l_run[a] = l_run[a] * a0 + l1 * a1;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) o[a][dt][r] = o[a][dt][r] * a0 + mrg[(a * (2 + NDT * 4) + 2 + dt * 4 + r) * 256 + tid] * a1;
            m_run[a] = m;
        }
    }
    // ---- normalise and store: lane holds O[q = qrow][dv = dt*16 + 4g + r]
#pragma unroll
    for (int a = 0; a < QT; ++a) {
        float l;
        if (SUM_BY_MFMA) {
            // the row sum sits in output column D = (D / 16) * 16 + 4 g + r  ->  lane group g = (D % 16) / 4, register r = D % 4
            l = __shfl(o[a][D / 16][D % 4], ((D % 16) / 4) * 16 + li);
        } else {
            l = quad_rows_sum(l_run[a]);
        }
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        if (qrow[a] < p.Lq) {
            f16* op = p.out + ((size_t)b * p.Lq + qrow[a]) * p.ldo + h * D;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const int dv0 = dt * 16 + g * 4;
                if (dv0 < D) {
                    f16x4 hv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) hv[r] = (f16)(o[a][dt][r] * inv);
                    *reinterpret_cast<f16x4*>(op + dv0) = hv;
                }
            }
        }
    }
}

template <int D, int QT, bool TR, int KVS = 1>
hipError_t attn_launch(const AttnP& p, hipStream_t st) {
    constexpr int DP = ((D + 31) / 32) * 32;
    constexpr int DV = ((D + 15) / 16) * 16;
    constexpr size_t smem = (size_t)KVS * 2 * 64 * (DP + 8 + odd16(DV)) * sizeof(f16);
    static_assert(smem <= 160 * 1024, "LDS");
    static std::atomic<unsigned long long> attr_devs{0};
    if (sdod::first_use_on_device(attr_devs)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<D, QT, TR, KVS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    dim3 grid(((p.Lq + 64 * QT - 1) / (64 * QT)) * p.H * p.B);
    SDOD_LAUNCH((attn_kernel<D, QT, TR, KVS>), grid, dim3(256 * KVS), smem, st, p);
    return hipGetLastError();
}

template <int D>
hipError_t attn_dispatch(const AttnP& p, bool big, bool tr, hipStream_t st) {
    if (tr) return big ? attn_launch<D, 2, true>(p, st) : attn_launch<D, 1, true>(p, st);
    return big ? attn_launch<D, 2, false>(p, st) : attn_launch<D, 1, false>(p, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// Folded cross-attention (once per prompt).  The text context -- and with it every cross-attention K and V -- is constant
// over the sampler run, so the two Linears around the attention can be multiplied INTO K and V:
//   scores_h = LN(x) Wq_h^T K_h^T = LN(x) . W1_h,      W1_h = Wq_h^T K_h^T   [C x L]
//   out      = sum_h P_h V_h Wo_h^T = [P_1 | ... | P_H] . [W2_1 ; ... ; W2_H],   W2_h = V_h Wo_h^T   [L x C]
// i.e. per evaluation the block is TWO plain GEMMs (the softmax runs in the epilogue of the first, sdod_gemm_desc::softmax_cols)
// instead of Linear + attention kernel + Linear; L = 77 keys are padded to 80 columns per head.  This kernel builds the
// per-image matrices: C[bt][m][n] = alpha * sum_k A[bt][m][k] B[bt][n][k] for many small (64 x 64 x d) problems with arbitrary
// element strides (the operands are slices of the K|V projection and of the weight matrices), rows / columns past the valid
// range produce zeros.  fp16 in, fp32 accumulate, fp16 out; 16 outputs per thread from LDS tiles.
struct FoldP {
    const f16* kv; int ld_kv, k_off, v_off, L, n_img;
    const f16* wq; int ldq;
    const float* sq; const float* tq;
    const f16* wo; int ldwo;
    int heads, d, C;
    float alpha;
    f16* w1; float* s1; float* t1; f16* w2;
    int tiles_c; // C / 80
    int n_w;     // work items of W1 (and of W2): n_img * heads * tiles_c
};
constexpr int kFoldKMax = 160;
// One launch per transformer block: workgroups [0, n_w) build 80 x 80 tiles of W1 (rows = a head's 80 key slots, columns = 80
// input channels), [n_w, 2 n_w) tiles of W2 (rows = 80 output channels, columns = a head's key slots), the rest the fold
// vectors.  Both products contract over the head dimension d (<= 160, zero-padded to the MFMA's 32): operands staged in LDS as
// [row][k], v_mfma_f32_16x16x32_f16, 5 x 5 blocks of 16 x 16 per tile over 4 waves.
__global__ __launch_bounds__(256) void xattn_fold_kernel(const FoldP p) {
    __shared__ __attribute__((aligned(16))) f16 sa[80][kFoldKMax + 8], sb[80][kFoldKMax + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NK = p.heads * 80, d = p.d;
    int item = blockIdx.x;
    if (item >= 2 * p.n_w) {
        // fold vectors of the score GEMM: s1 = alpha * K_h . s_q, t1 = alpha * K_h . t_q; padding columns: s = 0 and a bias that
        // the softmax turns into an exact zero
        const int idx = (item - 2 * p.n_w) * 256 + tid;
        if (idx >= p.n_img * NK) return;
        const int j = idx % 80, bh = idx / 80, h = bh % p.heads, img = bh / p.heads;
        if (j >= p.L) {
            p.s1[idx] = 0.f;
            p.t1[idx] = -30000.f;
            return;
        }
        const f16* kr = p.kv + ((long long)img * p.L + j) * p.ld_kv + p.k_off + h * d;
        float a = 0.f, b = 0.f;
        for (int dd = 0; dd < d; ++dd) {
            const float kval = (float)kr[dd];
            a = fmaf(kval, p.sq[h * d + dd], a);
            b = fmaf(kval, p.tq[h * d + dd], b);
        }
        p.s1[idx] = p.alpha * a;
        p.t1[idx] = p.alpha * b;
        return;
    }
    const bool second = item >= p.n_w;
    if (second) item -= p.n_w;
    const int tc = item % p.tiles_c, bh = item / p.tiles_c, h = bh % p.heads, img = bh / p.heads;
    const int KP = (d + 31) & ~31, kc = KP / 8;
    // rows of K / V (key slots; slots past L and columns past d are zero) and of Wo: contiguous along k, 16-byte pieces
    auto stage_rows = [&](f16 (*dst)[kFoldKMax + 8], const f16* base, long long row_stride, int valid_rows) {
        for (int idx = tid; idx < 80 * kc; idx += 256) {
            const int r = idx / kc, ch = idx - r * kc;
            const f16x8 v = (r < valid_rows && ch * 8 < d) ? ldg8(base + (long long)r * row_stride + ch * 8) : zero8();
            *reinterpret_cast<f16x8*>(&dst[r][ch * 8]) = v;
        }
    };
    if (!second) {
        stage_rows(sa, p.kv + (long long)img * p.L * p.ld_kv + p.k_off + h * d, p.ld_kv, p.L);
        // Wq'[(h, k)][c]: contiguous along c -> transposed into [c][k]
        for (int idx = tid; idx < KP * 10; idx += 256) {
            const int k = idx / 10, ch = idx - k * 10;
            const f16x8 v = k < d ? ldg8(p.wq + (long long)(h * d + k) * p.ldq + tc * 80 + ch * 8) : zero8();
#pragma unroll
            for (int e = 0; e < 8; ++e) sb[ch * 8 + e][k] = v[e];
        }
    } else {
        stage_rows(sa, p.wo + (long long)(tc * 80) * p.ldwo + h * d, p.ldwo, 80);
        stage_rows(sb, p.kv + (long long)img * p.L * p.ld_kv + p.v_off + h * d, p.ld_kv, p.L);
    }
    __syncthreads();
    const int fr = lane & 15, fg = lane >> 4;
    for (int rb = wave; rb < 5; rb += 4) { // (wave-uniform)
        f32x4 acc[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KP / 32; ++ks) {
            const f16x8 fa = *reinterpret_cast<const f16x8*>(&sa[rb * 16 + fr][ks * 32 + fg * 8]);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const f16x8 fb = *reinterpret_cast<const f16x8*>(&sb[j * 16 + fr][ks * 32 + fg * 8]);
                acc[j] = mfma16(fb, fa, acc[j]); // lane: row rb*16 + fr, columns j*16 + 4 fg + r
            }
        }
        const int m = rb * 16 + fr;
        f16* row = second ? p.w2 + ((long long)img * p.C + tc * 80 + m) * NK + h * 80
                          : p.w1 + ((long long)(img * p.heads + h) * 80 + m) * p.C + tc * 80;
        const float al = second ? 1.0f : p.alpha;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            f16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (f16)(al * acc[j][r]);
            *reinterpret_cast<f16x4*>(row + j * 16 + fg * 4) = o;
        }
    }
}

} // namespace

extern "C" int sdod_attention_f16(const void* q, const void* k, const void* v, void* out, int batch, int heads, int lq,
                                  int lk, int d, int ldq, int ldk, int ldv, int ldo, float scale, int causal,
                                  void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(q && k && v && out, "null pointer");
    SDOD_REQUIRE(batch > 0 && heads > 0 && lq > 0 && lk > 0, "bad shape");
    SDOD_REQUIRE(d == 40 || d == 64 || d == 80 || d == 160, "head dim must be one of 40, 64, 80, 160");
    SDOD_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "row strides must keep 16-byte alignment");
    SDOD_REQUIRE(ldq >= heads * d && ldk >= heads * d && ldv >= heads * d && ldo >= heads * d, "row stride < heads*d");
    SDOD_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) == 0 && ((uintptr_t)out & 7) == 0, "misaligned pointer");
    AttnP p{};
    p.q = (const f16*)q; p.k = (const f16*)k; p.v = (const f16*)v; p.out = (f16*)out;
    p.B = batch; p.H = heads; p.Lq = lq; p.Lk = lk;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo;
    p.scale_log2 = scale * 1.4426950408889634f;
    p.causal = causal;
    const bool no_tr = std::getenv("SDOD_ATTN_NO_TR") != nullptr; // debugging aid: scalar LDS reads instead of ds_read_b64_tr_b16
    const bool tr = !no_tr;
    bool big = lq >= 2048 && d != 160; // two query tiles per wave once there is enough work to fill the chip
    if (const char* e = std::getenv("SDOD_ATTN_QT")) big = e[0] == '2' && d != 160; // developer override (tools/attn_bench.py)
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    switch (d) {
    case 40: e = attn_dispatch<40>(p, big, tr, st); break;
    case 64: e = attn_dispatch<64>(p, big, tr, st); break;
    case 80: {
        // one workgroup per CU or less and at least four key tiles: split the keys inside the workgroup (two groups of four waves)
        // (19.4 vs 21.3 us at 32x32, tools/attn_bench.py; SDOD_ATTN_KVS=0 keeps the four-wave form)
        static const bool kvs_off = [] { const char* e2 = std::getenv("SDOD_ATTN_KVS"); return e2 && e2[0] == '0'; }();
        const long long wgs = (long long)((lq + 63) / 64) * heads * batch;
        if (!big && tr && !kvs_off && !causal && wgs <= 512 && lk >= 256) e = attn_launch<80, 1, true, 2>(p, st);
        else e = attn_dispatch<80>(p, big, tr, st);
        break;
    }
    default: e = attn_dispatch<160>(p, false, tr, st); break;
    }
    SDOD_HIP_CHECK(e);
    return 0;
    SDOD_CATCH
}

extern "C" int sdod_xattn_fold_f16(const void* kv, int ld_kv, int k_off, int v_off, int n_img, int L, const void* wq, int ldq,
                                   const void* sq, const void* tq, const void* wo, int ldwo, int heads, int d, float scale, void* w1,
                                   void* s1, void* t1, void* w2, void* stream) {
    SDOD_TRY
    SDOD_REQUIRE(kv && wq && sq && tq && wo && w1 && s1 && t1 && w2, "null pointer");
    SDOD_REQUIRE(n_img > 0 && heads > 0 && d > 0 && d <= kFoldKMax && d % 8 == 0 && L > 0 && L <= 80, "bad shape (d <= 160, d % 8 == 0, L <= 80)");
    const int C = heads * d, NK = heads * 80;
    SDOD_REQUIRE(C % 80 == 0 && ldq >= C && ldwo >= C && ld_kv >= C, "heads * d must be a multiple of 80; row strides >= heads * d");
    SDOD_REQUIRE(ld_kv % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && ldq % 8 == 0 && ldwo % 8 == 0 &&
                     (((uintptr_t)kv | (uintptr_t)wq | (uintptr_t)wo | (uintptr_t)w1 | (uintptr_t)w2) & 15) == 0,
                 "operands must keep 16-byte alignment");
    FoldP p{};
    p.kv = (const f16*)kv; p.ld_kv = ld_kv; p.k_off = k_off; p.v_off = v_off; p.L = L; p.n_img = n_img;
    p.wq = (const f16*)wq; p.ldq = ldq; p.sq = (const float*)sq; p.tq = (const float*)tq;
    p.wo = (const f16*)wo; p.ldwo = ldwo;
    p.heads = heads; p.d = d; p.C = C;
    p.alpha = scale * 1.4426950408889634f; // the softmax of the score GEMM's epilogue works in the exp2 domain
    p.w1 = (f16*)w1; p.s1 = (float*)s1; p.t1 = (float*)t1; p.w2 = (f16*)w2;
    p.tiles_c = C / 80;
    p.n_w = n_img * heads * p.tiles_c;
    const int blocks = 2 * p.n_w + (n_img * NK + 255) / 256;
    SDOD_LAUNCH(xattn_fold_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    SDOD_HIP_CHECK(hipGetLastError());
    return 0;
    SDOD_CATCH
}

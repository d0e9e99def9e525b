/*
 * sdod_hip.h -- kernel-level C ABI of the MI355X (gfx950) hot path.
 *
 * Every entry point takes plain device pointers, sizes and a HIP stream handle
 * (void* == hipStream_t); no torch / C++ types cross this boundary.  Return value
 * is 0 on success, otherwise a libsdod status code (include/libsdod.h) with a
 * message retrievable through sdod_hip_last_error().
 *
 * What each group replaces in the reference (vaenyr/stable-diffusion-on-device):
 *   - sdod_group_norm_*    : the `sdod::GroupNorm` / `sdod::ParameterlessGroupNorm` custom op whose
 *                            schema is csrc/sdod_ops/config/group_norm.{xml,json} and whose Python
 *                            surface is sdod/efficient_gn.py:9-30 (no kernel exists in the reference)
 *   - sdod_gemm_f16, sdod_attention_f16, sdod_layer_norm_f16, ... : the arithmetic inside the opaque
 *                            QNN graphs executed by QnnGraph::execute (csrc/libsdod/src/qnn_context.cpp:711-713)
 *   - sdod_cfg_* / sdod_dpm_update / sdod_plms_* : the host-side CFG combine
 *                            (qnn_context.cpp:1018-1081 via context.cpp:359-373) and DPMSolver::update
 *                            (dpm_solver.cpp:136-181), moved on-device
 *   - sdod_timestep_features: context.cpp:257-274
 *   - sdod_image_to_u8      : context.cpp:392-395
 *
 * Layouts: activations are NHWC fp16 ("[N][H*W][C]", C contiguous); weights are [Cout][K] fp16 with
 * K contiguous (conv3x3: K = (r*3+s)*Cin + c, i.e. KRSC); bias / norm parameters are fp32.
 */
#ifndef SDOD_HIP_H
#define SDOD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifndef SDOD_API
#define SDOD_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

enum sdod_act { SDOD_ACT_NONE = 0, SDOD_ACT_SILU = 1, SDOD_ACT_GELU = 2, SDOD_ACT_QUICK_GELU = 3 };
/* SDOD_U8Q (graph parameters only): per-tensor affine uint8, the reference's QNN weight format (`quantize=8`, todlc.py:108;
 * qnn_context.cpp:1018-1033): payload = {float scale; int32 offset (<= 0); uint8 q[numel]}, real = (q + offset) * scale */
enum sdod_dtype { SDOD_F16 = 0, SDOD_F32 = 1, SDOD_U8Q = 2, SDOD_BF16 = 3 /* sdod_group_norm_nchw only */ };
enum sdod_a_mode { SDOD_A_ROWS = 0, SDOD_A_CONV3X3 = 1 /* NHWC gather: 3x3 pad 1 or 1x1 */ };

/* out[M][N] = act(alpha * A[M][K] . W[N][K]^T + bias + row_bias) + residual        (fp16 in/out, fp32 acc)
 * A is either a row-major matrix (SDOD_A_ROWS) or gathered on the fly from an NHWC image for a 3x3
 * pad-1 convolution (SDOD_A_CONV3X3): row m = (img, oy, ox), k = (r, s, c); the image may be the
 * channel-concatenation of two tensors (a, a2) and may be nearest-2x upsampled on the fly. */
typedef struct sdod_gemm_desc {
    const void* a;        /* fp16; rows: [M][lda]; conv: [n_img][h_in][w_in][c0] */
    const void* a2;       /* conv only: second concat source [n_img][h_in][w_in][c1], or NULL */
    const void* w;        /* fp16 [N][ldw] */
    const void* bias;     /* fp32 [N] (or [M] if bias_on_m), may be NULL */
    const void* row_bias; /* fp16 [M / rows_per_img][ld_row_bias] added per image (time embedding), may be NULL */
    const void* residual; /* fp16 [M][ldr], may be NULL */
    void* out;            /* fp16 [M][ldo] */
    void* workspace;      /* fp32 split-K slabs, >= split_k*M*N*4 bytes when split_k > 1 */
    size_t workspace_bytes;
    int M, N, K;
    int lda, ldw, ldo, ldr;
    int a_mode;
    int n_img, h_in, w_in, c0, c1; /* conv geometry: input (pre-upsample) */
    int stride;                    /* conv: 1 or 2 */
    int upsample;                  /* conv: 1 = nearest 2x before the conv */
    int ksize;                     /* conv: 3 (default when 0) or 1 */
    int rows_per_img;              /* for row_bias */
    int ld_row_bias;               /* row stride of row_bias (0 = N) */
    int act;
    float alpha;
    int bias_on_m;
    int split_k;                   /* 0 = auto; 1 = none; >1 = number of K slices (needs workspace) */
    int tile;                      /* 0 = auto; else forces a tile config (see gemm.hip) */
    /* --- fusions (LDS-DMA kernel family only) ---
     * geglu: W has N rows laid out as interleaved 16-row blocks [a(16) | gate(16)]...; out gets N/2 columns:
     *        out = (acc_a + bias_a) * gelu(acc_gate + bias_gate)    (ldm GEGLU: x * gelu(gate), fused into ff.net.0.proj)
     * tail segment: K columns [k_tail, K) of W multiply a SECOND, 1x1-gathered NHWC source (t0 | t1 channel concat) at the
     *        output pixel: out = conv3x3(a|a2) + conv1x1(t0|t1)  (ResBlock out_layers.3 + skip_connection in one GEMM);
     *        bias2 is added like bias.
     * ln: the rows of A are LayerNorm-ed on the fly (rows mode, K = normalised width, no split-K): the kernel sums
     *        x and x^2 of every row while the slabs pass through LDS and the epilogue applies
     *        out = rstd_m * (acc - mean_m * ln_s[n]) + bias[n]; the LayerNorm weight is pre-multiplied into W, ln_s[n] =
     *        sum_k W'[n][k] and bias[n] = sum_k beta[k] W[n][k] (+ linear bias) come from sdod_ln_fold_f16. */
    int geglu;
    int k_tail;                    /* 0 = no tail segment; else 9*(c0+c1) */
    const void* t0;
    const void* t1;
    int tc0, tc1;
    const void* bias2;
    int ln;
    const void* ln_s;              /* fp32 [N] */
    float ln_eps;
    int phase;                     /* split-K only: 0 = GEMM + reduce (default), 1 = GEMM slabs only, 2 = reduce + epilogue only
                                    * (lets a launch list time / profile the two kernels separately) */
    /* --- int8 weight streaming (LDS-DMA kernel family only; BASELINE config 5, the reference's `quantize=8` path,
     * todlc.py:105-108): w holds the affine-uint8 CODES of the reference's encoding real = (q + offset) * scale
     * (qnn_context.cpp:1018-1033), one byte per element, row stride ldw BYTES; the slab is streamed as bytes (half the
     * weight traffic of fp16) and expanded to the integer q + offset in fp16 (exact) on the fragment read.  Per output
     * column n: w_scale[n] = scale, w_off[n] = offset + 128 (fp32 holding an INTEGER, offset in [-1024, 0] -- the QNN
     * zero point of a uint8 tensor is in [-255, 0]), so fused parameter groups may mix tensors with different
     * encodings.  out = act(alpha * w_scale[n] * sum_k A (q + offset[n]) + bias ...).  Not with ln / k_tail. */
    int wq;
    const void* w_scale;           /* fp32 [N] */
    const void* w_off;             /* fp32 [N] */
    /* --- split-K without a reduce launch (halo-patch convolution tiles): a buffer of >= sdod_gemm_fixup_counters() 32-bit
     * words, ZEROED ONCE by its owner, used by one launch at a time (one per stream of concurrent callers) and left zeroed by
     * every launch.  With it, a phase-0 call of a split-K plan on such a tile is ONE launch: every K slice publishes its fp32
     * tile, the slice that arrives last at the tile's counter reduces (in slice order: bit-identical to the reduce kernel)
     * and runs the fused epilogue.  NULL: partial slabs + splitk_reduce_kernel as before. */
    void* fix_counters;
    /* XCD-aware tile order: 0 = chosen per shape (the number of n-tile panels in {1, 2, 4, 8} that minimises the bytes the
     * eight L2s fetch between them, panels * A + (8 / panels) * W); 1 / 2 / 4 / 8 force it (1 = m-major: every XCD reads all
     * of W; 8 = n-major: every XCD reads all of A).  Speed only. */
    int xcd_panels;
    /* --- per-image weights (rows mode, LDS-DMA ring tiles): the rows of image i = m / rows_per_img multiply the matrix at
     * w + i * w_img_stride (elements, a multiple of 8) and take bias / ln_s from + i * vec_img_stride floats (0: shared vectors).
     * rows_per_img must be a multiple of 32 dividing M; the plan only uses tiles whose rows divide rows_per_img.
     * softmax_cols = 80: the epilogue replaces every run of 80 output columns by its row softmax in the exp2 domain
     * (p = 2^(x - max) / sum; N a multiple of 160; columns to be ignored carry a bias of -30000).  Together they are the two
     * GEMMs of the FOLDED cross-attention (sdod_xattn_fold_f16): scores = LN(x) . (Wq^T K_h^T) with the softmax in the
     * epilogue, then out = P . (V_h Wo_h^T) + bias + residual. */
    int w_img_stride;
    int vec_img_stride;
    int softmax_cols;
} sdod_gemm_desc;

SDOD_API int sdod_gemm_f16(const sdod_gemm_desc* d, void* stream);
/* 1 when the call above reduces its split-K slices inside the GEMM launch (see fix_counters), else 0 */
SDOD_API int sdod_gemm_fixup(const sdod_gemm_desc* d);
/* counter words a fix_counters buffer must hold for any descriptor (one per output tile; 64 Ki covers M*N up to 2^28 at 64x64) */
SDOD_API size_t sdod_gemm_fixup_counters(void);
/* picks split_k / tile as the auto heuristic would; returns required workspace bytes */
SDOD_API size_t sdod_gemm_workspace_bytes(const sdod_gemm_desc* d);
/* which tile configuration (1: 128x128, 2: 128x64, 3: 64x64, 4: 256x16, 5: 64x128) and split-K factor the call would use */
SDOD_API int sdod_gemm_plan(const sdod_gemm_desc* d, int* tile, int* splits);
/* 1 when halo-patch convolution tile `tile` takes the descriptor (3x3, stride 1, tile rows divide the image, patch fits LDS);
 * 0 otherwise -- sdod_gemm_f16 refuses such a forced tile with LIBSDOD_INVALID_ARGUMENT.  Host-only. */
SDOD_API int sdod_gemm_halo_ok(const sdod_gemm_desc* d, int tile);
/* 1 when A-panel tile `tile` (gemm_apanel_kernel: a row panel x the whole K resident in LDS, n-tiles streamed past it) takes
 * the descriptor: rows mode, fp16 weights, K >= 192, panel + ring within 160 KiB of LDS, no split-K / row_bias / bias2 /
 * bias_on_m / tail segment.  Host-only. */
SDOD_API int sdod_gemm_panel_ok(const sdod_gemm_desc* d, int tile);
/* n-tile panels (1, 2, 4 or 8) of the XCD-aware tile order the call would use (see sdod_gemm_desc::xcd_panels).  Host-only. */
SDOD_API int sdod_gemm_xcd_panels(const sdod_gemm_desc* d);
/* number of tile configurations (valid `tile` values are 1..this); a tune table naming anything else is stale */
SDOD_API int sdod_gemm_num_tiles(void);
/* rows x columns of tile configuration `tile`; lds_dma = 1 for the LDS-DMA kernel family (tiles >= 6) */
SDOD_API int sdod_gemm_tile_shape(int tile, int* bm, int* bn, int* lds_dma);
/* template arguments of the kernel behind `tile`: {BM, BN, WM, WN, STAGES (0: register-staged gemm_kernel), SPEC, KSUB} */
SDOD_API int sdod_gemm_tile_info(int tile, int out[7]);
/* developer aid: average duration in ms of `iters` back-to-back launches (HIP events on `stream`) */
SDOD_API int sdod_gemm_time(const sdod_gemm_desc* d, void* stream, int iters, float* ms_avg);
/* Same, with cold caches: before every timed launch a sweep of `scratch` (>= 64 MiB; use >= 512 MiB to clear the 256 MiB
 * Infinity Cache) evicts the weights, then the activation operands are re-read so that they are cache-resident as they are
 * behind a producer launch.  This is what a GEMM meets inside a graph replay; the engine's tile autotuner ranks with it. */
SDOD_API int sdod_gemm_time_cold(const sdod_gemm_desc* d, void* stream, int iters, void* scratch, size_t scratch_bytes, float* ms_avg,
                                float* ms_min /* may be NULL */);

/* GroupNorm over NHWC [N][HW][C] (optionally the channel concat of x (c0) and x2 (c1)), G groups,
 * y = (x-mean)*rstd*w+b, optional SiLU.  dtype applies to x and y; weight/bias fp32 or NULL.
 * workspace: >= sdod_group_norm_workspace_bytes(N, G) bytes of fp32 scratch, 128-byte aligned, ZEROED ONCE by its owner before
 * the first call and never shared by two calls that may run concurrently (different streams: one workspace each): the
 * one-launch kernel for big maps (>= 5 MB) keeps the words of its grid barrier in the FIRST 4 KiB of it (a fixed offset, so
 * that one workspace may serve calls of different (N, G): no layout's partial sums reach them) and leaves them re-armed.
 * A grid barrier that cannot meet (barrier words clobbered; two such launches sharing one workspace or starving each other of
 * CUs) gives up after ~1 s instead of hanging the device, and its output is INVALID.  It is never silent: the kernel sets a
 * sticky per-device error word, after which sdod_group_norm_status() -- read it behind a host synchronisation -- and every
 * later sdod_group_norm_nhwc / sdod_graph_execute on that device return LIBSDOD_RUNTIME_ERROR until the owner has re-zeroed
 * the workspace and called sdod_group_norm_clear_error(). */
SDOD_API size_t sdod_group_norm_workspace_bytes(int n, int groups);
/* byte offsets of the workspace's parts for (n, groups): barrier lines, partial sums, statistics, pilot shifts, end */
SDOD_API int sdod_group_norm_layout(int n, int groups, size_t* sync_off, size_t* partial_off, size_t* stats_off, size_t* shift_off,
                                    size_t* end_off);
/* 0, or LIBSDOD_RUNTIME_ERROR once a one-launch GroupNorm on the current device has timed out at its grid barrier (sticky) */
SDOD_API int sdod_group_norm_status(void);
SDOD_API int sdod_group_norm_clear_error(void);
/* 1 = the single-launch kernel (whole image x channel set in LDS) handles this shape, 2 = statistics + apply launches */
SDOD_API int sdod_group_norm_launches(int hw, int c, int groups, int dtype);
/* which kernel the call below launches for this shape: 0 = one-launch grid-barrier kernel (maps >= 5 MB), 1 = one launch, a
 * workgroup per (image, group), 2 = one launch, small-map LDS kernel, 3 = statistics + apply launches; -1 = bad shape */
SDOD_API int sdod_group_norm_path(int n, int hw, int c0, int c1, int groups, int dtype);
SDOD_API int sdod_group_norm_nhwc(const void* x, const void* x2, void* y, const float* weight, const float* bias,
                                  int n, int hw, int c0, int c1, int groups, float eps, int silu, int dtype,
                                  void* workspace, void* stream);

/* The same operator on torch's default layout, NCHW "[N][C][spatial]" (what sdod.EfficientGN receives from a model that was
 * not converted to channels_last, efficient_gn.py:61-86): a group is one contiguous slab of (C / G) * spatial elements, so
 * there is no transpose and no constraint on C; fp16 / bf16 / fp32, y may be x.  Slabs up to 32 Ki elements take one launch
 * (read once); bigger ones a statistics + an apply launch and `workspace` (>= sdod_group_norm_nchw_workspace_bytes(n,
 * groups) bytes, no initialisation needed, one per stream). */
SDOD_API size_t sdod_group_norm_nchw_workspace_bytes(int n, int groups);
SDOD_API int sdod_group_norm_nchw(const void* x, void* y, const float* weight, const float* bias, int n, int c, long long spatial,
                                  int groups, float eps, int silu, int dtype, void* workspace, void* stream);

/* GroupNorm whose source-0 tensor is still in split-K form: the producing GEMM ran with phase = 1 (fp32 partial slabs only),
 * and this call does what its phase 2 (splitk_reduce + fused epilogue) would have done -- x = fp16(act(alpha * sum_s partial
 * + bias + bias2 + row_bias[image])) + residual, same arithmetic and rounding points -- while loading the group: x is
 * written to x_out ([n*hw][c0], dense) for its later readers and GroupNorm(+SiLU) of (x | x2) goes to y, in ONE launch.
 * sdod_gemm_reduce_info fills the descriptor from the GEMM's; sdod_group_norm_reduce_ok tells whether the one-launch
 * (image, group) kernel takes the shape (otherwise run phase 2 and the plain GroupNorm). */
typedef struct sdod_gn_reduce {
    const void* partial;   /* fp32 [splits][M][N] */
    int splits;
    size_t slab_floats;    /* M * N */
    const void* bias;      /* fp32 [N] or NULL */
    const void* bias2;     /* fp32 [N] or NULL */
    const void* row_bias;  /* fp16 [n_img][ld_row_bias] or NULL (rows_per_img must equal hw) */
    int ld_row_bias;
    const void* residual;  /* fp16 [M][ldr] or NULL */
    int ldr;
    void* x_out;           /* fp16 [M][N] */
    float alpha;
    int act;
    int M, N;
} sdod_gn_reduce;
SDOD_API int sdod_gemm_reduce_info(const sdod_gemm_desc* d, sdod_gn_reduce* out);
SDOD_API int sdod_group_norm_reduce_ok(int hw, int c0, int c1, int groups);
SDOD_API int sdod_group_norm_reduce_nhwc(const sdod_gn_reduce* red, const void* x2, void* y, const float* weight,
                                         const float* bias, int n, int hw, int c0, int c1, int groups, float eps, int silu,
                                         void* stream);

/* Folds a LayerNorm (gamma, beta over K) into the Linear that consumes it, in place: t_out[n] = sum_k beta[k]*W[n][k]
 * (+ bias_in[n]); W[n][k] <- fp16(W[n][k]*gamma[k]); s_out[n] = sum_k W'[n][k].  Used once per weight at graph build. */
SDOD_API int sdod_ln_fold_f16(void* w, int n, int k, int ldw, const float* gamma, const float* beta, const float* bias_in,
                              float* s_out, float* t_out, void* stream);
/* Composes two Linear layers that follow each other with nothing in between, y = P (W x + bw) + bp, into one (one-time, at
 * graph build): c[o][k] = sum_j p[o][j] w[j][k] (fp32 accumulate, rounded to fp16 once), bias_out[o] = sum_j p[o][j] bw[j]
 * + bp[o].  p: fp16 [n_out][ldp] (n_mid columns), w: fp16 [n_mid][ldw] (k columns), c: fp16 [n_out][ldc]; c must not alias
 * p or w.  The UNet uses it for transformer_blocks.0.ff.net.2 -> proj_out (analyze_results.py:69-79 lists them as two ops). */
SDOD_API int sdod_compose_linear_f16(const void* p, int ldp, const void* w, int ldw, void* c, int ldc, int n_out, int n_mid, int k,
                                     const float* bias_w, const float* bias_p, float* bias_out, void* stream);
/* LayerNorm over the last dim of fp16 [M][C] rows, fp32 weight/bias (either may be NULL); C % 8 == 0, C <= 3072. */
SDOD_API int sdod_layer_norm_f16(const void* x, void* y, const float* weight, const float* bias, int m, int c,
                                 float eps, void* stream);

/* Fused attention: out[b][q][h*D..] = softmax(scale * Q K^T (+causal mask)) V, never materialising scores.
 * q: [B][Lq][ldq], k/v: [B][Lk][ldk]/[ldv], head h at column offset h*D.  D in {40, 64, 80, 160}. */
SDOD_API int sdod_attention_f16(const void* q, const void* k, const void* v, void* out, int batch, int heads,
                                int lq, int lk, int d, int ldq, int ldk, int ldv, int ldo, float scale, int causal,
                                void* stream);

/* Folded cross-attention, the once-per-prompt part.  The text context is constant over the sampler run, so to_q and to_out can
 * be multiplied into K and V: scores_h = LN(x) . W1_h with W1_h = Wq_h^T K_h^T, out = [P_1 | .. | P_H] . [W2_1 ; .. ; W2_H] with
 * W2_h = V_h Wo_h^T -- two GEMMs per evaluation (sdod_gemm_desc: w_img_stride / vec_img_stride / softmax_cols) instead of
 * Linear + attention + Linear (the reference's /attn2/to_q, /attn2/MatMul, /attn2/to_out.0 ops, analyze_results.py:69-79).
 * kv: fp16 [n_img * L][ld_kv], K at column k_off, V at v_off (heads * d columns each), L <= 80 keys; wq: [C][ldq] with the
 * LayerNorm weight folded in, sq / tq its fold vectors (sdod_ln_fold_f16); wo: [C][ldwo]; scale = the softmax scale.
 * Writes w1 fp16 [n_img][heads*80][C] (scaled by scale * log2 e), s1 / t1 fp32 [n_img][heads*80] (padding columns: s = 0,
 * t = -30000), w2 fp16 [n_img][C][heads*80] (padding columns zero). */
SDOD_API int sdod_xattn_fold_f16(const void* kv, int ld_kv, int k_off, int v_off, int n_img, int L, const void* wq, int ldq,
                                 const void* sq, const void* tq, const void* wo, int ldwo, int heads, int d, float scale, void* w1,
                                 void* s1, void* t1, void* w2, void* stream);

/* Row softmax on fp16 [M][N] in place semantics allowed (y may equal x); fp32 math. */
SDOD_API int sdod_softmax_rows_f16(const void* x, void* y, int m, int n, void* stream);

/* Elementwise / glue (fp16 unless stated) */
SDOD_API int sdod_geglu_f16(const void* x, void* y, int m, int c, void* stream); /* x:[M][2c] -> y = x[:, :c]*gelu(x[:, c:]) */
SDOD_API int sdod_act_f16(const void* x, void* y, size_t n, int act, void* stream);
SDOD_API int sdod_add_f16(const void* a, const void* b, void* y, size_t n, void* stream);
SDOD_API int sdod_concat_channels_f16(const void* a, const void* b, void* y, size_t rows, int c0, int c1, void* stream);
SDOD_API int sdod_im2col3x3_small_f16(const void* x, void* y, int n_img, int h, int w, int c, int kpad, void* stream);
/* sdod_latent_prep_f16 (w = NULL) followed by sdod_im2col3x3_small_f16 in one launch: x NCHW fp32 [n][c][h][w] -> the im2col matrix
 * [n*h*w][kpad] fp16 of a 3x3 pad-1 convolution (k = tap * c + channel, zero beyond 9c), values (fp16)(x * scale); kpad % 8 == 0 */
SDOD_API int sdod_latent_im2col_f16(const float* x, void* y, int n_img, int h, int w, int c, int kpad, float scale, void* stream);

/* The UNet's input convolution in one launch (the reference's /input_blocks.0/Conv, analyze_results.py:69-79): x NCHW fp32
 * [n][c][h][w] (scaled by `scale`, rounded to fp16 as sdod_latent_im2col_f16 does) -> 3x3 pad-1 convolution with w fp16 [cout][64]
 * (k = tap * c + channel, zero beyond 9 c: the PK_CONV3_SMALL packing) + bias fp32 [cout] -> y NHWC fp16 [n*h*w][cout].
 * 9 * c <= 64; cout in {64, 128, 256, 320}.  Equals sdod_latent_im2col_f16 followed by the K = 64 sdod_gemm_f16. */
SDOD_API int sdod_conv_in_f16(const float* x, const void* w, const float* bias, void* y, int n_img, int h, int wd, int c, int cout,
                              float scale, void* stream);
SDOD_API int sdod_nchw_f32_to_nhwc_f16(const float* x, void* y, int n, int c, int hw, float scale, void* stream);
/* y(NHWC fp16)[img][pix][o] = sum_c w[o][c]*(scale*x(NCHW fp32)[img][c][pix]) + b[o]; w/b fp32 [c][c]/[c] or NULL
 * (identity).  Folds ldm's z/0.18215 and first_stage_model.post_quant_conv into the layout change. */
SDOD_API int sdod_latent_prep_f16(const float* x, const float* w, const float* b, void* y, int n, int c, int hw,
                                  float scale, void* stream);
SDOD_API int sdod_nhwc_f16_to_nchw_f32(const void* x, float* y, int n, int c, int hw, void* stream);
SDOD_API int sdod_embedding_f16(const int32_t* ids, const void* table, const void* pos, void* y, int rows, int seq,
                                int c, void* stream);
/* context.cpp:257-274: out[i][j]=cos(t_i*f_j), out[i][half+j]=sin(t_i*f_j), f_j=exp(-ln(1e4)*j/half); fp16 out */
SDOD_API int sdod_timestep_features_f16(const float* t, void* y, int n, int dim, void* stream);

/* Inputs of one guided UNet evaluation in one launch (host loop: context.cpp:348-349): x (fp32, lat_count values = n
 * images) -> x_dst written `reps` times back to back; temb_row (fp16, temb_width) -> temb_dst, `temb_reps` rows. */
SDOD_API int sdod_stage_unet_inputs(const float* x, float* x_dst, size_t lat_count, int reps, const void* temb_row,
                                    void* temb_dst, size_t temb_width, int temb_reps, void* stream);
/* x_T ~ N(0, I) on the device (replaces the host generator of context.cpp:333-334 for throughput runs): Philox4x32-10
 * keyed by `seed`, counter (i / 4, stream_id), Box-Muller on the word pairs (0,1), (2,3) with u = ((w >> 8) + 0.5) / 2^24.
 * words_out (optional, uint32[count]) receives the raw Philox words for bit-exact checks. */
SDOD_API int sdod_randn_f32(float* out, uint32_t* words_out, size_t count, uint64_t seed, uint64_t stream_id, void* stream);

/* Classifier-free guidance on the batched UNet output.  eps_nhwc: fp16 NHWC [2n][hw][c]; rows [0,n) are the
 * unconditional half when uncond_first != 0 (ldm ordering), else the conditional half.  e_out: fp32 NCHW [n][c][hw].
 *   mode 0: e = g*e_cond + (1-g)*e_uncond         -- the reference driver (context.cpp:359-373, libsdod.h:88)
 *   mode 1: e = e_uncond + g*(e_cond - e_uncond)  -- ldm's PLMS/DDIM samplers (config 1's CPU reference) */
SDOD_API int sdod_cfg_combine(const void* eps_nhwc, float* e_out, int n, int c, int hw, float guidance,
                              int uncond_first, int mode, void* stream);
/* DPM-Solver++(2M) update, dpm_solver.cpp:136-181, on fp32 device vectors of `count` elements, same operation
 * order and no FMA contraction (bit-identical to the host loop for identical inputs):
 *   y = (x - sigma_s*eps)/alpha_s ; x = sigma_ratio*x (+ c_prev*y_prev if order==2) + c_cur*y ; y_prev = y
 * with sigma_ratio = sigma[s+1]/sigma[s], c_prev = alpha[s+1]*phi[s+1]*i2r[s+1],
 * c_cur = -alpha[s+1]*phi[s+1] (order 1) or -alpha[s+1]*phi[s+1]*(1+i2r[s+1]) (order 2), all from the host tables. */
SDOD_API int sdod_dpm_update(float* x, const float* eps, float* y_prev, size_t count, int order, float sigma_s,
                             float alpha_s, float sigma_ratio, float c_prev, float c_cur, void* stream);
/* ldm DDIM/PLMS step (eta=0): x0 = (x - sqrt(1-abar_t)*e)/sqrt(abar_t); x = sqrt(abar_prev)*x0 + dir_coef*e */
SDOD_API int sdod_ddim_step_f32(float* x, const float* e, size_t count, float sqrt_one_minus_at, float sqrt_at,
                                float sqrt_a_prev, float dir_coef, void* stream);
/* out = (c0*e0 + c1*e1 + c2*e2 + c3*e3)/div, left to right (PLMS Adams-Bashforth combinations); e1..e3 may be NULL */
SDOD_API int sdod_lincomb4_f32(float* out, const float* e0, const float* e1, const float* e2, const float* e3,
                               float c0, float c1, float c2, float c3, float div, size_t count, void* stream);

/* One PLMS step behind a UNet evaluation in one launch (Python host loop, sdod/amd/pipeline.py: sample_plms; the reference's
 * sampler is ldm's PLMSSampler.p_sample_plms): e_t = CFG(eps) [sdod_cfg_combine; optional v -> eps conversion
 * e_t = vc0 * e + vc1 * x], e' = (c0 e_t + c1 old1 + c2 old2 + c3 old3) / div [sdod_lincomb4_f32], x <- DDIM step with e'
 * [sdod_ddim_step_f32], then the next evaluation's inputs [sdod_stage_unet_inputs: x into x_stage[stage_reps][..], temb_row into
 * temb_dst[temb_reps][..]]; the same fp32 operations in the same order as those four calls, bit for bit.  old1..3, x_stage,
 * temb_row may be NULL. */
typedef struct sdod_plms_update_args {
    const void* eps_nhwc;   /* fp16 [2n][hw][c] */
    float* e_out;           /* fp32 [n][c][hw]: e_t, kept by the caller as history */
    float* x;               /* fp32 [n][c][hw], updated in place */
    const float* old1; const float* old2; const float* old3;
    float* x_stage;         /* fp32 [stage_reps][n][c][hw] or NULL */
    const void* temb_row;   /* fp16 [temb_width] or NULL */
    void* temb_dst;         /* fp16 [temb_reps][temb_width] */
    int n, c, hw, uncond_first, mode, v_pred, stage_reps, temb_width, temb_reps;
    float guidance, vc0, vc1, c0, c1, c2, c3, div, sqrt_one_minus_at, sqrt_at, sqrt_a_prev, dir_coef;
} sdod_plms_update_args;
SDOD_API int sdod_plms_update(const sdod_plms_update_args* a, void* stream);

/* The reference driver's per-step arithmetic behind a UNet evaluation in one launch (src/context.cpp:359-373: CFG; src/dpm_solver.cpp:
 * 139-180: DPM-Solver++(2M) update; :348-352: staging of the next step's inputs): sdod_cfg_combine + sdod_dpm_update +
 * sdod_stage_unet_inputs, the same fp32 operations in the same order, bit for bit.  e_out, x_stage, temb_row may be NULL. */
typedef struct sdod_dpm_step_args {
    const void* eps_nhwc;   /* fp16 [2n][hw][c] */
    float* e_out;           /* fp32 [n][c][hw] or NULL */
    float* x;               /* fp32 [n][c][hw], updated in place */
    float* y_prev;          /* fp32 [n][c][hw], read (order 2) and replaced */
    float* x_stage;         /* fp32 [stage_reps][n][c][hw] or NULL */
    const void* temb_row;   /* fp16 [temb_width] or NULL */
    void* temb_dst;         /* fp16 [temb_reps][temb_width] */
    int n, c, hw, uncond_first, mode, order, stage_reps, temb_width, temb_reps;
    float guidance, sigma_s, alpha_s, sigma_ratio, c_prev, c_cur;
} sdod_dpm_step_args;
SDOD_API int sdod_dpm_step(const sdod_dpm_step_args* a, void* stream);
/* img: fp16 NHWC [n][hw][3] -> uint8 HWC per image, f = a*v+b, truncating cast:
 *   mode 0: clamp(255*f, 0, 255)   (context.cpp:392-395; a=1,b=0 is the reference's already-[0,1] convention)
 *   mode 1: 255*clamp(f, 0, 1)     (ldm txt2img with a=0.5, b=0.5) */
SDOD_API int sdod_image_to_u8(const void* img, uint8_t* out, size_t count, float a, float b, int mode, void* stream);

SDOD_API const char* sdod_hip_last_error(void);
SDOD_API int sdod_hip_device_info(int* cu_count, size_t* hbm_bytes, char* arch, int arch_len);

/* touch every 128-byte line of [ptr, ptr + bytes) once (a few workgroups on `stream`): the engine pulls the weights of the
 * deep, weight-heavy GEMMs towards the Infinity Cache a few launches ahead of their consumer (engine.hip: Graph::run_ops) */
SDOD_API int sdod_l2_prefetch(const void* ptr, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDOD_HIP_H */

/*
 * sdod_engine.h -- graph-level C ABI of the MI355X txt2img engine.
 *
 * A "graph" here plays the role the serialized QNN graphs play in the reference: the reference's
 * Context loads four graphs -- "unet.serialized", "text_encoder.serialized", "vae_decoder.serialized",
 * "temb" (csrc/libsdod/src/context.cpp:105) -- allocates their I/O tensors (context.cpp:201-218) and
 * calls QnnGraph::execute() on them (qnn_context.cpp:711-713).  Here each graph is a static launch list
 * of the hand-written gfx950 kernels of include/sdod_hip.h over a weight arena and an activation arena
 * in HBM, optionally replayed as one hipGraph.
 *
 * I/O slots (device memory owned by the graph; sdod_graph_io() returns pointer + size):
 *   UNET          in0 x    fp32 NCHW [B][4][H][W]        (context.cpp:214  x = unet.allocate_input(0))
 *                 in1 temb fp16 [B][E]                   (context.cpp:215  t = unet.allocate_input(1)); E = width of the
 *                                                        TEMB graph's output: the time-MLP output pushed through every
 *                                                        ResBlock's emb_layers projection (all of it depends on t only,
 *                                                        so it is computed once per step and cached like context.cpp:276-278)
 *                 in2 ctx  fp16 [B][77][ctx_dim]         (context.cpp:216  p_cond / p_uncond = input(2))
 *                 out0 e   fp16 NHWC [B][H][W][4]        (context.cpp:218  e = unet.allocate_output(0))
 *   TEMB          in0 t    fp32 [B]                      (model time, dpm_solver.cpp:115)
 *                 out0     fp16 [B][E]                   (context.cpp:257-278: sinusoid + temb graph, + emb_layers)
 *   TEXT_ENCODER  in0 ids  int32 [B][77]                 (context.cpp:207 tokens)
 *                 out0     fp16 [B][77][ctx_dim]         (context.cpp:208 p)
 *   VAE_DECODER   in0 z    fp32 NCHW [B][4][H][W]        (context.cpp:220 y)
 *                 out0 img fp16 NHWC [B][8H][8W][3] in [-1,1]  (context.cpp:221 img)
 *
 * Parameters are addressed by their CompVis-ldm / HF-CLIP state-dict names (without the
 * `model.diffusion_model.` / `first_stage_model.` / `cond_stage_model.transformer.` prefixes) and are
 * given in their canonical PyTorch layouts (conv [Cout][Cin][kh][kw], linear [out][in]); the engine
 * repacks them to KRSC fp16 in HBM.  All functions return 0 or a libsdod status code; the message is
 * available from sdod_hip_last_error().
 */
#ifndef SDOD_ENGINE_H
#define SDOD_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifndef SDOD_API
#define SDOD_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

enum sdod_graph_kind { SDOD_GRAPH_UNET = 0, SDOD_GRAPH_VAE_DECODER = 1, SDOD_GRAPH_TEXT_ENCODER = 2, SDOD_GRAPH_TEMB = 3 };

typedef struct sdod_model_config {
    int latent_channels; /* 4 */
    int latent_h;        /* 64 (SD1.x 512px), 96 (SD2.1 768px) */
    int latent_w;
    int model_channels;  /* 320 */
    int context_dim;     /* 768 (SD1.x), 1024 (SD2.x) */
    int context_len;     /* 77 */
    int num_heads;       /* 8 (SD1.x: head dim = C/8); 0 = use head_dim */
    int head_dim;        /* 64 (SD2.x); ignored when num_heads > 0 */
    int vocab_size;      /* 49408 */
    int text_layers;     /* 12 */
    int text_heads;      /* 12 */
    int vae_channels;    /* 128 */
    int linear_proj;     /* 0: transformer proj_in / proj_out are 1x1 convs [C,C,1,1] (SD1.x); 1: Linear [C,C] (SD2.x
                          * `use_linear_in_transformer`) -- same arithmetic, different checkpoint shapes */
    int text_arch;       /* TEXT_ENCODER graph: 0 = CLIP ViT-L/14 as HF `CLIPTextModel` names it (quick-GELU, text_layers
                          * blocks, last_hidden_state); 1 = OpenCLIP text tower as open_clip names it (`transformer.resblocks.N.*`,
                          * fused `attn.in_proj_*`, erf GELU) stopped after text_layers blocks + ln_final: SD2.x conditions on the
                          * PENULTIMATE block of ViT-H/14 (24 blocks in the checkpoint, text_layers = 23) */
    int weight_quant;    /* 0: GEMM weights live in HBM as fp16 (an SDOD_U8Q checkpoint tensor is dequantised once at load);
                          * 1 (UNET / TEMB graphs): every conv / linear weight must be given as SDOD_U8Q and STAYS affine uint8 in
                          * HBM (half the weight footprint and stream, BASELINE config 5 "int8 weight quant (mirrors QNN quant
                          * path)"): the GEMM expands the codes on the fragment read (sdod_gemm_desc.wq).  Costs two fusions the
                          * fp16 build has: LayerNorm is a launch again (its gamma cannot be folded into integer codes) and the
                          * ResBlock skip 1x1 conv is its own GEMM (its tensor has its own scale / offset).
                          * 2 ("where it pays"): same checkpoint format, but the codes are streamed only by the blocks whose GEMMs
                          * have <= 128 rows; every other block's tensors are dequantised once at load and the block is built
                          * exactly as with weight_quant = 0.  On MI355X no GEMM of the UNet is weight-bandwidth bound at batch 2,
                          * so the uint8 kernels lose wherever there are >= 288 rows and tie at 72
                          * (profiles/r03_config5_op_tables.txt): this is the setting that is never slower than fp16. */
} sdod_model_config;

SDOD_API void sdod_model_config_sd14(sdod_model_config* cfg);
/* SD v2.1-768 UNet / VAE shapes (BASELINE config 5): 96x96 latent, context 1024, 64-wide heads (5/10/20/20 of them),
 * linear transformer projections, OpenCLIP ViT-H/14 text tower (width 1024, 16 heads, 23 of its 24 blocks). */
SDOD_API void sdod_model_config_sd21(sdod_model_config* cfg);

SDOD_API int sdod_graph_create(void** graph, int kind, const sdod_model_config* cfg, int batch);
SDOD_API int sdod_graph_destroy(void* graph);

/* parameter table (fixed by kind + config) */
SDOD_API int sdod_graph_num_params(void* graph);
SDOD_API int sdod_graph_param_info(void* graph, int index, const char** name, int* ndim, int64_t shape[4]);
/* data: host pointer in canonical layout, dtype SDOD_F32 or SDOD_F16 (include/sdod_hip.h), numel from shape */
SDOD_API int sdod_graph_set_param(void* graph, const char* name, const void* data, int dtype, const int64_t* shape, int ndim);
/* load every parameter from a .sdodw container (see DESIGN.md "weight file"); prefix is prepended to graph names */
SDOD_API int sdod_graph_load_file(void* graph, const char* path, const char* prefix);
/* checks that every parameter is set, sizes and allocates the activation arena, builds the launch list */
SDOD_API int sdod_graph_finalize(void* graph);
SDOD_API int sdod_graph_io(void* graph, int is_output, int index, void** device_ptr, size_t* bytes);
/* run once on `stream`.  flags bit 0 (SDOD_EXEC_HIP_GRAPH): the launch list is captured on first use and replayed as one
 * hipGraph; bit 1 (SDOD_EXEC_STATIC_UNCHANGED): the graph's static inputs (UNET in2, the text context, which is constant
 * over a sampler run) have not changed since the previous execute, so the launches depending only on them are skipped */
enum sdod_exec_flags { SDOD_EXEC_HIP_GRAPH = 1, SDOD_EXEC_STATIC_UNCHANGED = 2 };
SDOD_API int sdod_graph_execute(void* graph, void* stream, int flags);
/* Device-side failures a launch cannot report itself: LIBSDOD_RUNTIME_ERROR once a one-launch GroupNorm of any graph on this
 * device has timed out at its grid barrier (sdod_group_norm_status, include/sdod_hip.h) -- call it behind a host
 * synchronisation of the stream the graph ran on; sdod_graph_execute makes the same check on entry. */
SDOD_API int sdod_graph_check(void* graph);
/* launch list introspection + per-launch timing (HIP events on `stream`, eager, averaged over iters runs after one
 * warm-up): label = kernel family/variant ("gemm_t2", "gemm_t3_splitk", "attn_d40", "group_norm", ...), flops/bytes =
 * algorithmic work of that launch.  bench.py derives its roofline block from these. */
SDOD_API int sdod_graph_num_ops(void* graph);
SDOD_API int sdod_graph_op_info(void* graph, int index, const char** label, double* flops, double* bytes);
SDOD_API int sdod_graph_op_detail(void* graph, int index, const char** detail); /* shape string of launch `index` */
SDOD_API int sdod_graph_profile(void* graph, void* stream, int iters, float* ms_out, int n);
SDOD_API int sdod_graph_stats(void* graph, size_t* weight_bytes, size_t* arena_bytes, int* num_launches, double* flops);
/* provenance of the launch list: how many distinct GEMM shapes took their tile from the tune table(s) (the shipped
 * tune/gfx950.tune, then $SDOD_TUNE_CACHE) and how many were timed in this process (0 = the list is reproducible across
 * processes); table_path receives the path(s) consulted */
SDOD_API int sdod_graph_tune_info(void* graph, int* from_table, int* tuned_in_process, char* table_path, int cap);

#ifdef __cplusplus
}
#endif
#endif /* SDOD_ENGINE_H */

// engine.h -- the graph engine behind include/sdod_engine.h: weight arena, activation arena, launch list.
// It replaces, for MI355X, what QnnBackend/QnnGraph/QnnTensor do for the Hexagon HTP in the reference
// (csrc/libsdod/src/qnn_context.{h,cpp}): "load a graph, own its I/O tensors, execute it".
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "host_util.h"
#include "sdod_engine.h"
#include "sdod_hip.h"

namespace sdod {

typedef _Float16 f16;

enum ParamKind {
    PK_CONV3,       // [Cout][Cin][3][3] -> fp16 [Cout][(r*3+s)*Cin + c]
    PK_CONV3_SMALL, // Cin < 64: same order, K zero-padded to 64
    PK_CONV1,       // [Cout][Cin][1][1] -> fp16 [Cout][Cin]
    PK_LINEAR,      // [out][in] -> fp16
    PK_EMBED,       // [rows][dim] -> fp16
    PK_VEC,         // [n] -> fp32
    PK_MAT_F32,     // small matrix kept in fp32 (post_quant_conv 4x4)
    PK_LINEAR_GEGLU, // [2H][in] -> fp16, rows interleaved in 16-row blocks [value | gate] for the fused GEGLU epilogue
    PK_VEC_GEGLU,    // [2H] -> fp32, same interleave
};

struct Param {
    std::string name;
    std::vector<int64_t> shape; // canonical (PyTorch) shape
    ParamKind kind;
    int group = -1;
    int ld = 0;       // row stride in elements when the rows sit inside a wider matrix (0 = dense)
    int col_off = 0;  // first column inside that matrix
    int owner = -1;   // >= 0: this parameter aliases columns of params_[owner]'s allocation
    size_t dev_bytes = 0;
    char* dev = nullptr;
    bool set = false;
    bool quant = false; // kept as affine-uint8 codes in HBM (sdod_model_config.weight_quant); qrow = first row in qscale_/qoff_
    size_t qrow = 0;
};

// activation tensor: NHWC fp16 (sequences: h = 1, w = tokens)
struct Act {
    f16* p = nullptr;
    int n = 0, h = 0, w = 0, c = 0;
    int rows() const { return n * h * w; }
    size_t numel() const { return (size_t)rows() * c; }
};

struct IoSlot {
    void* ptr = nullptr;
    size_t bytes = 0;
};

class Graph {
public:
    Graph(int kind, const sdod_model_config& cfg, int batch);
    ~Graph();
    Graph(const Graph&) = delete;

    int num_params() const { return (int)params_.size(); }
    const Param& param(int i) const { return params_.at(i); }
    void set_param(const std::string& name, const void* data, int dtype, const int64_t* shape, int ndim);
    void load_file(const std::string& path, const std::string& prefix);
    void finalize();
    // skip_static: the inputs marked "static across a sampler run" (UNet: the text context) are unchanged since the
    // previous execute(), so the launches that depend only on them (all cross-attention K/V projections) are skipped
    void execute(hipStream_t st, bool use_hip_graph, bool skip_static = false);
    void check_health() const; // throws RUNTIME_ERROR once a GroupNorm grid barrier has timed out on this device (sticky)
    IoSlot io(bool output, int index) const;
    void stats(size_t* wbytes, size_t* abytes, int* launches, double* flops) const;

private:
    enum Mode { DECLARE, DRY, REAL };
    int kind_;
    sdod_model_config cfg_;
    int batch_;
    Mode mode_ = DECLARE;
    bool finalized_ = false;

    // ---- parameters
    std::vector<Param> params_;
    std::unordered_map<std::string, int> pindex_;
    std::vector<std::vector<int>> groups_;
    std::unordered_map<std::string, int> gindex_;
    char* weight_base_ = nullptr;
    size_t weight_bytes_ = 0;
    float* qscale_ = nullptr; // per weight ROW (output channel): scale and offset + 128 of its tensor's encoding, laid out in the
    float* qoff_ = nullptr;   // order of the weight arena, so that a fused parameter group is one contiguous run
    size_t qrows_ = 0;
    bool quant_mode() const { return cfg_.weight_quant != 0 && (kind_ == SDOD_GRAPH_UNET || kind_ == SDOD_GRAPH_TEMB); }
    // weight_quant = 1: every conv / Linear weight stays affine uint8 in HBM.  weight_quant = 2 ("where it pays"): the codes are
    // streamed only by the blocks whose GEMMs have at most kQuantAutoMaxRows rows; every other block's tensors are dequantised
    // ONCE at load (set_param: the reference's arithmetic, qnn_context.cpp:1018-1033) and the block is built exactly as in an
    // fp16 graph (LayerNorm fold, fused skip conv, composed ff.net.2 + proj_out).  Measured on MI355X (batch 2, SD v2.1 shapes,
    // profiles/r03_config5_op_tables.txt): no GEMM of the UNet is weight-BANDWIDTH bound -- the deepest 3x3 convolutions move
    // 15-30 MB of weights in 14-25 us, 1-1.5 TB/s -- so halving the stream buys nothing while the expansion is VALU work on an
    // issue-bound loop: uint8 loses 30-50 % on the convolutions with >= 1152 rows, ~1 % over the 288-row level (12x12 at 768 px),
    // and ties at 72 rows.  Hence 128: at config 5's own size the setting builds the fp16 graph from the uint8 checkpoint.
    // quant_decl_ is the decision for the block being built: identical in every build pass (it depends on the row count only).
    static constexpr int kQuantAutoMaxRows = 128;
    bool quant_decl_ = false;
    void quant_rows(int rows) { quant_decl_ = quant_mode() && (cfg_.weight_quant == 1 || rows <= kQuantAutoMaxRows); }
    void quant_global() { quant_decl_ = quant_mode(); }
    const float* qscale_of(int w) const { return params_[w].quant ? qscale_ + params_[w].qrow : nullptr; }
    const float* qoff_of(int w) const { return params_[w].quant ? qoff_ + params_[w].qrow : nullptr; }
    int P(const std::string& name, std::vector<int64_t> shape, ParamKind kind, const std::string& group = "");
    // parameter stored as a column block [col_off, col_off + K) of a wider [rows][ld] fp16 matrix (owner < 0: allocates it)
    int Pc(const std::string& name, std::vector<int64_t> shape, ParamKind kind, int ld, int col_off, int owner);
    template <typename T>
    const T* W(int idx) const { return reinterpret_cast<const T*>(params_[idx].dev); }
    void allocate_weights();

    // ---- activation arena (first-fit free list; DRY mode only measures the high-water mark)
    char* arena_base_ = nullptr;
    size_t arena_cap_ = 0, arena_high_ = 0;
    std::map<size_t, size_t> free_;   // offset -> size
    std::map<size_t, size_t> used_;   // offset -> size
    void arena_reset();
    f16* alloc(size_t halves);
    void release(const void* p);
    Act act(int n, int h, int w, int c);
    void release(const Act& a) { release(a.p); }

    // ---- scratch
    char* ws_ = nullptr;      // split-K slabs
    size_t ws_bytes_ = 0, ws_need_ = 0;
    char* gn_ws_ = nullptr;   // GroupNorm partials
    size_t gn_ws_bytes_ = 0, gn_ws_need_ = 0;

    // ---- io
    std::vector<IoSlot> inputs_, outputs_;
    void* io_alloc(std::vector<IoSlot>& v, size_t bytes);

    // ---- launch list
    struct Op {
        std::function<void(hipStream_t)> fn;
        std::string label;   // kernel family + variant, e.g. "gemm_t2", "gemm_t3_splitk", "attn_d40", "group_norm"
        double flops = 0;    // algorithmic FLOPs of this launch (2*M*N*K, 4*B*H*Lq*Lk*D)
        double bytes = 0;    // algorithmic HBM bytes of this launch (operands read once + result written once)
        std::string detail;  // shape, for the per-layer profile table (tools/unet_profile.py)
        const void* pf_ptr = nullptr; // weight-heavy GEMM: its weight matrix, pulled towards the Infinity Cache ahead of the launch
        size_t pf_bytes = 0;
    };
    std::vector<Op> ops_;
    std::vector<Op> static_ops_; // depend only on static inputs; run by execute() unless skip_static
    bool to_static_ = false;     // emitters append to static_ops_ while set
    double flops_ = 0;
    hipGraphExec_t graph_exec_ = nullptr;
    hipGraph_t hip_graph_ = nullptr;
    hipStream_t capture_stream_ = nullptr;
    unsigned* fix_counters_ = nullptr;      // tile counters of the in-kernel split-K reduce (gemm.hip), zeroed once
    hipStream_t side_stream_ = nullptr;     // weight prefetch branch (run_ops)
    std::vector<hipEvent_t> pf_events_;     // fork / join events of that branch
    void run_ops(hipStream_t st);
    int eager_runs_ = 0;
    int tune_hits_ = 0, tune_misses_ = 0;

    void build();       // dispatches on kind_
    void build_unet();
    void build_vae();
    void build_clip();
    void build_temb();

    // ---- op emitters (record in REAL mode, account in DRY mode, nothing in DECLARE mode)
    struct GemmOpt {
        int bias = -1;             // param index (PK_VEC) or -1
        const float* bias_raw = nullptr; // raw fp32 bias pointer (fused parameter groups)
        f16* out = nullptr;        // conv(): write here instead of allocating from the arena
        const f16* row_bias = nullptr;
        int ld_row_bias = 0, rows_per_img = 0;
        const f16* residual = nullptr;
        int act = 0;
        float alpha = 1.0f;
        bool bias_on_m = false;
        int lda = 0, ldo = 0;      // 0 = dense
        int ldr = 0;               // row stride of `residual` (0 = the output's)
        bool geglu = false;        // fused value*gelu(gate) epilogue (weights packed PK_LINEAR_GEGLU)
        const Act* tail0 = nullptr; // conv(): 1x1-gathered tail segment sources (ResBlock skip connection)
        const Act* tail1 = nullptr;
        int bias2 = -1;
        int ln_w = -1, ln_b = -1;  // fold LayerNorm(ln_w, ln_b) of the input rows into this Linear (see sdod_ln_fold_f16)
        const float* wq_scale = nullptr; // uint8 weights (linear_raw on a fused group): per-row scale / offset + 128 vectors
        const float* wq_off = nullptr;
        const float* ln_s_raw = nullptr; // LayerNorm fold with vectors the caller owns: ln_s (bias_raw carries the matching t)
        int w_img_stride = 0, vec_img_stride = 0; // per-image weights / vectors (sdod_gemm_desc), with rows_per_img
        int softmax_cols = 0;
    };
    // registers the LayerNorm fold of Linear `w_param` ([N][K], row stride ldw): gamma is multiplied into W in place at finalize,
    // the returned device vectors (owned by the graph) are s[n] = sum_k W'[n][k] and t[n] = sum_k beta[k] W[n][k] + bias[n]
    struct LnVecs { float* s; float* t; };
    LnVecs ln_fold_vectors(f16* w, int N, int K, int ldw, int ln_w, int ln_b, const float* bias);
    struct FoldJob {
        f16* w; int n, k, ldw;
        const float *gamma, *beta, *bias_in;
        float *s, *t;
    };
    std::vector<FoldJob> fold_jobs_;
    // Linear o Linear composition at finalize (sdod_compose_linear_f16): the [n_out][k] block at `c` (row stride ld) holds W on
    // entry and P . W afterwards; P is the [n_out][n_mid] block at `p` of the same matrix
    struct ComposeJob {
        f16* c; const f16* p; int ld, n_out, n_mid, k;
        const float *bias_w, *bias_p;
        float* bias_out;
    };
    std::vector<ComposeJob> compose_jobs_;
    std::vector<void*> derived_; // device buffers created at build time (folded LayerNorm vectors)
    void emit_gemm(sdod_gemm_desc d);
    // out[rows][N] = x[rows][K] . W^T ; W is params_[w] (or a raw fp16 [N][K] pointer through *_raw)
    void linear(const f16* x, int rows, int K, int w, int N, f16* out, const GemmOpt& o);
    void linear_raw(const f16* x, int rows, int K, const f16* w, int ldw, int N, f16* out, const GemmOpt& o);
    // NHWC conv through the gather GEMM (ksize 1 or 3); x2 = optional concat source
    Act conv(const Act& x, const Act* x2, int w, int cout, int ksize, int stride, bool upsample, const GemmOpt& o);
    Act group_norm(const Act& x, const Act* x2, int gw, int gb, float eps, bool silu);
    Act layer_norm(const Act& x, int lw, int lb, float eps);
    void attention(const f16* q, const f16* k, const f16* v, f16* out, int B, int heads, int lq, int lk, int d, int ldq,
                   int ldk, int ldv, int ldo, bool causal);

    // ---- model blocks
    Act res_block(const std::string& pfx, const Act& x, const Act* x2, int cout, const f16* emb_all, int emb_ld, int& emb_off);
    std::vector<Op>& sink() { return to_static_ ? static_ops_ : ops_; }
    // A split-K GEMM whose reduce + epilogue launch (phase 2) has not been emitted yet: if the tensor's first consumer is a
    // GroupNorm the (image, group) kernel takes over that work (sdod_group_norm_reduce_nhwc: one launch instead of three);
    // every other emitter flushes it first.  At most one is pending: the partial slabs live in the shared workspace ws_.
    struct PendingReduce {
        bool active = false;
        sdod_gemm_desc d{};      // phase = 2
        double bytes = 0;
        std::string detail;
    } pending_;
    void flush_pending();
    // The deferred phase 2 still READS the GEMM's residual operand, so a tensor that is a residual must not return to the
    // arena before its consumer has been emitted: release_after_consumer() parks it until the next emitter has settled
    // (same order in the DECLARE / DRY / REAL passes, so the arena plan stays exact).
    std::vector<const void*> parked_;
    void release_after_consumer(const Act& a) { parked_.push_back(a.p); }
    void drain_parked() {
        for (const void* p : parked_) release(p);
        parked_.clear();
    }
    void settle() {
        flush_pending();
        drain_parked();
    }
    Act spatial_transformer(const std::string& pfx, const Act& x, const Act& ctx);
    Act vae_res_block(const std::string& pfx, const Act& x, int cout);
    Act vae_attn_block(const std::string& pfx, const Act& x);
    int emb_total_ = 0; // sum of ResBlock output channels (set by the DECLARE pass)
    int kv_total_ = 0;  // sum over SpatialTransformers of 2*C: width of the batched cross-attention K/V matrix
    f16* kv_all_ = nullptr; // [B*context_len][kv_total_], persistent (not arena memory)
    int kv_off_ = 0;
    std::vector<std::pair<std::string, int>> unet_res_blocks() const; // (prefix, cout) in declaration order
    const char* group_base(const std::string& group) const;
    int group_first(const std::string& group) const; // index of the first parameter of a fused group
    void emit(std::function<void(hipStream_t)> fn, const char* label = "elementwise", double flops = 0, double bytes = 0) {
        settle();
        if (mode_ == REAL) sink().push_back(Op{std::move(fn), label, flops, bytes, ""});
    }

public:
    // GEMM shapes of this graph whose tile came from the tune table(s) / had to be timed in this process
    int tune_hits() const { return tune_hits_; }
    int tune_misses() const { return tune_misses_; }
    int num_ops() const { return (int)ops_.size(); }
    const Op& op(int i) const { return ops_.at(i); }
    // eager run with a HIP event pair around every launch; ms[i] = duration of op i (averaged over `iters` runs)
    void profile(hipStream_t st, int iters, float* ms, int n);
};

} // namespace sdod

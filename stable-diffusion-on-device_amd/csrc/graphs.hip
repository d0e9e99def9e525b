// graphs.hip -- launch-list builders for the four graphs of the txt2img path: UNet eps-model, VAE decoder,
// CLIP text encoder, time-embedding MLP.  They stand in for the serialized QNN graphs the reference loads
// (context.cpp:105: "unet.serialized", "text_encoder.serialized", "vae_decoder.serialized", "temb"); the
// block structure follows the op names visible in the reference's own profiler table
// (analyze_results.py:20-93: in_layers / emb_layers / out_layers / skip_connection, norm / proj_in /
// transformer_blocks.0.{norm1-3, attn1, attn2, ff.net.0.proj, ff.net.2} / proj_out) and the public SD v1.x
// architecture.  Parameter names are the CompVis-ldm / HF-CLIP state-dict keys.
//
// Fusions decided here (all are reads/writes removed from HBM, the UNet sits at the roofline ridge):
//   * SiLU into every ResBlock GroupNorm; the skip-connection channel concat into the GroupNorm reads and
//     into the conv gathers (no concat tensor); nearest-2x upsampling into the conv gather;
//   * conv bias + time-embedding broadcast + residual into the conv epilogue;
//   * to_q/to_k/to_v as ONE GEMM (weights contiguous in the arena); all 22 ResBlock time-embedding
//     projections as ONE GEMM per UNet evaluation; attention output / FF output residuals in the GEMM epilogue;
//   * z/0.18215 and post_quant_conv into the latent layout change.
#include "engine.h"

namespace sdod {

static void check_rc2(int rc) {
    if (rc != 0) throw Error(rc, get_last_error());
}

int Graph::group_first(const std::string& group) const {
    auto it = gindex_.find(group);
    if (it == gindex_.end()) throw Error(INTERNAL_ERROR, "unknown parameter group " + group);
    return groups_[it->second].front();
}

const char* Graph::group_base(const std::string& group) const {
    auto it = gindex_.find(group);
    if (it == gindex_.end()) throw Error(INTERNAL_ERROR, "unknown parameter group " + group);
    return params_[groups_[it->second].front()].dev;
}

// ------------------------------------------------------------------------------------------------ UNet
Act Graph::res_block(const std::string& pfx, const Act& x, const Act* x2, int cout, const f16* emb_all, int emb_ld, int& emb_off) {
    const int cin = x.c + (x2 ? x2->c : 0);
    quant_rows(x.rows());
    const int n1w = P(pfx + ".in_layers.0.weight", {cin}, PK_VEC), n1b = P(pfx + ".in_layers.0.bias", {cin}, PK_VEC);
    const int c1w = P(pfx + ".in_layers.2.weight", {cout, cin, 3, 3}, PK_CONV3), c1b = P(pfx + ".in_layers.2.bias", {cout}, PK_VEC);
    const int n2w = P(pfx + ".out_layers.0.weight", {cout}, PK_VEC), n2b = P(pfx + ".out_layers.0.bias", {cout}, PK_VEC);
    // out_layers.3 (3x3) and skip_connection (1x1) share ONE weight matrix [cout][9*cout + cin]: the skip conv is the tail
    // K-segment of the same GEMM (one launch, no intermediate skip tensor)
    int c2w, skw = -1, skb = -1;
    const bool fuse_skip = !quant_decl_; // uint8 weights: the skip conv's tensor has its own encoding -> its own GEMM
    if (cin != cout && !fuse_skip) {
        c2w = P(pfx + ".out_layers.3.weight", {cout, cout, 3, 3}, PK_CONV3);
        skw = P(pfx + ".skip_connection.weight", {cout, cin, 1, 1}, PK_CONV1);
        skb = P(pfx + ".skip_connection.bias", {cout}, PK_VEC);
    } else if (cin != cout) {
        const int ld = 9 * cout + cin;
        c2w = Pc(pfx + ".out_layers.3.weight", {cout, cout, 3, 3}, PK_CONV3, ld, 0, -1);
        skw = Pc(pfx + ".skip_connection.weight", {cout, cin, 1, 1}, PK_CONV1, ld, 9 * cout, c2w);
        skb = P(pfx + ".skip_connection.bias", {cout}, PK_VEC);
    } else {
        c2w = P(pfx + ".out_layers.3.weight", {cout, cout, 3, 3}, PK_CONV3);
    }
    const int c2b = P(pfx + ".out_layers.3.bias", {cout}, PK_VEC);
    (void)skw;
    const int my_off = emb_off;
    emb_off += cout;

    Act g1 = group_norm(x, x2, n1w, n1b, 1e-5f, true);
    GemmOpt o1;
    o1.bias = c1b;
    o1.row_bias = emb_all ? emb_all + my_off : nullptr;
    o1.ld_row_bias = emb_ld;
    o1.rows_per_img = x.h * x.w;
    Act h = conv(g1, nullptr, c1w, cout, 3, 1, false, o1);
    release(g1);
    Act g2 = group_norm(h, nullptr, n2w, n2b, 1e-5f, true);
    release(h);
    GemmOpt o2;
    o2.bias = c2b;
    Act sk;
    if (cin != cout && !fuse_skip) {
        GemmOpt os;
        os.bias = skb;
        sk = conv(x, x2, skw, cout, 1, 1, false, os); // 1x1 over the (possibly concatenated) block input
        o2.residual = sk.p;
    } else if (cin != cout) {
        o2.tail0 = &x;
        o2.tail1 = x2;
        o2.bias2 = skb;
    } else {
        if (x2) throw Error(INTERNAL_ERROR, "identity skip with a concatenated input");
        o2.residual = x.p;
    }
    Act out = conv(g2, nullptr, c2w, cout, 3, 1, false, o2);
    release(g2);
    if (sk.p) release_after_consumer(sk);
    return out;
}

Act Graph::spatial_transformer(const std::string& pfx, const Act& x, const Act& ctx) {
    const int C = x.c, cd = ctx.c;
    const int heads = cfg_.num_heads > 0 ? cfg_.num_heads : C / cfg_.head_dim;
    const int d = C / heads;
    const int rows = x.rows(), L = x.h * x.w, B = x.n;
    const std::string tb = pfx + ".transformer_blocks.0";
    quant_rows(rows);
    const bool quant_block = quant_decl_;
    const int nw = P(pfx + ".norm.weight", {C}, PK_VEC), nb = P(pfx + ".norm.bias", {C}, PK_VEC);
    const bool lin = cfg_.linear_proj != 0; // SD2.x stores these as Linear [C, C]; the arithmetic is the same
    const int piw = lin ? P(pfx + ".proj_in.weight", {C, C}, PK_LINEAR) : P(pfx + ".proj_in.weight", {C, C, 1, 1}, PK_CONV1);
    const int pib = P(pfx + ".proj_in.bias", {C}, PK_VEC);
    const int l1w = P(tb + ".norm1.weight", {C}, PK_VEC), l1b = P(tb + ".norm1.bias", {C}, PK_VEC);
    const std::string gq = tb + ".attn1.qkv";
    P(tb + ".attn1.to_q.weight", {C, C}, PK_LINEAR, gq);
    P(tb + ".attn1.to_k.weight", {C, C}, PK_LINEAR, gq);
    P(tb + ".attn1.to_v.weight", {C, C}, PK_LINEAR, gq);
    const int o1w = P(tb + ".attn1.to_out.0.weight", {C, C}, PK_LINEAR), o1b = P(tb + ".attn1.to_out.0.bias", {C}, PK_VEC);
    const int l2w = P(tb + ".norm2.weight", {C}, PK_VEC), l2b = P(tb + ".norm2.bias", {C}, PK_VEC);
    const int q2w = P(tb + ".attn2.to_q.weight", {C, C}, PK_LINEAR);
    // every layer's to_k|to_v lives in ONE group: the context projections of all 16 transformers are a single GEMM that
    // runs once per prompt (the context is constant over the sampler run), see build_unet()
    quant_global(); // (one encoding policy for the whole group: its GEMM runs on M = 77 rows per prompt)
    P(tb + ".attn2.to_k.weight", {C, cd}, PK_LINEAR, "attn2_kv_all");
    P(tb + ".attn2.to_v.weight", {C, cd}, PK_LINEAR, "attn2_kv_all");
    quant_decl_ = quant_block;
    const int my_kv = kv_off_;
    kv_off_ += 2 * C;
    const int o2w = P(tb + ".attn2.to_out.0.weight", {C, C}, PK_LINEAR), o2b = P(tb + ".attn2.to_out.0.bias", {C}, PK_VEC);
    const int l3w = P(tb + ".norm3.weight", {C}, PK_VEC), l3b = P(tb + ".norm3.bias", {C}, PK_VEC);
    const int f1w = P(tb + ".ff.net.0.proj.weight", {8 * C, C}, PK_LINEAR_GEGLU), f1b = P(tb + ".ff.net.0.proj.bias", {8 * C}, PK_VEC_GEGLU);
    // ff.net.2 and proj_out follow each other with nothing but a residual add in between:
    //   out = P (W2 g + b2 + t2) + bp + x = [P W2 | P] [g ; t2] + (P b2 + bp) + x
    // so the two Linears (analyze_results.py:69-79 lists them as separate ops) are ONE GEMM over K = 5C on the row-wise concat
    // [g | t2]: both weights live in one [C][5C] matrix whose first 4C columns are replaced by P W2 at finalize
    // (sdod_compose_linear_f16, fp32 accumulate, one fp16 rounding), GEGLU writes g and attn2.to_out writes t2 straight into the
    // concatenated buffer.  One launch and one [rows][C] round trip less per transformer block.  Not with uint8 weights (integer
    // codes cannot be composed); SDOD_COMPOSE=0 keeps the two-GEMM form (A/B switch).
    static const bool compose_on = [] { const char* e = std::getenv("SDOD_COMPOSE"); return !(e && e[0] == '0'); }();
    const bool compose = compose_on && !quant_block;
    int f2w, pow_;
    if (compose) {
        f2w = Pc(tb + ".ff.net.2.weight", {C, 4 * C}, PK_LINEAR, 5 * C, 0, -1);
        pow_ = lin ? Pc(pfx + ".proj_out.weight", {C, C}, PK_LINEAR, 5 * C, 4 * C, f2w)
                   : Pc(pfx + ".proj_out.weight", {C, C, 1, 1}, PK_CONV1, 5 * C, 4 * C, f2w);
    } else {
        f2w = P(tb + ".ff.net.2.weight", {C, 4 * C}, PK_LINEAR);
        pow_ = lin ? P(pfx + ".proj_out.weight", {C, C}, PK_LINEAR) : P(pfx + ".proj_out.weight", {C, C, 1, 1}, PK_CONV1);
    }
    const int f2b = P(tb + ".ff.net.2.bias", {C}, PK_VEC);
    const int pob = P(pfx + ".proj_out.bias", {C}, PK_VEC);
    if (mode_ == DECLARE) return act(x.n, x.h, x.w, C);

    Act g = group_norm(x, nullptr, nw, nb, 1e-6f, false);
    Act t0 = act(B, 1, L, C);
    { GemmOpt o; o.bias = pib; linear(g.p, rows, C, piw, C, t0.p, o); }
    release(g);
    // self-attention
    // the three LayerNorms of the block are folded into the Linear that consumes them (no LN launch, no LN tensor) -- except
    // with uint8 weights, whose integer codes cannot absorb gamma: there LayerNorm is a launch and the Linear a plain one
    const bool fold = !quant_block;
    auto normed = [&](const Act& src, int lw, int lb) -> Act { return layer_norm(src, lw, lb, 1e-5f); };
    f16* qkv = alloc((size_t)rows * 3 * C);
    if (fold) {
        GemmOpt o; o.ln_w = l1w; o.ln_b = l1b;
        linear_raw(t0.p, rows, C, reinterpret_cast<const f16*>(group_base(gq)), C, 3 * C, qkv, o);
    } else {
        Act nrm = normed(t0, l1w, l1b);
        GemmOpt o; o.wq_scale = qscale_of(group_first(gq)); o.wq_off = qoff_of(group_first(gq));
        linear_raw(nrm.p, rows, C, reinterpret_cast<const f16*>(group_base(gq)), C, 3 * C, qkv, o);
        release(nrm);
    }
    f16* a1 = alloc((size_t)rows * C);
    attention(qkv, qkv + C, qkv + 2 * C, a1, B, heads, L, L, d, 3 * C, 3 * C, 3 * C, C, false);
    release(qkv);
    Act t1 = act(B, 1, L, C);
    { GemmOpt o; o.bias = o1b; o.residual = t0.p; linear(a1, rows, C, o1w, C, t1.p, o); }
    release(a1); release(t0);
    // cross-attention on the text context.  FOLDED form (attention.hip: sdod_xattn_fold_f16): the context, and with it K and V,
    // is constant over the sampler run, so to_q is multiplied into K and to_out into V once per prompt (static launch list) and
    // the block is two GEMMs per evaluation -- scores = LN(t1) . (Wq^T K_h^T) with the row softmax over each head's 77 (+3
    // padding) columns in the epilogue, then [P_1 | .. | P_H] . (V_h Wo_h^T) + bias + residual -- instead of Linear + attention
    // kernel + Linear: one launch less per block, and at the deep levels half the weight bytes (2 x 640 x C against 2 x C x C).
    // Needs the LayerNorm fold (fp16 weights), heads * 80 columns tiled by the 160-wide GEMM tiles and the 64-deep K slabs,
    // image rows a multiple of 32.  SDOD_XATTN_FOLD=0 keeps the three-launch form (A/B switch).
    static const bool xfold_on = [] { const char* e = std::getenv("SDOD_XATTN_FOLD"); return !(e && e[0] == '0'); }();
    const int NK = heads * 80;
    const bool xfold = xfold_on && fold && ctx.w <= 80 && NK % 320 == 0 && L % 32 == 0 && d <= 160 && d % 4 == 0;
    if (xfold) {
        f16 *w1 = nullptr, *w2 = nullptr;
        float *s1 = nullptr, *t1v = nullptr;
        f16* wq = reinterpret_cast<f16*>(params_[q2w].dev);
        const f16* wo = reinterpret_cast<const f16*>(params_[o2w].dev);
        if (mode_ == REAL) {
            SDOD_HIP_CHECK(hipMalloc((void**)&w1, (size_t)B * NK * C * sizeof(f16)));
            SDOD_HIP_CHECK(hipMalloc((void**)&w2, (size_t)B * NK * C * sizeof(f16)));
            SDOD_HIP_CHECK(hipMalloc((void**)&s1, (size_t)B * NK * sizeof(float)));
            SDOD_HIP_CHECK(hipMalloc((void**)&t1v, (size_t)B * NK * sizeof(float)));
            derived_.push_back(w1); derived_.push_back(w2); derived_.push_back(s1); derived_.push_back(t1v);
            const LnVecs qv = ln_fold_vectors(wq, C, C, C, l2w, l2b, nullptr);
            const f16* kvp = kv_all_;
            const int ldkv = kv_total_, koff = my_kv, Lk = ctx.w, hh = heads, dd = d;
            const float scale = 1.0f / sqrtf((float)d);
            to_static_ = true;
            sink().push_back(Op{[=](hipStream_t st) {
                check_rc2(sdod_xattn_fold_f16(kvp, ldkv, koff, koff + C, B, Lk, wq, C, qv.s, qv.t, wo, C, hh, dd, scale, w1, s1, t1v, w2, st));
            }, "xattn_fold", 4.0 * B * NK * (double)C * d, 4.0 * B * NK * C * 2, "C" + std::to_string(C) + " d" + std::to_string(d)});
            to_static_ = false;
        }
        const f16* w1p = mode_ == REAL ? w1 : wq; // (placeholders during the sizing pass)
        const f16* w2p = mode_ == REAL ? w2 : wq;
        f16* pb = alloc((size_t)rows * NK);
        { GemmOpt o; o.ln_s_raw = mode_ == REAL ? s1 : reinterpret_cast<const float*>(wq); o.bias_raw = mode_ == REAL ? t1v : reinterpret_cast<const float*>(wq);
          o.w_img_stride = NK * C; o.vec_img_stride = NK; o.softmax_cols = 80; o.rows_per_img = L;
          linear_raw(t1.p, rows, C, w1p, C, NK, pb, o); }
        GemmOpt o2;
        o2.bias = o2b; o2.residual = t1.p; o2.w_img_stride = C * NK; o2.rows_per_img = L;
        if (compose) {
            f16* cat = alloc((size_t)rows * 5 * C);
            f16* t2c = cat + 4 * C;
            o2.ldr = C; o2.ldo = 5 * C;
            linear_raw(pb, rows, NK, w2p, NK, C, t2c, o2);
            release(pb); release(t1);
            { GemmOpt o; o.bias = f1b; o.geglu = true; o.ln_w = l3w; o.ln_b = l3b; o.lda = 5 * C; o.ldo = 5 * C; linear(t2c, rows, C, f1w, 8 * C, cat, o); }
            float* bc = nullptr;
            if (mode_ == REAL) {
                SDOD_HIP_CHECK(hipMalloc((void**)&bc, (size_t)C * sizeof(float)));
                derived_.push_back(bc);
                f16* wc = reinterpret_cast<f16*>(params_[f2w].dev);
                compose_jobs_.push_back(ComposeJob{wc, wc + 4 * C, 5 * C, C, C, 4 * C, W<float>(f2b), W<float>(pob), bc});
            }
            Act out = act(x.n, x.h, x.w, C);
            { GemmOpt o; o.bias_raw = mode_ == REAL ? bc : reinterpret_cast<const float*>(params_[f2w].dev); o.residual = x.p;
              linear_raw(cat, rows, 5 * C, reinterpret_cast<const f16*>(params_[f2w].dev), 5 * C, C, out.p, o); }
            release(cat);
            return out;
        }
        Act t2 = act(B, 1, L, C);
        linear_raw(pb, rows, NK, w2p, NK, C, t2.p, o2);
        release(pb); release(t1);
        f16* gg = alloc((size_t)rows * 4 * C);
        { GemmOpt o; o.bias = f1b; o.geglu = true; o.ln_w = l3w; o.ln_b = l3b; linear(t2.p, rows, C, f1w, 8 * C, gg, o); }
        Act t3 = act(B, 1, L, C);
        { GemmOpt o; o.bias = f2b; o.residual = t2.p; linear(gg, rows, 4 * C, f2w, C, t3.p, o); }
        release(gg); release(t2);
        Act out = act(x.n, x.h, x.w, C);
        { GemmOpt o; o.bias = pob; o.residual = x.p; linear(t3.p, rows, C, pow_, C, out.p, o); }
        release(t3);
        return out;
    }
    f16* q2 = alloc((size_t)rows * C);
    if (fold) {
        GemmOpt o; o.ln_w = l2w; o.ln_b = l2b; linear(t1.p, rows, C, q2w, C, q2, o);
    } else {
        Act nrm = normed(t1, l2w, l2b);
        linear(nrm.p, rows, C, q2w, C, q2, GemmOpt{});
        release(nrm);
    }
    const int Lk = ctx.w;
    const f16* kv = kv_all_ + my_kv;
    f16* a2 = alloc((size_t)rows * C);
    attention(q2, kv, kv + C, a2, B, heads, L, Lk, d, C, kv_total_, kv_total_, C, false);
    release(q2);
    if (compose) {
        // [g | t2] rows of 5C: attn2.to_out writes its C columns (row stride 5C), the GEGLU projection reads them (LayerNorm
        // folded) and writes the first 4C, the composed Linear reads all 5C
        f16* cat = alloc((size_t)rows * 5 * C);
        f16* t2c = cat + 4 * C;
        { GemmOpt o; o.bias = o2b; o.residual = t1.p; o.ldr = C; o.ldo = 5 * C; linear(a2, rows, C, o2w, C, t2c, o); }
        release(a2); release(t1);
        { GemmOpt o; o.bias = f1b; o.geglu = true; o.ln_w = l3w; o.ln_b = l3b; o.lda = 5 * C; o.ldo = 5 * C; linear(t2c, rows, C, f1w, 8 * C, cat, o); }
        float* bc = nullptr;
        if (mode_ == REAL) {
            SDOD_HIP_CHECK(hipMalloc((void**)&bc, (size_t)C * sizeof(float)));
            derived_.push_back(bc);
            f16* wc = reinterpret_cast<f16*>(params_[f2w].dev);
            compose_jobs_.push_back(ComposeJob{wc, wc + 4 * C, 5 * C, C, C, 4 * C, W<float>(f2b), W<float>(pob), bc});
        }
        Act out = act(x.n, x.h, x.w, C);
        { GemmOpt o; o.bias_raw = mode_ == REAL ? bc : reinterpret_cast<const float*>(params_[f2w].dev); o.residual = x.p;
          linear_raw(cat, rows, 5 * C, reinterpret_cast<const f16*>(params_[f2w].dev), 5 * C, C, out.p, o); }
        release(cat);
        return out;
    }
    Act t2 = act(B, 1, L, C);
    { GemmOpt o; o.bias = o2b; o.residual = t1.p; linear(a2, rows, C, o2w, C, t2.p, o); }
    release(a2); release(t1);
    // GEGLU feed-forward
    f16* gg = alloc((size_t)rows * 4 * C); // GEGLU fused into the ff.net.0.proj epilogue: the [rows][8C] tensor never exists
    if (fold) {
        GemmOpt o; o.bias = f1b; o.geglu = true; o.ln_w = l3w; o.ln_b = l3b; linear(t2.p, rows, C, f1w, 8 * C, gg, o);
    } else {
        Act nrm = normed(t2, l3w, l3b);
        GemmOpt o; o.bias = f1b; o.geglu = true; linear(nrm.p, rows, C, f1w, 8 * C, gg, o);
        release(nrm);
    }
    Act t3 = act(B, 1, L, C);
    { GemmOpt o; o.bias = f2b; o.residual = t2.p; linear(gg, rows, 4 * C, f2w, C, t3.p, o); }
    release(gg); release(t2);
    Act out = act(x.n, x.h, x.w, C);
    { GemmOpt o; o.bias = pob; o.residual = x.p; linear(t3.p, rows, C, pow_, C, out.p, o); }
    release(t3);
    return out;
}

void Graph::build_unet() {
    const int B = batch_, H = cfg_.latent_h, Wd = cfg_.latent_w, LC = cfg_.latent_channels;
    const int MC = cfg_.model_channels, EC = 4 * MC, cd = cfg_.context_dim, CL = cfg_.context_len;
    SDOD_REQUIRE(LC * 9 <= 64, "latent_channels too large for the small-Cin path");
    const int mult[4] = {1, 2, 4, 4};

    if (mode_ != DECLARE) { // widths fixed by the DECLARE pass
        int tot = 0;
        for (auto& rb : unet_res_blocks()) tot += rb.second;
        emb_total_ = tot;
    }
    float* x_in = (float*)io_alloc(inputs_, (size_t)B * LC * H * Wd * sizeof(float));
    f16* temb_in = (f16*)io_alloc(inputs_, (size_t)B * std::max(emb_total_, 1) * sizeof(f16));
    f16* ctx_in = (f16*)io_alloc(inputs_, (size_t)B * CL * cd * sizeof(f16));
    f16* e_out = (f16*)io_alloc(outputs_, (size_t)B * H * Wd * LC * sizeof(f16));
    Act ctx;
    ctx.p = ctx_in; ctx.n = B; ctx.h = 1; ctx.w = CL; ctx.c = cd;
    (void)EC;
    // the time conditioning arrives already projected for every ResBlock (TEMB graph): row b, columns [off, off+cout)
    const int emb_ld = emb_total_;
    const f16* emb_all = temb_in;
    int emb_off = 0;
    kv_off_ = 0;
    // cross-attention K/V of ALL transformers: one GEMM on the text context, re-run only when the context changes
    if (mode_ != DECLARE && kv_total_ > 0) {
        to_static_ = true;
        GemmOpt okv;
        if (quant_mode()) { okv.wq_scale = qscale_of(group_first("attn2_kv_all")); okv.wq_off = qoff_of(group_first("attn2_kv_all")); }
        linear_raw(ctx_in, B * CL, cd, reinterpret_cast<const f16*>(group_base("attn2_kv_all")), cd, kv_total_,
                   mode_ == REAL ? kv_all_ : ctx_in /* placeholder during the sizing pass */, okv);
        to_static_ = false;
    }

    // input conv (Cin = 4): im2col to K = 64, then the GEMM
    const int ciw = P("input_blocks.0.0.weight", {MC, LC, 3, 3}, PK_CONV3_SMALL), cib = P("input_blocks.0.0.bias", {MC}, PK_VEC);
    Act h = act(B, H, Wd, MC);
    if (MC == 320 || MC == 256 || MC == 128 || MC == 64) {
        // ONE launch: im2col rows built in LDS, the whole [MC][64] weight matrix next to them (elementwise.hip: conv_in_kernel)
        const f16* wp = mode_ != DECLARE ? W<f16>(ciw) : nullptr;
        const float* bp = mode_ != DECLARE ? W<float>(cib) : nullptr;
        f16* hp = h.p;
        settle();
        if (mode_ == REAL) {
            flops_ += 2.0 * B * H * Wd * MC * 64;
            ops_.push_back(Op{[=](hipStream_t st) { check_rc2(sdod_conv_in_f16(x_in, wp, bp, hp, B, H, Wd, LC, MC, 1.0f, st)); }, "conv_in",
                              2.0 * B * H * Wd * MC * 64, (double)B * H * Wd * (LC * 4 + MC * 2) + MC * 64 * 2,
                              "M" + std::to_string(B * H * Wd) + " N" + std::to_string(MC) + " K64"});
        }
    } else {
        f16* cols = alloc((size_t)B * H * Wd * 64);
        emit([=](hipStream_t st) { check_rc2(sdod_latent_im2col_f16(x_in, cols, B, H, Wd, LC, 64, 1.0f, st)); });
        { GemmOpt o; o.bias = cib; linear(cols, B * H * Wd, 64, ciw, MC, h.p, o); }
        release(cols);
    }

    std::vector<Act> hs;
    hs.push_back(h);
    int ch = MC, ds = 1, idx = 1;
    for (int level = 0; level < 4; ++level) {
        for (int i = 0; i < 2; ++i) {
            const std::string pfx = "input_blocks." + std::to_string(idx++);
            Act r = res_block(pfx + ".0", h, nullptr, mult[level] * MC, emb_all, emb_ld, emb_off);
            ch = mult[level] * MC;
            if (ds <= 4) {
                Act t = spatial_transformer(pfx + ".1", r, ctx);
                release(r);
                r = t;
            }
            h = r;
            hs.push_back(h);
        }
        if (level != 3) {
            const std::string pfx = "input_blocks." + std::to_string(idx++) + ".0.op";
            quant_rows(h.rows() / 4);
            const int w = P(pfx + ".weight", {ch, ch, 3, 3}, PK_CONV3), b = P(pfx + ".bias", {ch}, PK_VEC);
            GemmOpt o; o.bias = b;
            h = conv(h, nullptr, w, ch, 3, 2, false, o);
            hs.push_back(h);
            ds *= 2;
        }
    }
    // middle (h is the last skip and stays on the stack until popped)
    {
        Act r1 = res_block("middle_block.0", h, nullptr, ch, emb_all, emb_ld, emb_off);
        Act t = spatial_transformer("middle_block.1", r1, ctx);
        release(r1);
        Act r2 = res_block("middle_block.2", t, nullptr, ch, emb_all, emb_ld, emb_off);
        release_after_consumer(t); // residual of r2's (possibly still split-K) out conv
        h = r2;
    }
    int oidx = 0;
    for (int level = 3; level >= 0; --level) {
        for (int i = 0; i < 3; ++i) {
            const std::string pfx = "output_blocks." + std::to_string(oidx++);
            Act skip = hs.back();
            hs.pop_back();
            Act r = res_block(pfx + ".0", h, &skip, mult[level] * MC, emb_all, emb_ld, emb_off);
            release(h); release(skip);
            ch = mult[level] * MC;
            int sub = 1;
            if (ds <= 4) {
                Act t = spatial_transformer(pfx + "." + std::to_string(sub++), r, ctx);
                release(r);
                r = t;
            }
            if (level != 0 && i == 2) {
                const std::string up = pfx + "." + std::to_string(sub) + ".conv";
                quant_rows(r.rows() * 4);
                const int w = P(up + ".weight", {ch, ch, 3, 3}, PK_CONV3), b = P(up + ".bias", {ch}, PK_VEC);
                GemmOpt o; o.bias = b;
                Act u = conv(r, nullptr, w, ch, 3, 1, true, o);
                release(r);
                r = u;
                ds /= 2;
            }
            h = r;
        }
    }
    const int ow = P("out.0.weight", {ch}, PK_VEC), ob = P("out.0.bias", {ch}, PK_VEC);
    quant_rows(h.rows());
    const int cw = P("out.2.weight", {LC, ch, 3, 3}, PK_CONV3), cb = P("out.2.bias", {LC}, PK_VEC);
    Act g = group_norm(h, nullptr, ow, ob, 1e-5f, true);
    release(h);
    { GemmOpt o; o.bias = cb; o.out = e_out; conv(g, nullptr, cw, LC, 3, 1, false, o); }
    release(g);
    if (mode_ == DECLARE) kv_total_ = kv_off_;
    else if (emb_off != emb_total_ || kv_off_ != kv_total_) throw Error(INTERNAL_ERROR, "UNet conditioning bookkeeping mismatch");
}

// (prefix, cout) of every ResBlock in the order build_unet() visits them: the column layout of the TEMB graph's output
std::vector<std::pair<std::string, int>> Graph::unet_res_blocks() const {
    std::vector<std::pair<std::string, int>> v;
    const int MC = cfg_.model_channels;
    const int mult[4] = {1, 2, 4, 4};
    int idx = 1;
    for (int level = 0; level < 4; ++level) {
        for (int i = 0; i < 2; ++i) v.emplace_back("input_blocks." + std::to_string(idx++) + ".0", mult[level] * MC);
        if (level != 3) ++idx;
    }
    v.emplace_back("middle_block.0", 4 * MC);
    v.emplace_back("middle_block.2", 4 * MC);
    int oidx = 0;
    for (int level = 3; level >= 0; --level)
        for (int i = 0; i < 3; ++i) v.emplace_back("output_blocks." + std::to_string(oidx++) + ".0", mult[level] * MC);
    return v;
}

// ------------------------------------------------------------------------------------------------ temb
void Graph::build_temb() {
    // context.cpp:257-278 (sinusoid -> time MLP), extended by everything else that depends on t only: SiLU and the
    // emb_layers projection of all 22 ResBlocks, as ONE GEMM.  Output row = what the UNet graph consumes as in1.
    const int B = batch_, MC = cfg_.model_channels, EC = 4 * MC;
    const auto blocks = unet_res_blocks();
    int total = 0;
    for (auto& rb : blocks) total += rb.second;
    float* t_in = (float*)io_alloc(inputs_, (size_t)B * sizeof(float));
    f16* out = (f16*)io_alloc(outputs_, (size_t)B * total * sizeof(f16));
    const int w0 = P("time_embed.0.weight", {EC, MC}, PK_LINEAR), b0 = P("time_embed.0.bias", {EC}, PK_VEC);
    const int w2 = P("time_embed.2.weight", {EC, EC}, PK_LINEAR), b2 = P("time_embed.2.bias", {EC}, PK_VEC);
    for (auto& rb : blocks) {
        P(rb.first + ".emb_layers.1.weight", {rb.second, EC}, PK_LINEAR, "emb_w");
        P(rb.first + ".emb_layers.1.bias", {rb.second}, PK_VEC, "emb_b");
    }
    f16* feat = alloc((size_t)B * MC);
    emit([=](hipStream_t st) { check_rc2(sdod_timestep_features_f16(t_in, feat, B, MC, st)); });
    f16* hmid = alloc((size_t)B * EC);
    { GemmOpt o; o.bias = b0; o.act = SDOD_ACT_SILU; linear(feat, B, MC, w0, EC, hmid, o); }
    f16* emb = alloc((size_t)B * EC);
    { GemmOpt o; o.bias = b2; o.act = SDOD_ACT_SILU; linear(hmid, B, EC, w2, EC, emb, o); } // SiLU(time_embed(t)): emb_layers.0
    if (mode_ != DECLARE) {
        GemmOpt o;
        o.bias_raw = reinterpret_cast<const float*>(group_base("emb_b"));
        if (quant_mode()) { o.wq_scale = qscale_of(group_first("emb_w")); o.wq_off = qoff_of(group_first("emb_w")); }
        linear_raw(emb, B, EC, reinterpret_cast<const f16*>(group_base("emb_w")), EC, total, out, o);
    }
    release(feat); release(hmid); release(emb);
}

// ------------------------------------------------------------------------------------------------ VAE
Act Graph::vae_res_block(const std::string& pfx, const Act& x, int cout) {
    const int cin = x.c;
    const int n1w = P(pfx + ".norm1.weight", {cin}, PK_VEC), n1b = P(pfx + ".norm1.bias", {cin}, PK_VEC);
    const int c1w = P(pfx + ".conv1.weight", {cout, cin, 3, 3}, PK_CONV3), c1b = P(pfx + ".conv1.bias", {cout}, PK_VEC);
    const int n2w = P(pfx + ".norm2.weight", {cout}, PK_VEC), n2b = P(pfx + ".norm2.bias", {cout}, PK_VEC);
    const int c2w = P(pfx + ".conv2.weight", {cout, cout, 3, 3}, PK_CONV3), c2b = P(pfx + ".conv2.bias", {cout}, PK_VEC);
    int skw = -1, skb = -1;
    if (cin != cout) {
        skw = P(pfx + ".nin_shortcut.weight", {cout, cin, 1, 1}, PK_CONV1);
        skb = P(pfx + ".nin_shortcut.bias", {cout}, PK_VEC);
    }
    Act g1 = group_norm(x, nullptr, n1w, n1b, 1e-6f, true);
    GemmOpt o1; o1.bias = c1b;
    Act h = conv(g1, nullptr, c1w, cout, 3, 1, false, o1);
    release(g1);
    Act g2 = group_norm(h, nullptr, n2w, n2b, 1e-6f, true);
    release(h);
    Act s;
    GemmOpt o2; o2.bias = c2b;
    if (cin != cout) {
        s = act(x.n, x.h, x.w, cout);
        GemmOpt os; os.bias = skb;
        linear(x.p, x.rows(), cin, skw, cout, s.p, os);
        o2.residual = s.p;
    } else {
        o2.residual = x.p;
    }
    Act out = conv(g2, nullptr, c2w, cout, 3, 1, false, o2);
    release(g2);
    if (s.p) release_after_consumer(s);
    return out;
}

Act Graph::vae_attn_block(const std::string& pfx, const Act& x) {
    const int C = x.c, L = x.h * x.w, B = x.n, rows = x.rows();
    const int nw = P(pfx + ".norm.weight", {C}, PK_VEC), nb = P(pfx + ".norm.bias", {C}, PK_VEC);
    const std::string gw = pfx + ".qk_w", gb = pfx + ".qk_b";
    P(pfx + ".q.weight", {C, C, 1, 1}, PK_CONV1, gw);
    P(pfx + ".k.weight", {C, C, 1, 1}, PK_CONV1, gw);
    P(pfx + ".q.bias", {C}, PK_VEC, gb);
    P(pfx + ".k.bias", {C}, PK_VEC, gb);
    const int vw = P(pfx + ".v.weight", {C, C, 1, 1}, PK_CONV1), vb = P(pfx + ".v.bias", {C}, PK_VEC);
    const int pw = P(pfx + ".proj_out.weight", {C, C, 1, 1}, PK_CONV1), pb = P(pfx + ".proj_out.bias", {C}, PK_VEC);
    if (mode_ == DECLARE) return act(x.n, x.h, x.w, C);
    SDOD_REQUIRE(L % 64 == 0 && L <= 16384, "VAE attention needs H*W % 64 == 0 and <= 16384");

    Act g = group_norm(x, nullptr, nw, nb, 1e-6f, false);
    f16* qk = alloc((size_t)rows * 2 * C);
    { GemmOpt o; o.bias_raw = reinterpret_cast<const float*>(group_base(gb));
      linear_raw(g.p, rows, C, reinterpret_cast<const f16*>(group_base(gw)), C, 2 * C, qk, o); }
    f16* att = alloc((size_t)rows * C);
    for (int b = 0; b < B; ++b) {
        const f16* gb_ = g.p + (size_t)b * L * C;
        f16* vt = alloc((size_t)C * L); // V^T [C][L] = Wv . g_b^T + bv   (bias indexed by the output row)
        { GemmOpt o; o.bias = vb; o.bias_on_m = true;
          linear_raw(W<f16>(vw), C, C, gb_, C, L, vt, o); }
        f16* sc = alloc((size_t)L * L);  // scores, softmaxed in place
        { GemmOpt o; o.alpha = 1.0f / sqrtf((float)C); o.lda = 2 * C;
          linear_raw(qk + (size_t)b * L * 2 * C, L, C, qk + (size_t)b * L * 2 * C + C, 2 * C, L, sc, o); }
        emit([=](hipStream_t st) { check_rc2(sdod_softmax_rows_f16(sc, sc, L, L, st)); }, "softmax_rows", 0, 2.0 * L * L * 2);
        linear_raw(sc, L, L, vt, L, C, att + (size_t)b * L * C, GemmOpt{});
        release(sc); release(vt);
    }
    release(qk); release(g);
    Act out = act(x.n, x.h, x.w, C);
    { GemmOpt o; o.bias = pb; o.residual = x.p; linear(att, rows, C, pw, C, out.p, o); }
    release(att);
    return out;
}

void Graph::build_vae() {
    const int B = batch_, H = cfg_.latent_h, Wd = cfg_.latent_w, LC = cfg_.latent_channels, VC = cfg_.vae_channels;
    SDOD_REQUIRE(LC * 9 <= 64, "latent_channels too large for the small-Cin path");
    const int mult[4] = {1, 2, 4, 4};
    float* z_in = (float*)io_alloc(inputs_, (size_t)B * LC * H * Wd * sizeof(float));
    f16* img_out = (f16*)io_alloc(outputs_, (size_t)B * (8 * H) * (8 * Wd) * 3 * sizeof(f16));

    const int pqw = P("post_quant_conv.weight", {LC, LC, 1, 1}, PK_MAT_F32), pqb = P("post_quant_conv.bias", {LC}, PK_VEC);
    Act zl = act(B, H, Wd, LC);
    {
        const float* wq = W<float>(pqw); const float* bq = W<float>(pqb);
        emit([=](hipStream_t st) { check_rc2(sdod_latent_prep_f16(z_in, wq, bq, zl.p, B, LC, H * Wd, 1.0f / 0.18215f, st)); });
    }
    int ch = VC * mult[3];
    const int ciw = P("decoder.conv_in.weight", {ch, LC, 3, 3}, PK_CONV3_SMALL), cib = P("decoder.conv_in.bias", {ch}, PK_VEC);
    f16* cols = alloc((size_t)B * H * Wd * 64);
    emit([=](hipStream_t st) { check_rc2(sdod_im2col3x3_small_f16(zl.p, cols, B, H, Wd, LC, 64, st)); });
    Act h = act(B, H, Wd, ch);
    { GemmOpt o; o.bias = cib; linear(cols, B * H * Wd, 64, ciw, ch, h.p, o); }
    release(cols); release(zl);

    { Act r = vae_res_block("decoder.mid.block_1", h, ch); release_after_consumer(h); h = r; }
    { Act r = vae_attn_block("decoder.mid.attn_1", h); release(h); h = r; }
    { Act r = vae_res_block("decoder.mid.block_2", h, ch); release_after_consumer(h); h = r; }
    for (int level = 3; level >= 0; --level) {
        const int cout = VC * mult[level];
        for (int i = 0; i < 3; ++i) {
            Act r = vae_res_block("decoder.up." + std::to_string(level) + ".block." + std::to_string(i), h, cout);
            release_after_consumer(h);
            h = r;
        }
        ch = cout;
        if (level != 0) {
            const std::string up = "decoder.up." + std::to_string(level) + ".upsample.conv";
            const int w = P(up + ".weight", {ch, ch, 3, 3}, PK_CONV3), b = P(up + ".bias", {ch}, PK_VEC);
            GemmOpt o; o.bias = b;
            Act u = conv(h, nullptr, w, ch, 3, 1, true, o);
            release(h);
            h = u;
        }
    }
    const int nw = P("decoder.norm_out.weight", {ch}, PK_VEC), nb = P("decoder.norm_out.bias", {ch}, PK_VEC);
    const int cw = P("decoder.conv_out.weight", {3, ch, 3, 3}, PK_CONV3), cb = P("decoder.conv_out.bias", {3}, PK_VEC);
    Act g = group_norm(h, nullptr, nw, nb, 1e-6f, true);
    release(h);
    { GemmOpt o; o.bias = cb; o.out = img_out; conv(g, nullptr, cw, 3, 3, 1, false, o); }
    release(g);
}

// ------------------------------------------------------------------------------------------------ CLIP
void Graph::build_clip() {
    const int B = batch_, L = cfg_.context_len, D = cfg_.context_dim, heads = cfg_.text_heads, NL = cfg_.text_layers;
    const int rows = B * L, inter = 4 * D;
    SDOD_REQUIRE(D % heads == 0 && D / heads == 64, "text encoder head dim must be 64");
    int32_t* ids = (int32_t*)io_alloc(inputs_, (size_t)rows * sizeof(int32_t));
    f16* out = (f16*)io_alloc(outputs_, (size_t)rows * D * sizeof(f16));
    // two checkpoint dialects of the same pre-LN causal transformer (sdod_model_config.text_arch): HF CLIPTextModel names with
    // separate q/k/v projections and quick-GELU (SD1.x), or open_clip names with a fused in_proj and erf GELU (SD2.x)
    const bool oc = cfg_.text_arch == 1;
    const int te = P(oc ? "token_embedding.weight" : "text_model.embeddings.token_embedding.weight", {cfg_.vocab_size, D}, PK_EMBED);
    const int pe = P(oc ? "positional_embedding" : "text_model.embeddings.position_embedding.weight", {L, D}, PK_EMBED);
    Act x = act(B, 1, L, D);
    {
        const f16* tp = W<f16>(te); const f16* pp = W<f16>(pe);
        emit([=](hipStream_t st) { check_rc2(sdod_embedding_f16(ids, tp, pp, x.p, rows, L, D, st)); });
    }
    for (int l = 0; l < NL; ++l) {
        const std::string pfx = (oc ? "transformer.resblocks." : "text_model.encoder.layers.") + std::to_string(l);
        const int l1w = P(pfx + (oc ? ".ln_1.weight" : ".layer_norm1.weight"), {D}, PK_VEC), l1b = P(pfx + (oc ? ".ln_1.bias" : ".layer_norm1.bias"), {D}, PK_VEC);
        const std::string gw = pfx + ".qkv_w", gb = pfx + ".qkv_b";
        int qkvw = -1, qkvb = -1;
        if (oc) {
            qkvw = P(pfx + ".attn.in_proj_weight", {3 * D, D}, PK_LINEAR);
            qkvb = P(pfx + ".attn.in_proj_bias", {3 * D}, PK_VEC);
        } else {
            for (const char* n : {"q_proj", "k_proj", "v_proj"}) P(pfx + ".self_attn." + n + ".weight", {D, D}, PK_LINEAR, gw);
            for (const char* n : {"q_proj", "k_proj", "v_proj"}) P(pfx + ".self_attn." + n + ".bias", {D}, PK_VEC, gb);
        }
        const int ow = P(pfx + (oc ? ".attn.out_proj.weight" : ".self_attn.out_proj.weight"), {D, D}, PK_LINEAR);
        const int ob = P(pfx + (oc ? ".attn.out_proj.bias" : ".self_attn.out_proj.bias"), {D}, PK_VEC);
        const int l2w = P(pfx + (oc ? ".ln_2.weight" : ".layer_norm2.weight"), {D}, PK_VEC), l2b = P(pfx + (oc ? ".ln_2.bias" : ".layer_norm2.bias"), {D}, PK_VEC);
        const int f1w = P(pfx + (oc ? ".mlp.c_fc.weight" : ".mlp.fc1.weight"), {inter, D}, PK_LINEAR), f1b = P(pfx + (oc ? ".mlp.c_fc.bias" : ".mlp.fc1.bias"), {inter}, PK_VEC);
        const int f2w = P(pfx + (oc ? ".mlp.c_proj.weight" : ".mlp.fc2.weight"), {D, inter}, PK_LINEAR), f2b = P(pfx + (oc ? ".mlp.c_proj.bias" : ".mlp.fc2.bias"), {D}, PK_VEC);
        if (mode_ == DECLARE) continue;
        // both LayerNorms of a block are folded into the Linear that consumes them (as in the UNet's transformer blocks): no
        // LayerNorm launch, no normalised tensor -- 2 x text_layers launches fewer per prompt
        f16* qkv = alloc((size_t)rows * 3 * D);
        if (oc) {
            GemmOpt o; o.bias = qkvb; o.ln_w = l1w; o.ln_b = l1b;
            linear(x.p, rows, D, qkvw, 3 * D, qkv, o);
        } else {
            GemmOpt o; o.bias_raw = reinterpret_cast<const float*>(group_base(gb)); o.ln_w = l1w; o.ln_b = l1b;
            linear_raw(x.p, rows, D, reinterpret_cast<const f16*>(group_base(gw)), D, 3 * D, qkv, o);
        }
        f16* a = alloc((size_t)rows * D);
        attention(qkv, qkv + D, qkv + 2 * D, a, B, heads, L, L, D / heads, 3 * D, 3 * D, 3 * D, D, true);
        release(qkv);
        Act x1 = act(B, 1, L, D);
        { GemmOpt o; o.bias = ob; o.residual = x.p; linear(a, rows, D, ow, D, x1.p, o); }
        release(a); release(x);
        f16* f = alloc((size_t)rows * inter);
        { GemmOpt o; o.bias = f1b; o.act = oc ? SDOD_ACT_GELU : SDOD_ACT_QUICK_GELU; o.ln_w = l2w; o.ln_b = l2b; linear(x1.p, rows, D, f1w, inter, f, o); }
        Act x2 = act(B, 1, L, D);
        { GemmOpt o; o.bias = f2b; o.residual = x1.p; linear(f, rows, inter, f2w, D, x2.p, o); }
        release(f); release(x1);
        x = x2;
    }
    const int fw = P(oc ? "ln_final.weight" : "text_model.final_layer_norm.weight", {D}, PK_VEC);
    const int fb = P(oc ? "ln_final.bias" : "text_model.final_layer_norm.bias", {D}, PK_VEC);
    settle();
    if (mode_ == REAL) {
        const f16* xp = x.p; const float* wp = W<float>(fw); const float* bp = W<float>(fb);
        ops_.push_back(Op{[=](hipStream_t st) { check_rc2(sdod_layer_norm_f16(xp, out, wp, bp, rows, D, 1e-5f, st)); }, "layer_norm", 0,
                          2.0 * rows * D * 2, ""});
    }
    release(x);
}

} // namespace sdod

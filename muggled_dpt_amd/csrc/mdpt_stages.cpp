// libmdpt: launch sequences of the five stages - patch embed, encoder (4 taps), reassemble, RefineNet fusion, depth head - for all four
// model families. Stage structure mirrors DPTModel.forward (reference muggled_dpt/dpt_model.py:61-83). Every launch goes on the caller's
// stream; nothing here allocates or synchronises.
#include "mdpt_internal.h"

namespace mdpt {

GemmParams base_params(const Ctx& c, const Mat& w, Planes a, int M, int lda) {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.npass = c.h->terms(w.cls);  // the class of the weight matrix decides; an A buffer shared with a 3-pass class may carry an unused lo plane
    g.A_hi = a.hi; g.A_lo = g.npass >= 2 ? a.lo : nullptr;
    g.W_hi = w.hi; g.W_lo = g.npass == 3 ? w.lo : nullptr;
    if (g.npass >= 2 && c.h->f8(w.cls) && a.f8 && w.w8) {  // cross terms on fp8 planes (f8_cross.h): the A planes were written in that form for this class
        g.f8 = 1; g.a8_off = a.f8;
        g.W8 = w.w8; g.W8_lo = w.wlo8; g.S8 = w.s8; g.S8_lo = w.slo8;
    }
    g.wscale = w.wscale;
    g.M = M; g.N = w.Np; g.K = w.Kp; g.lda = lda;
    g.zero_page = c.h->zero_page;
    g.amode = MDPT_A_DENSE; g.ekind = MDPT_E_GENERIC; g.tile = c.h->gemm_tile;
    g.throughput_mode = c.split ? 1 : 0;
    g.ldc = w.Np; g.ldr = w.Np;
    return g;
}

// output planes of a GEMM: hi, lo and the form of the lo plane (16-bit residue, or the fp8 planes an F8 consumer reads: Planes::f8)
static void out_planes(GemmParams& g, const Planes& o) {
    g.out_hi = o.hi; g.out_lo = o.lo;
    g.out_f8 = o.lo ? o.f8 : 0; g.out_a8 = o.f8_a8;
}

void as_conv(GemmParams& g, int Hi, int Wi, int Cin, int Ho, int Wo, int stride) {
    g.amode = MDPT_A_CONV3;
    g.Hi = Hi; g.Wi = Wi; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.cstride = stride;
}


// ---- stage: patch embed (fused form: writes the residual stream incl. position embedding)
int run_pos(const Ctx& c) {
    const mdpt_handle* h = c.h;
    return OPLC(mdpt_launch_posembed, h->V("imgencoder.posenc.base_patch_embedding"), c.at<float>(c.p.pos), h->cfg.base_patch_grid_h,
                                h->cfg.base_patch_grid_w, c.p.gh, c.p.gw, h->F, c.s);
}

// im2col rows of the patch embedding: from an image tensor, or (mdpt_forward_bgr) straight from the caller's uint8 BGR image
int run_patchify(const Ctx& c, const void* image, int image_dtype, const Planes& im, int H, int W) {
    const mdpt_handle* h = c.h;
    if (c.bgr.ptr)
        return OPLC(mdpt_launch_prepare_patchify, c.bgr.ptr, c.bgr.round_dtype, im.hi, im.lo, c.bgr.ih, c.bgr.iw, H, W, h->P, h->Kpatch, c.bgr.mean, c.bgr.inv_std, c.bgr.interp, c.s);
    return OPLC(mdpt_launch_patchify, image, image_dtype, im.hi, im.lo, c.p.B, H, W, h->P, h->Kpatch, c.s, c.poison);
}

int run_patch_embed_fused(const Ctx& c, const void* image, int image_dtype) {
    const mdpt_handle* h = c.h;
    const Plan& p = c.p;
    Planes im = c.pl(p.im2col);
    CHK(run_patchify(c, image, image_dtype, im, p.H, p.W));
    const bool beit = is_beit(h);
    if (!beit && !c.consts_cached) CHK(run_pos(c));
    CHK(OPLC(mdpt_launch_init_tokens, c.at<float>(p.resid), h->V("imgencoder.cls_token"), beit ? nullptr : h->V("imgencoder.posenc.cls_embedding"),
                                p.B, p.N, p.npad, h->F, c.s));
    GemmParams g = base_params(c, h->M("patch_embed.proj.weight"), im, p.B * p.Np, h->Kpatch);
    g.ekind = MDPT_E_PATCH;
    g.bias = h->V("patch_embed.proj.bias");
    g.pos = beit ? nullptr : c.at<float>(p.pos);
    g.out_f32 = c.at<float>(p.resid);
    g.tok_np = p.Np; g.npad = p.npad; g.ldc = h->F;
    CHK(OPLC(mdpt_launch_gemm, g, c.s));
    return 0;
}

// Token-mean compensation of the weight rounding for one single-pass Linear of the encoder (fp16 operand modes). The GEMM computes
// A_r W_r^T; the lost part A_r (W - W_r)^T is dominated by what all tokens of an image share, mean_t(A_r) (W - W_r)^T - a per-image bias.
// Two small launches - the column means of every step-th token as a [B, K] operand, and the skinny [B, K] x [N, K]^T product with W_lo (the lo plane
// the pack kernel already produces for the 3-pass modes) - build the table bias_img[b][n] = bias[n] + sum_k mean_t(A_r[b,t,k]) W_lo[n][k],
// and the big GEMM's epilogue adds row (m / npad) of it instead of the bias vector. Measured on ViT-L
// (tests/precision_budget/, profiles/r04_precision_budget.md): QKV error -90 %, proj -50 %, fc1 / fc2 -35 ... 40 %, for ~2 % of the step.
// Measured and not kept (round 5, profiles/r05_kernel_share_mixed_lnsum.txt): the column sums of a LayerNorm's output taken inside the LayerNorm
// launch (extra workgroups re-normalising the sampled rows) instead of mdpt_launch_colmean - the LayerNorm grew by 15 us and the table kernel,
// reading fp32 partial sums, by 8 us against the 9 us launch it saved.
int wrc_bias(const Ctx& c, GemmParams& g, const Mat& w, const float* bias, int rows_per_img, int nreal) {  // (0, 0: the ViT families' padded token rows)
    const mdpt_handle* h = c.h;
    if (!h->wrc(w.cls) || !w.lo) return 0;
    const Plan& p = c.p;
    if (rows_per_img <= 0) { rows_per_img = p.npad; nreal = p.N; }
    op_t* mean = c.at<op_t>(p.wrc_mean);
    float* tab = c.at<float>(p.wrc_tab);
    // every step-th token estimates the shared component as well as all of them (tests/precision_budget/); the step depends on the token
    // count only, so an image's table does not depend on the batch it is part of
    const int step = nreal >= 1024 ? 8 : (nreal >= 256 ? 4 : 1);
    CHK(OPLC(mdpt_launch_colmean, g.A_hi, g.lda, p.B, rows_per_img, nreal, step, w.Kp, mean, c.s));
    CHK(OPLC(mdpt_launch_wrc_table, mean, w.lo, bias, tab, p.B, w.Np, w.Kp, c.s, w.wscale));
    g.bias = tab; g.bias_img_stride = w.Np; g.bias_img_rows = rows_per_img;
    return 0;
}

// Latency mode (mdpt_set_latency_mode), fc2 of a small batch: on the 64x64 tile every workgroup walks all of K (ViT-L: 64 K tiles) as one serial
// chain of L2 round trips while most of the MFMA pipes idle. Splitting K in two fixed halves doubles the workgroups in flight and halves
// the chain; half 1's partial sums are added by the LayerNorm that follows anyway (mdpt_launch_layernorm_addp) - no reduction launch. The
// summation order differs from the default path's (hence latency mode) but not with the batch: the split is fixed.
bool fc2_ksplit_fits(int rows, int F) { return (long)((rows + 127) / 128) * ((F + 127) / 128) <= 330; }  // = resolve_tile()'s 64x64 range (gemm.hip)
constexpr int KS_MAX_PARTS = 3;  // partial-sum planes the plan reserves (a split in four)

// ---- stage: encoder. taps_f32 != null: also emit fp32 copies of the 4 out-normed taps (reference layout)
int run_encoder(const Ctx& c, void* const taps_f32[4]) {
    const mdpt_handle* h = c.h;
    const Plan& p = c.p;
    const int F = h->F, rows = p.B * p.npad;
    float* resid = c.at<float>(p.resid);
    Planes xn = c.pl(p.xn), q = c.pl(p.q), k = c.pl(p.k), vt = c.pl(p.vt), att = c.pl(p.att), hb = c.pl(p.hbuf);
    // (grid cache: the pad columns of Vt and a zero lo plane are written by nobody else - they stay what the previous forward left)
    if (!c.consts_cached) CHK(OPLC(mdpt_launch_zero_vt_pad, vt.hi, vt.lo, p.B * h->heads * 64, p.N, p.npadv, c.s));
    // a 3-pass projection behind a 1-pass attention kernel (which writes no lo plane): the plane is zero, i.e. the projection keeps
    // the rounding of its A operand and loses only that of its weights
    if (att.lo && !h->x3c(CLS_ATTN) && !c.consts_cached) CHK(hipMemsetAsync(att.lo, 0, (size_t)rows * F * 2, c.s));
    const Planes xn_qkv = {xn.hi, h->alo(CLS_QKV) ? xn.lo : nullptr}, xn_fc1 = {xn.hi, h->alo(CLS_FC1) ? xn.lo : nullptr};
#define DBG_STOP(step) if (h->dbg_block == b && h->dbg_step == (step)) return 0
    const size_t relpos_stride = is_beit(h) ? (size_t)h->heads * mdpt_beit_relpos_elen(p.gh, p.gw) : 0;
    const bool relpos_batched = is_beit(h) && h->nblocks <= 32;
    if (relpos_batched && !c.consts_cached) {  // every block's relative-position table, resized to the current grid: one launch per forward (per grid with the cache)
        BeitRelposBatch rb;
        memset(&rb, 0, sizeof(rb));
        for (int b = 0; b < h->nblocks; ++b) rb.ref[b] = h->V(blk_name(h, b) + ".attn.relpos_enc.ref_bias_lut");
        rb.ext0 = c.at<float>(p.relpos_lut); rb.ext_stride = relpos_stride;
        rb.tq = c.at<int>(p.relpos_tq); rb.tk = c.at<int>(p.relpos_tk);
        rb.n = h->nblocks; rb.heads = h->heads; rb.Gh = h->cfg.base_patch_grid_h; rb.Gw = h->cfg.base_patch_grid_w;
        rb.gh = p.gh; rb.gw = p.gw; rb.N = p.N; rb.ntok_pad = p.npadv;
        CHK(OPLC(mdpt_launch_beit_relpos_batch, rb, c.s));
    }
    // K-split fc2: `pending` = partial sums the residual stream still lacks; the next LayerNorm over it folds them in
    const float* pending = nullptr;
    int npending = 0;
    auto layernorm = [&](const float* gamma, const float* beta, op_t* ohi, op_t* olo, float* of32, size_t of8 = 0, int oa8 = 0) -> int {
        if (!pending) return OPLC(mdpt_launch_layernorm, resid, gamma, beta, ohi, olo, of32, rows, F, c.s, of8, oa8);
        const float* part = pending;
        pending = nullptr;
        return OPLC(mdpt_launch_layernorm_addp, resid, part, (size_t)rows * F, npending, gamma, beta, ohi, olo, of32, rows, F, c.s, of8, oa8);
    };
    // latency mode, small batch, enough K tiles; never in a debug-stop run. Two ranges from ks_min_ktiles K tiles on, four from ks_big_ktiles on
    auto ksplit_setup = [&](GemmParams& g) -> bool {
        const int ktiles = g.K / 64;
        if (!(h->latency_mode && p.kspart != SIZE_MAX && h->gemm_tile == MDPT_TILE_AUTO && (ktiles >= h->ks_min_ktiles || ktiles >= h->ks_big_ktiles) && h->dbg_block < 0)) return false;
        g.ksplit = ktiles >= h->ks_big_ktiles ? 4 : 2;
        if (ktiles % g.ksplit) { g.ksplit = 0; return false; }
        g.ks_part = c.at<float>(p.kspart);
        return true;
    };
    for (int b = 0; b < h->nblocks; ++b) {
        const std::string n = blk_name(h, b);
        CHK(layernorm(h->V(n + ".norm1.weight"), h->V(n + ".norm1.bias"), xn_qkv.hi, xn_qkv.lo, nullptr));
        DBG_STOP(0);
        {
            GemmParams g = base_params(c, h->M(n + ".attn.qkv.weight"), xn, rows, F);
            g.M_alg = p.B * p.N;
            g.ekind = MDPT_E_QKV;
            g.bias = h->V(is_beit(h) ? n + ".attn.qkv.bias@qv" : n + ".attn.qkv.bias");
            g.q_hi = q.hi; g.q_lo = q.lo; g.k_hi = k.hi; g.k_lo = k.lo; g.vt_hi = vt.hi; g.vt_lo = vt.lo;
            g.F = F; g.heads = h->heads; g.npad = p.npad; g.npadv = p.npadv; g.qscale = 0.125f;
            CHK(wrc_bias(c, g, h->M(n + ".attn.qkv.weight"), g.bias));
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
        }
        DBG_STOP(1);
        {
            AttnParams a;
            memset(&a, 0, sizeof(a));
            a.q_hi = q.hi; a.q_lo = q.lo; a.k_hi = k.hi; a.k_lo = k.lo; a.vt_hi = vt.hi; a.vt_lo = vt.lo;
            a.out_hi = att.hi; a.out_lo = att.lo;
            a.B = p.B; a.heads = h->heads; a.N = p.N; a.npad = p.npad; a.npadv = p.npadv; a.F = F; a.x3 = h->x3c(CLS_ATTN);
            a.allow_split_kv = h->latency_mode;
            if (is_beit(h)) {
                float* lut_b = c.at<float>(p.relpos_lut) + (relpos_batched ? (size_t)b * relpos_stride : 0);
                if (!relpos_batched)  // more than 32 blocks: this layer's table on its own (tiny kernel)
                    CHK(OPLC(mdpt_launch_beit_relpos, h->V(n + ".attn.relpos_enc.ref_bias_lut"), lut_b, c.at<int>(p.relpos_tq), c.at<int>(p.relpos_tk),
                                                h->heads, h->cfg.base_patch_grid_h, h->cfg.base_patch_grid_w, p.gh, p.gw, p.N, p.npadv, c.s));
                a.bias_lut = lut_b; a.bias_elen = mdpt_beit_relpos_elen(p.gh, p.gw);
                a.tq = c.at<int>(p.relpos_tq); a.tk = c.at<int>(p.relpos_tk);
            }
            if (c.attn_dump && c.attn_dump[b]) CHK(OPLC(mdpt_launch_attn_weights, a, (float*)c.attn_dump[b], c.s));
            CHK(OPLC(mdpt_launch_attention, a, c.s));
        }
        DBG_STOP(2);
        {
            GemmParams g = base_params(c, h->M(n + ".attn.proj.weight"), att, rows, F);
            g.M_alg = p.B * p.N;
            g.bias = h->V(n + ".attn.proj.bias@ls");  // layer scale folded into W and the bias at pack time
            g.acc_init = 1;
            g.resid = resid; g.out_f32 = resid; g.ldr = F; g.ldc = F;
            CHK(wrc_bias(c, g, h->M(n + ".attn.proj.weight"), g.bias));
            if (ksplit_setup(g)) { pending = g.ks_part; npending = g.ksplit - 1; }  // LN2 is the next reader of the residual stream
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
        }
        DBG_STOP(3);
        CHK(layernorm(h->V(n + ".norm2.weight"), h->V(n + ".norm2.bias"), xn_fc1.hi, xn_fc1.lo, nullptr));
        DBG_STOP(4);
        if (h->gh_hidden) {  // ViT-G: (a | b) = x W12^T + b12 ; hidden = silu(a) * b
            GemmParams g = base_params(c, h->M(n + ".mlp.inner_linear_doubled.weight"), xn, rows, F);
            g.M_alg = p.B * p.N;
            g.bias = h->V(n + ".mlp.inner_linear_doubled.bias");
            g.out_f32 = c.at<float>(p.swi); g.ldc = 2 * h->gh_hidden;
            CHK(wrc_bias(c, g, h->M(n + ".mlp.inner_linear_doubled.weight"), g.bias));
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
            CHK(OPLC(mdpt_launch_swiglu, c.at<float>(p.swi), hb.hi, hb.lo, (size_t)rows, h->gh_hidden, h->gh_hidden_p, c.s));
        } else {
            GemmParams g = base_params(c, h->M(n + ".mlp.layers.0.weight"), xn, rows, F);
            g.M_alg = p.B * p.N;
            g.bias = h->V(n + ".mlp.layers.0.bias");
            g.act = MDPT_ACT_GELU;
            g.out_hi = hb.hi; g.out_lo = hb.lo; g.ldc = 4 * F;
            CHK(wrc_bias(c, g, h->M(n + ".mlp.layers.0.weight"), g.bias));
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
        }
        DBG_STOP(5);
        const bool v1 = h->cfg.family == MDPT_FAMILY_DAV1;
        const bool is_tap = v1 ? b >= h->nblocks - 4 : (b + 1) % h->bps == 0;
        {
            const bool giant = h->gh_hidden != 0;
            GemmParams g = base_params(c, h->M(giant ? n + ".mlp.outer_linear.weight" : n + ".mlp.layers.2.weight"), hb, rows,
                                       giant ? h->gh_hidden_p : 4 * F);
            g.M_alg = p.B * p.N;
            g.bias = h->V(giant ? n + ".mlp.outer_linear.bias@ls" : n + ".mlp.layers.2.bias@ls");
            g.acc_init = 1;
            g.resid = resid; g.out_f32 = resid; g.ldr = F; g.ldc = F;
            CHK(wrc_bias(c, g, h->M(giant ? n + ".mlp.outer_linear.weight" : n + ".mlp.layers.2.weight"), g.bias));
            // a LayerNorm must be the next reader of the residual stream: not when this block's raw output is exported (block hooks, BEiT's
            // un-normed taps, a debug stop) or nothing follows (BEiT's last block is a tap)
            if (!(c.block_dump && c.block_dump[b]) && !(is_beit(h) && is_tap) && (b + 1 < h->nblocks || is_tap) && ksplit_setup(g)) {
                pending = g.ks_part; npending = g.ksplit - 1;
            }
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
        }
        DBG_STOP(6);
        if (c.block_dump && c.block_dump[b])  // TransformerBlock output (transformer_block.py:61-62), pad rows dropped
            CHK(OPLC(mdpt_launch_tokens_export, nullptr, nullptr, resid, (float*)c.block_dump[b], p.B, p.N, p.npad, F, 0, c.s));
        if (is_tap) {
            const int st = v1 ? b - (h->nblocks - 4) : b / h->bps;
            Planes tp = c.pl(p.tap[st]);
            float* f32 = taps_f32 ? c.at<float>(p.tapf32) : nullptr;
            if (is_beit(h)) {  // BEiT taps the raw residual stream (no out-norm, v31_beit/image_encoder_model.py:84-91)
                CHK(OPLC(mdpt_launch_tokens_import, resid, tp.hi, tp.lo, p.B, p.npad, p.npad, F, c.s, tp.lo ? tp.f8 : 0, tp.f8_a8));
                if (taps_f32) CHK(OPLC(mdpt_launch_tokens_export, nullptr, nullptr, resid, (float*)taps_f32[st], p.B, p.N, p.npad, F, 0, c.s));
            } else {
                CHK(layernorm(h->V("imgencoder.outnorm.weight"), h->V("imgencoder.outnorm.bias"), tp.hi, tp.lo, f32, tp.lo ? tp.f8 : 0, tp.f8_a8));
                if (taps_f32)
                    CHK(OPLC(mdpt_launch_tokens_export, nullptr, nullptr, f32, (float*)taps_f32[st], p.B, p.N, p.npad, F, 0, c.s));
            }
            if (c.tap_stream) {
                // small batches: the forward is a chain of ~215 dependent launches on a mostly idle GPU. Tap st's reassembly branch (3-4 launches)
                // depends on nothing the rest of the encoder writes, so it goes to the side stream now and runs beside the next blocks; the branches
                // share that one stream (BEiT's readout buffers are reused from branch to branch). Same kernels, same bits.
                CHK(hipEventRecord(c.tap_event, c.s));
                CHK(hipStreamWaitEvent(c.tap_stream, c.tap_event, 0));
                Ctx cs = c;
                cs.s = c.tap_stream; cs.tap_stream = nullptr; cs.side = true;
                CHK(run_reassemble_stage(cs, st));
                // ... and the first conv of the fusion block's conv_reassembly unit (fusion_model.py:148-150), which reads that branch's map only
                if (st < 3) CHK(fusion_rcu_a_first(cs, st));
            }
        }
    }
    return 0;
}

int conv3_to_fusion(const Ctx& c, const Mat& w, Planes in, int Cin, int sh, int sw, const float* bias, const float* skip, const float* up_src,
                    int Hu, int Wu, float* out_f32, Planes out, int relu_bf16);

// Latency mode, small batch: a generic-epilogue conv / GEMM of FEW 64x64 workgroups that each walk a LONG K (the 3x3 convs of the 18^2 / 36^2 levels at
// batch 1: 24 ... 96 workgroups x 72 ... 144 K tiles, 33 ... 65 us each) runs as K ranges over grid.y that all store bare partial planes, and a small
// finishing kernel adds them in the order z = 0, 1, ... and applies bias / ReLU / the output planes - the launch boundary is the fence between the
// ranges and their sum (an in-kernel reduction needs a device-scope release per workgroup that costs what the split saves: DESIGN.md section 3).
// Plain epilogues only (no skip, no upsample-add, no activation other than the planes' ReLU). *done = false: the caller launches the GEMM itself.
int try_ksplit_conv(const Ctx& c, const GemmParams& g, bool* done) {
    *done = false;
    const mdpt_handle* h = c.h;
    const Plan& p = c.p;
    if (!h->latency_mode || c.split || c.side || p.kspart == SIZE_MAX || h->gemm_tile != MDPT_TILE_AUTO || h->dbg_block >= 0) return 0;  // (side: kspart belongs to the encoder running beside it)
    if (g.ekind != MDPT_E_GENERIC || g.resid || g.up_src || g.gamma || g.acc_init || g.act != MDPT_ACT_NONE || g.bias_img_stride || g.f8) return 0;  // (f8: 128-element K tiles, no K split)
    const long tiles = (long)((g.M + 63) / 64) * ((g.N + 63) / 64);
    const int kt = g.K / 64;
    if (tiles >= 256 || kt < 64) return 0;
    int ks = 1;
    for (int d = 2; d <= 8 && kt / d >= 8; ++d)
        if (kt % d == 0) { ks = d; if (tiles * d >= 256) break; }
    const size_t plane = (size_t)g.M * g.ldc;
    if (ks < 2 || plane * ks * 4 > (size_t)p.B * p.npad * h->F * 4 * 3) return 0;  // (the three partial planes of the plan: kspart)
    float* part = c.at<float>(p.kspart);
    GemmParams q = g;
    q.ksplit = ks; q.ks_all = 1; q.ks_part = part;
    q.bias = nullptr; q.out_f32 = nullptr; q.out_hi = nullptr; q.out_lo = nullptr; q.out_f8 = 0; q.out_a8 = 0;
    CHK(OPLC(mdpt_launch_gemm, q, c.s));
    CHK(OPLC(mdpt_launch_ksplit_finish, part, plane, ks, g.bias, g.out_f32, g.out_hi, g.out_lo, g.relu_bf16, g.M, g.N, g.ldc, c.s, g.out_f8, g.out_a8));
    *done = true;
    return 0;
}

// ---- stage: reassemble. One branch per encoder tap; the four branches are independent of each other (reassembly_model.py:61-94), which the
//      small-batch forward uses: branch i is queued on the side stream as soon as tap i exists (Ctx::tap_stream, mdpt_api.cpp forward_one)
int run_reassemble(const Ctx& c) {
    for (int i = 0; i < 4; ++i) CHK(run_reassemble_stage(c, i));
    return 0;
}

int run_reassemble_stage(const Ctx& c, int i) {
    const mdpt_handle* h = c.h;
    const Plan& p = c.p;
    const int F = h->F, gh = p.gh, gw = p.gw;
    {
        const std::string n = std::string("reassemble.") + kStageNames[i];
        const int hp = h->hidp[i];
        Planes tp = c.pl(p.tap[i]), t = c.pl(p.t[i]);
        bool tokens_mode = true;
        if (is_beit(h)) {
            // readout projection: GELU(W [tok ; cls] + b) = GELU(W_tok tok + (W_cls cls + b)); the cls term is one row per image
            {
                GemmParams g = base_params(c, h->M(n + ".readout_proj.1.weight@cls"), tp, p.B, p.npad * F);  // row b = cls token of image b
                g.bias = h->V(n + ".readout_proj.1.bias");
                g.out_f32 = c.at<float>(p.cbuf); g.ldc = F;
                CHK(OPLC(mdpt_launch_gemm, g, c.s));
            }
            Planes tr = c.pl(p.tokr);
            {
                GemmParams g = base_params(c, h->M(n + ".readout_proj.1.weight"), tp, p.B * p.Np, F);
                g.amode = MDPT_A_TOKENS; g.tok_np = p.Np; g.tok_stride = p.npad;
                g.bias = c.at<float>(p.cbuf); g.bias_img_stride = F; g.bias_img_rows = p.Np;
                g.act = MDPT_ACT_GELU;
                out_planes(g, tr); g.ldc = F;
                CHK(OPLC(mdpt_launch_gemm, g, c.s));
            }
            tp = tr;
            tokens_mode = false;
        }
        {   // 1x1 conv on the patch tokens (cls row skipped by the A-row generator)
            GemmParams g = base_params(c, h->M(n + ".resample.0.weight"), tp, p.B * p.Np, F);
            if (tokens_mode) { g.amode = MDPT_A_TOKENS; g.tok_np = p.Np; g.tok_stride = p.npad; }
            g.bias = h->V(n + ".resample.0.bias");
            out_planes(g, t); g.ldc = hp;
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
        }
        Planes src = t;
        int sh = gh, sw = gw;
        if (i == 0 || i == 1) {  // ConvTranspose2d k == s: GEMM + depth-to-space
            const int kk = i == 0 ? 4 : 2;
            Planes u = c.pl(i == 0 ? p.u0 : p.u1);
            GemmParams g = base_params(c, h->M(n + ".resample.1.weight"), t, p.B * p.Np, hp);
            g.ekind = MDPT_E_D2S;
            g.bias = h->V(n + ".resample.1.bias");
            g.Ho = gh; g.Wo = gw; g.d2s_k = kk; g.d2s_cout = hp;
            out_planes(g, u);
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
            src = u; sh = gh * kk; sw = gw * kk;
        } else if (i == 3) {  // 3x3 stride-2
            Planes d = c.pl(p.d3);
            GemmParams g = base_params(c, h->M(n + ".resample.1.weight"), t, p.B * (gh / 2) * (gw / 2), hp);
            as_conv(g, gh, gw, hp, gh / 2, gw / 2, 2);
            g.bias = h->V(n + ".resample.1.bias");
            out_planes(g, d); g.ldc = hp;
            bool split_done = false;
            CHK(try_ksplit_conv(c, g, &split_done));
            if (!split_done) CHK(OPLC(mdpt_launch_gemm, g, c.s));
            src = d; sh = gh / 2; sw = gw / 2;
        }
        {   // 3x3 projection to the fusion width (no bias): fp32 copy (skip path) + ReLU'd bf16 (next conv input)
            CHK(conv3_to_fusion(c, h->M(n + ".fuse_proj.weight"), src, hp, sh, sw, nullptr, nullptr, nullptr, 0, 0, c.at<float>(p.r_f32[i]),
                                c.pl(p.r_bf[i]), 1));
        }
    }
    return 0;
}

// Halo-staged conv kernel (conv3h.hip) for a 3x3 stride-1 conv to the 256-wide fusion width, used for big launches; small ones run the
// implicit-GEMM kernels of gemm.hip. Both walk K in the same order and apply the same epilogue expressions (((conv + bias) + up) + skip),
// so one image's bits do not depend on the batch it is part of. Partial 16x16 tiles may waste at most 25 % of the MFMA work (72x72: 25
// tiles for 20.25 image-tiles' worth of pixels - the halo-staged loop is ~30 % faster per K tile; 36x36: 9 for 5.06 -> implicit GEMM).
// From how many 256-row tiles' worth of output pixels the halo-staged kernel replaces the implicit GEMM: measured on the bare kernels at
// batch 1 / 2 / 4 / 8 (profiles/r04_conv3h_small_batch.txt, tools/probes/gpu_conv3h_small_batch.py) - 144^2 x 1 image (81): 45.2 vs 48.8 us,
// 72^2 x 4 (81): 46.9 vs 49.4, 72^2 x 2 (41): 43.2 vs 27.5 (one workgroup per tile: few tiles leave the CUs idle). Under the two-stream
// batch split the other half fills idle CUs, so the faster-per-tile kernel is taken earlier. (The dense GEMMs' tile rule in gemm.hip has its
// own thresholds, 140 / 70: there the big tile competes with a 64x64 tile that is good at small sizes; here the alternative is slower per K tile.)
inline long conv3h_min_tiles(const Ctx& c) { return c.split ? 24 : 80; }

bool conv3h_shape_ok(const mdpt_handle* h, int H, int W, int Cin) {
    if (h->Cp != 256 || (Cin & 127) || H < 2 || W < 2) return false;
    const long tile_px = (long)((H + 15) / 16) * ((W + 15) / 16) * 256, px = (long)H * W;
    return tile_px * 4 <= px * 5;
}

// one 3x3 stride-1 conv Cin -> Cp: out = [skip +] conv(in) [+ bias] [+ up2(up_src)] -> fp32 map and / or bf16 planes (ReLU'd if relu_bf16)
int conv3_to_fusion(const Ctx& c, const Mat& w, Planes in, int Cin, int sh, int sw, const float* bias, const float* skip, const float* up_src,
                    int Hu, int Wu, float* out_f32, Planes out, int relu_bf16) {
    const mdpt_handle* h = c.h;
    const bool eligible = conv3h_shape_ok(h, sh, sw, Cin);
    if (eligible && h->gemm_tile == MDPT_TILE_AUTO) {
        Conv3hParams q;
        memset(&q, 0, sizeof(q));
        const int np = h->terms(w.cls);  // the weight's class decides (an input buffer may carry a lo plane this conv does not use)
        const bool f8 = np >= 2 && h->f8(w.cls) && in.f8 && w.w8;  // cross terms on fp8 planes (f8_cross.h)
        q.in = in.hi; q.in_lo = np >= 2 ? in.lo : nullptr; q.w = w.hi; q.w_lo = np == 3 && !f8 ? w.lo : nullptr; q.bias = bias; q.skip = skip; q.up_src = up_src; q.Hu = Hu; q.Wu = Wu;
        if (f8) { q.f8 = 1; q.a8_off = in.f8; q.w8 = w.w8; q.w8_lo = np == 3 ? w.wlo8 : nullptr; q.s8 = w.s8; q.s8_lo = np == 3 ? w.slo8 : nullptr; }
        q.out_f32 = out_f32; q.out_bf = out.hi; q.out_bf_lo = out.lo; q.relu_bf = relu_bf16;
        q.out_f8 = out.lo ? out.f8 : 0; q.out_a8 = out.f8_a8;
        q.B = c.p.B; q.H = sh; q.W = sw; q.Cin = Cin; q.Cout = 256;
        const long tiles256 = ((long)c.p.B * sh * sw + 255) / 256;
        if (tiles256 >= conv3h_min_tiles(c) && OPLC(mdpt_conv3h_supported, q)) return OPLC(mdpt_launch_conv3h, q, c.s);
    }
    GemmParams g = base_params(c, w, in, c.p.B * sh * sw, Cin);
    as_conv(g, sh, sw, Cin, sh, sw, 1);
    g.bias = bias;
    g.resid = skip; g.ldr = h->Cp;
    g.up_src = up_src; g.Hu = Hu; g.Wu = Wu;
    g.out_f32 = out_f32; out_planes(g, out); g.relu_bf16 = relu_bf16; g.ldc = h->Cp;
    bool split_done = false;
    CHK(try_ksplit_conv(c, g, &split_done));
    if (split_done) return 0;
    return OPLC(mdpt_launch_gemm, g, c.s);
}

// one 3x3 conv C->C of a residual conv unit at level `lv` (spatial sh x sw)
int rcu_conv(const Ctx& c, const std::string& wname, Planes in, int sh, int sw, const float* skip, const float* up_src, int Hu, int Wu,
             float* out_f32, Planes out, int relu_bf16) {
    const mdpt_handle* h = c.h;
    return conv3_to_fusion(c, h->M(wname + ".weight"), in, h->Cp, sh, sw, h->V(wname + ".bias"), skip, up_src, Hu, Wu, out_f32, out, relu_bf16);
}

// ---- stage: fusion. Level index i: 3 = coarsest (gh/2), 0 = finest (4gh). Output: flo[0] (fp32, 4gh x 4gw, before the
//      final x2 upsample) and `fused` planes (8gh x 8gw).
// bf16 mode, forward path (for_head): the last projection (level 0) writes its output as bf16 and the x2 upsample in front of the head is
// left to run_head, which either interpolates it inside the head's first conv (halo-staged kernel, big launches) or runs the stand-alone
// bf16 upsample - same arithmetic, same bits (up_bf16.h). The stage-level API and the bf16x3 mode keep the fp32 map + fp32 upsample.
bool head_upsamples_bf16(const mdpt_handle* h) { return h->terms(CLS_HEAD) == 1 && (h->Cp & 7) == 0; }
// everything behind the head's first conv as ONE kernel (head.hip: upsample + 3x3 conv + ReLU + 1x1 + ReLU | sigmoid out of LDS tiles): the
// single-pass form of the tail class (conv 1 then writes a 16-bit map whatever its own pass count) and its two-pass form (activations
// split: conv 1 writes hi + lo 16-bit planes, head_tail2_kernel). Three passes run the unfused kernels.
bool head_tail_fused(const mdpt_handle* h) { return h->terms(CLS_HEAD_TAIL) <= 2 && mdpt_head_tail_supported(h->C2p); }

// first conv of level i's conv_reassembly unit: relu(r_i) -> 3x3 conv -> ReLU'd planes a1[i] (fusion_model.py:148-150, 210-220)
int fusion_rcu_a_first(const Ctx& c, int i) {
    const mdpt_handle* h = c.h;
    const Plan& p = c.p;
    const int sh[4] = {4 * p.gh, 2 * p.gh, p.gh, p.gh / 2}, sw[4] = {4 * p.gw, 2 * p.gw, p.gw, p.gw / 2};
    char pb[64];
    snprintf(pb, sizeof(pb), "fusion.blocks.%d", i);
    return rcu_conv(c, std::string(pb) + ".conv_reassembly." + rcu_seq(h) + ".1", c.pl(p.r_bf[i]), sh[i], sw[i], nullptr, nullptr, 0, 0, nullptr, c.pl(p.a1[i]), 1);
}

int run_fusion(const Ctx& c, bool for_head) {
    const mdpt_handle* h = c.h;
    const Plan& p = c.p;
    const int sh[4] = {4 * p.gh, 2 * p.gh, p.gh, p.gh / 2}, sw[4] = {4 * p.gw, 2 * p.gw, p.gw, p.gw / 2};
    for (int i = 3; i >= 0; --i) {
        char pb[64];
        snprintf(pb, sizeof(pb), "fusion.blocks.%d", i);
        const std::string blk = pb;
        const float* x_f32;
        Planes x_bf;
        if (i == 3) {  // top-most block: no reassembly RCU, no prior (fusion_model.py:89-114)
            x_f32 = c.at<float>(p.r_f32[3]);
            x_bf = c.pl(p.r_bf[3]);
        } else {
            // x = RCU_a(r_i) + up2(prev)   (fusion_model.py:148-154)
            Planes a1 = c.pl(p.a1[i]);
            if (!c.a1_done) CHK(fusion_rcu_a_first(c, i));
            x_bf = c.pl(p.x_bf[i]);
            CHK(rcu_conv(c, blk + ".conv_reassembly." + rcu_seq(h) + ".3", a1, sh[i], sw[i], c.at<float>(p.r_f32[i]), c.at<float>(p.flo[i + 1]),
                         sh[i + 1], sw[i + 1], c.at<float>(p.x_f32[i]), x_bf, 1));
            x_f32 = c.at<float>(p.x_f32[i]);
        }
        Planes b1 = c.pl(p.b1[i]), b2 = c.pl(p.b2[i]);
        CHK(rcu_conv(c, blk + "." + proj_seq(h) + ".0." + rcu_seq(h) + ".1", x_bf, sh[i], sw[i], nullptr, nullptr, 0, 0, nullptr, b1, 1));
        CHK(rcu_conv(c, blk + "." + proj_seq(h) + ".0." + rcu_seq(h) + ".3", b1, sh[i], sw[i], x_f32, nullptr, 0, 0, nullptr, b2, 0));
        {   // 1x1 projection at LOW resolution; the x2 bilinear upsample commutes with it exactly (both linear, weights
            // sum to 1) and is applied by the consumer (next level's epilogue / final upsample kernel)
            GemmParams g = base_params(c, h->M(blk + "." + proj_seq(h) + ".2.weight"), b2, p.B * sh[i] * sw[i], h->Cp);
            g.bias = h->V(blk + "." + proj_seq(h) + ".2.bias");
            if (i == 0 && for_head && head_upsamples_bf16(h)) g.out_hi = c.at<op_t>(p.flo[0]);  // bf16 map in the fp32 map's buffer
            else g.out_f32 = c.at<float>(p.flo[i]);
            g.ldc = h->Cp;
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
        }
    }
    if (for_head && head_upsamples_bf16(h)) return 0;
    Planes fu = c.pl(p.fused);
    CHK(OPLC(mdpt_launch_upsample, c.at<float>(p.flo[0]), fu.hi, fu.lo, nullptr, p.B, sh[0], sw[0], 2 * sh[0], 2 * sw[0], h->Cp, c.s, fu.lo ? fu.f8 : 0, fu.f8_a8));
    return 0;
}

// ---- stage: head
// from_flo0b: the head's input is still the 16-bit output of the last fusion projection at half resolution (run_fusion(c, true))
// Two op classes: conv 1 (CLS_HEAD: 3x3, C -> C/2 at 8gh x 8gw) and the tail (CLS_HEAD_TAIL: x(P/8) upsample, 3x3 conv C/2 -> 32 + ReLU, 1x1 + ReLU |
// sigmoid). A single-pass tail is ONE kernel fed by a 16-bit map conv 1 writes whatever its own pass count; a multi-pass tail runs
// upsample kernel + implicit-GEMM conv with the fused 32 -> 1 epilogue on hi + lo planes of the upsampled fp32 map.
int run_head(const Ctx& c, void* depth, int depth_dtype, bool from_flo0b) {
    const mdpt_handle* h = c.h;
    const Plan& p = c.p;
    const int fh = 8 * p.gh, fw = 8 * p.gw;
    const int np1 = h->terms(CLS_HEAD);
    const Mat& w1 = h->M("head.spatial_upsampler.0.weight");
    const float* b1 = h->V("head.spatial_upsampler.0.bias");
    const bool tail_fused = head_tail_fused(h) && mdpt_head_tail_scale_ok(fh, fw, p.H, p.W);
    const bool halo_ok = h->C2p == 128 && conv3h_shape_ok(h, fh, fw, h->Cp) && h->gemm_tile == MDPT_TILE_AUTO;
    const bool big = ((long)p.B * fh * fw + 255) / 256 >= conv3h_min_tiles(c);
    bool fused_ready = !from_flo0b;
    auto materialise_fused = [&]() -> int {  // stand-alone bf16 upsample (small launches / shapes the fused kernel does not cover)
        if (!fused_ready) CHK(OPLC(mdpt_launch_upsample_bf16, c.at<op_t>(p.flo[0]), c.pl(p.fused).hi, p.B, fh / 2, fw / 2, fh, fw, h->Cp, c.s));
        fused_ready = true;
        return 0;
    };
    // ---- conv 1 -> a 16-bit map (fused tail; the buffer of the fp32 map is reused; hi + lo planes for the two-pass tail) or the fp32 map
    //      (the unfused tail's own upsample reads it)
    op_t* h1b = tail_fused ? c.at<op_t>(p.h1) : nullptr;
    op_t* h1b_lo = tail_fused && h->terms(CLS_HEAD_TAIL) == 2 ? h1b + (size_t)p.B * fh * fw * h->C2p : nullptr;
    float* h1f = tail_fused ? nullptr : c.at<float>(p.h1);
    bool done = false;
    if (halo_ok && big) {  // halo-staged form, 128 output channels
        Conv3hParams q;
        memset(&q, 0, sizeof(q));
        const bool f8h = np1 >= 2 && h->f8(CLS_HEAD) && w1.w8;  // cross terms on fp8 planes (the `fused` planes were written in that form)
        q.w = w1.hi; q.w_lo = np1 == 3 && !f8h ? w1.lo : nullptr; q.bias = b1;
        q.out_bf = h1b; q.out_bf_lo = h1b_lo; q.out_f32 = h1f; q.B = p.B; q.H = fh; q.W = fw; q.Cin = h->Cp; q.Cout = 128;
#ifndef MDPT_NO_UPIN  // (A/B builds: -DMDPT_NO_UPIN keeps the stand-alone upsample in front of the halo-staged conv)
        if (!fused_ready && np1 == 1 && h1b) {  // single pass: the x2 upsample folded into the conv's halo interpolation
            q.up_in = c.at<op_t>(p.flo[0]); q.Hs = fh / 2; q.Ws = fw / 2;
            if (OPLC(mdpt_conv3h_supported, q)) {
                CHK(OPLC(mdpt_launch_conv3h, q, c.s));
                done = true;
            }
            q.up_in = nullptr;
        }
#endif
        if (!done) {
            CHK(materialise_fused());
            Planes fu = c.pl(p.fused);
            q.in = fu.hi; q.in_lo = np1 >= 2 ? fu.lo : nullptr;
            if (f8h && fu.f8) { q.f8 = 1; q.a8_off = fu.f8; q.w8 = w1.w8; q.w8_lo = np1 == 3 ? w1.wlo8 : nullptr; q.s8 = w1.s8; q.s8_lo = np1 == 3 ? w1.slo8 : nullptr; }
            if (OPLC(mdpt_conv3h_supported, q)) {
                CHK(OPLC(mdpt_launch_conv3h, q, c.s));
                done = true;
            }
        }
    }
    if (!done) {
        CHK(materialise_fused());
        GemmParams g = base_params(c, w1, c.pl(p.fused), p.B * fh * fw, h->Cp);
        as_conv(g, fh, fw, h->Cp, fh, fw, 1);
        g.bias = b1;
        g.out_hi = h1b; g.out_lo = h1b_lo; g.out_f32 = h1f; g.ldc = h->C2p;
        CHK(OPLC(mdpt_launch_gemm, g, c.s));
    }
    // ---- tail
    if (tail_fused) {
        HeadTailParams t;
        memset(&t, 0, sizeof(t));
        t.src = h1b; t.src_lo = h1b_lo; t.w_kc = h->M("head.proj_1ch.0.weight@kc32").hi;
        t.bias = h->V("head.proj_1ch.0.bias"); t.head_w = h->V("head.proj_1ch.2.weight"); t.head_b = h->V("head.proj_1ch.2.bias");
        t.out = depth; t.out_dtype = depth_dtype; t.sigmoid = h->cfg.is_metric;
        t.B = p.B; t.Hi = fh; t.Wi = fw; t.Ho = p.H; t.Wo = p.W;
        CHK(OPLC(mdpt_launch_head_tail, t, h->C2p, c.s));
        return 0;
    }
    Planes hu = c.pl(p.h1u);
    CHK(OPLC(mdpt_launch_upsample, h1f, hu.hi, hu.lo, nullptr, p.B, fh, fw, p.H, p.W, h->C2p, c.s));
    {
        GemmParams g = base_params(c, h->M("head.proj_1ch.0.weight"), hu, p.B * p.H * p.W, h->C2p);
        as_conv(g, p.H, p.W, h->C2p, p.H, p.W, 1);
        g.ekind = MDPT_E_HEAD;
        g.bias = h->V("head.proj_1ch.0.bias");
        g.head_w = h->V("head.proj_1ch.2.weight");
        g.head_b = h->V("head.proj_1ch.2.bias");
        g.head_sigmoid = h->cfg.is_metric;
        g.head_out = depth; g.head_out_dtype = depth_dtype;
        CHK(OPLC(mdpt_launch_gemm, g, c.s));
    }
    return 0;
}

#include "mdpt_swin_stages.inc"

}  // namespace mdpt

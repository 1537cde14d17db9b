// libmdpt: parameter inventory (reference "new format" key names), packed-weight layout and activation workspace plan.
#include "mdpt_internal.h"

namespace mdpt {

thread_local std::string g_err = "";
int g_debug_f16 = 0;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

const char* const kStageNames[4] = {"spatial_upx4", "spatial_upx2", "spatial_noscale", "spatial_downx2"};
const char* const kSwinStageNames[4] = {"spatial_noscale", "spatial_downx2", "spatial_downx4", "spatial_downx8"};

int mat_class(const std::string& src) {
    if (src.compare(0, 12, "patch_embed.") == 0) return CLS_PATCH;
    if (src.compare(0, 11, "reassemble.") == 0) return CLS_REASM;
    if (src.compare(0, 7, "fusion.") == 0) {
        if (src.find(".conv_reassembly.") != std::string::npos) return CLS_FUSION_IN;
        return src.find("proj_seq.2.") != std::string::npos ? CLS_FUSION_PROJ : CLS_FUSION;  // the 1x1 output projection | the RCU's 3x3 convs
    }
    if (src.compare(0, 5, "head.") == 0) return src.compare(0, 14, "head.proj_1ch.") == 0 ? CLS_HEAD_TAIL : CLS_HEAD;
    if (src.find(".attn.qkv.") != std::string::npos) return CLS_QKV;
    if (src.find(".attn.proj.") != std::string::npos) return CLS_PROJ;
    if (src.find(".mlp.layers.0.") != std::string::npos || src.find("inner_linear_doubled") != std::string::npos) return CLS_FC1;
    if (src.find(".mlp.layers.2.") != std::string::npos || src.find("outer_linear") != std::string::npos) return CLS_FC2;
    if (src.find("patch_merge_layers") != std::string::npos) return CLS_PROJ;  // SwinV2 patch merge: a token-mixing projection
    return CLS_PROJ;
}

std::string blk_name(const mdpt_handle* h, int block) {
    char buf[96];
    if (h->cfg.family == MDPT_FAMILY_DAV1) snprintf(buf, sizeof(buf), "imgencoder.blocks.%d", block);
    else snprintf(buf, sizeof(buf), "imgencoder.stages.%d.blocks.%d", block / h->bps, block % h->bps);
    return buf;
}


// which classes CAN run their cross terms on fp8 planes (MDPT_PASSES_2F8 / _3F8, f8_cross.h): fp16 operands, every contraction length of
// the class a multiple of the 128-element fp8 K tile, ViT / BEiT encoders (the SwinV2 tap producers write 16-bit planes only). A class that
// cannot runs the fp16-plane form of the same term count. A property of the configuration: never of the batch or the image size.
void compute_f8ok(mdpt_handle* h) {
    for (int i = 0; i < NCLS; ++i) h->f8ok[i] = false;
    const bool base = h->f16 && !h->swin;
    const bool cp = base && h->Cp % 128 == 0;
    bool re = base && h->F % 128 == 0;
    for (int i = 0; i < 4; ++i) re = re && h->hidp[i] % 128 == 0;
    h->f8ok[CLS_REASM] = re;
    h->f8ok[CLS_FUSION] = h->f8ok[CLS_FUSION_IN] = h->f8ok[CLS_FUSION_PROJ] = h->f8ok[CLS_HEAD] = cp;
}

int build_inventory_swin_encoder(mdpt_handle* h);

int build_inventory_decoder(mdpt_handle* h);

int build_inventory(mdpt_handle* h) {
    const int F = h->F, P = h->P, C = h->C;
    const int G = h->cfg.base_patch_grid_h * h->cfg.base_patch_grid_w;
    h->packed_total = 0;
    h->wrc_maxn = h->wrc_maxk = 0;
    h->zero_off = 0;
    h->packed_total += 256;
    compute_f8ok(h);

    h->add_spec("patch_embed.proj.weight", {F, 3, P, P});
    h->add_spec("patch_embed.proj.bias", {F});
    h->add_mat("patch_embed.proj.weight", MDPT_PACK_LINEAR, F, 3 * P * P, F, h->Kpatch, 0);
    h->add_vec("patch_embed.proj.bias", F, F);
    if (h->swin) {
        build_inventory_swin_encoder(h);
        return build_inventory_decoder(h);
    }

    const bool beit = is_beit(h);
    const int nlut = (2 * h->cfg.base_patch_grid_h - 1) * (2 * h->cfg.base_patch_grid_w - 1) + 3;
    h->add_spec("imgencoder.cls_token", {1, 1, F});
    h->add_vec("imgencoder.cls_token", F, F);
    if (!beit) {
        h->add_spec("imgencoder.posenc.cls_embedding", {1, 1, F});
        h->add_spec("imgencoder.posenc.base_patch_embedding", {1, G, F});
        h->add_spec("imgencoder.outnorm.weight", {F});
        h->add_spec("imgencoder.outnorm.bias", {F});
        h->add_vec("imgencoder.posenc.cls_embedding", F, F);
        h->add_vec("imgencoder.posenc.base_patch_embedding", G * F, G * F);
        h->add_vec("imgencoder.outnorm.weight", F, F);
        h->add_vec("imgencoder.outnorm.bias", F, F);
    }

    for (int b = 0; b < h->nblocks; ++b) {
        const std::string p = blk_name(h, b);
        for (const char* ln : {"norm1", "norm2"}) {
            h->add_spec(p + "." + ln + ".weight", {F});
            h->add_spec(p + "." + ln + ".bias", {F});
            h->add_vec(p + "." + ln + ".weight", F, F);
            h->add_vec(p + "." + ln + ".bias", F, F);
        }
        h->add_spec(p + ".attn.qkv.weight", {3 * F, F});
        if (beit) {  // qkv Linear has no bias; q and v get separate biases, k none (v31_beit/image_encoder_model.py:296-297,341-342)
            h->add_spec(p + ".attn.q_bias", {1, h->heads, 1, 64});
            h->add_spec(p + ".attn.v_bias", {1, h->heads, 1, 64});
            h->add_spec(p + ".attn.relpos_enc.ref_bias_lut", {nlut, h->heads});
            h->add_vec(p + ".attn.relpos_enc.ref_bias_lut", nlut * h->heads, nlut * h->heads);
        } else {
            h->add_spec(p + ".attn.qkv.bias", {3 * F});
        }
        h->add_spec(p + ".attn.proj.weight", {F, F});
        h->add_spec(p + ".attn.proj.bias", {F});
        h->add_spec(p + ".scale_attn", {F});
        const int sh = h->gh_hidden, shp = h->gh_hidden_p;
        if (sh) {  // ViT-G: SwiGLU FFN (components/misc_helpers.py:162-168)
            h->add_spec(p + ".mlp.inner_linear_doubled.weight", {2 * sh, F});
            h->add_spec(p + ".mlp.inner_linear_doubled.bias", {2 * sh});
            h->add_spec(p + ".mlp.outer_linear.weight", {F, sh});
            h->add_spec(p + ".mlp.outer_linear.bias", {F});
        } else {
            h->add_spec(p + ".mlp.layers.0.weight", {4 * F, F});
            h->add_spec(p + ".mlp.layers.0.bias", {4 * F});
            h->add_spec(p + ".mlp.layers.2.weight", {F, 4 * F});
            h->add_spec(p + ".mlp.layers.2.bias", {F});
        }
        h->add_spec(p + ".scale_mlp", {F});
        h->add_mat(p + ".attn.qkv.weight", MDPT_PACK_LINEAR, 3 * F, F, 3 * F, F, 0);
        h->add_mat(p + ".attn.proj.weight", MDPT_PACK_LINEAR, F, F, F, F, 0);
        if (sh) {
            h->add_mat(p + ".mlp.inner_linear_doubled.weight", MDPT_PACK_LINEAR, 2 * sh, F, 2 * sh, F, 0);
            h->add_mat(p + ".mlp.outer_linear.weight", MDPT_PACK_LINEAR, F, sh, F, shp, 0);
            h->add_vec(p + ".mlp.inner_linear_doubled.bias", 2 * sh, 2 * sh);
        } else {
            h->add_mat(p + ".mlp.layers.0.weight", MDPT_PACK_LINEAR, 4 * F, F, 4 * F, F, 0);
            h->add_mat(p + ".mlp.layers.2.weight", MDPT_PACK_LINEAR, F, 4 * F, F, 4 * F, 0);
            h->add_vec(p + ".mlp.layers.0.bias", 4 * F, 4 * F);
        }
        // LayerScale (x + gamma * f(x), transformer_block.py:58,63) is folded into the producing Linear at pack time: rows of W and the
        // bias are multiplied by gamma, so the residual GEMMs compute out = (x + a W'^T) + b' with accumulators that START at x
        {
            const std::string fc2 = sh ? p + ".mlp.outer_linear" : p + ".mlp.layers.2";
            h->mats[h->mat_index.at(p + ".attn.proj.weight")].row_scale = p + ".scale_attn";
            h->mats[h->mat_index.at(fc2 + ".weight")].row_scale = p + ".scale_mlp";
#ifndef MDPT_NO_WSCALE  // (A/B builds)
            if (h->f16)  // fp16 operands: a power-of-two factor keeps gamma * W (and its lo plane) in fp16's normal range (GemmParams::wscale)
                for (const std::string& mn : {p + ".attn.proj.weight", fc2 + ".weight"}) {
                    Mat& mm = h->mats[h->mat_index.at(mn)];
                    mm.off_scale = h->packed_total;
                    h->packed_total += 256;
                }
#endif
            h->add_vec(p + ".attn.proj.bias@ls", F, F);
            h->vecs.back().scale = p + ".scale_attn";
            h->add_vec(fc2 + ".bias@ls", F, F);
            h->vecs.back().scale = p + ".scale_mlp";
        }
        if (beit) h->add_vec(p + ".attn.qkv.bias@qv", 0, 3 * F);  // assembled in finalize: [q_bias, 0, v_bias]
        else h->add_vec(p + ".attn.qkv.bias", 3 * F, 3 * F);
    }

    for (int i = 0; i < 4; ++i) {
        const std::string p = std::string("reassemble.") + kStageNames[i];
        const int hd = h->hid[i], hp = h->hidp[i];
        if (beit) {  // ReadoutProjectLayer: cat(token, cls) -> Linear(2F->F) -> GELU (components/readout_projection.py:42-46)
            h->add_spec(p + ".readout_proj.1.weight", {F, 2 * F});
            h->add_spec(p + ".readout_proj.1.bias", {F});
            h->add_mat(p + ".readout_proj.1.weight", MDPT_PACK_LINEAR, F, F, F, F, 0);          // token half (columns 0..F)
            h->add_mat(p + ".readout_proj.1.weight@cls", MDPT_PACK_LINEAR, F, F, F, F, 0);      // cls half (columns F..2F)
            h->add_vec(p + ".readout_proj.1.bias", F, F);
        }
        h->add_spec(p + ".resample.0.weight", {hd, F, 1, 1});
        h->add_spec(p + ".resample.0.bias", {hd});
        h->add_mat(p + ".resample.0.weight", MDPT_PACK_LINEAR, hd, F, hp, F, 0);
        h->add_vec(p + ".resample.0.bias", hd, hp);
        if (i == 0 || i == 1) {
            const int k = i == 0 ? 4 : 2;
            h->add_spec(p + ".resample.1.weight", {hd, hd, k, k});
            h->add_spec(p + ".resample.1.bias", {hd});
            h->add_mat(p + ".resample.1.weight", MDPT_PACK_CONVT, hd, hd, k * k * hp, hp, k);
            h->add_vec(p + ".resample.1.bias", hd, hp);
        } else if (i == 3) {
            h->add_spec(p + ".resample.1.weight", {hd, hd, 3, 3});
            h->add_spec(p + ".resample.1.bias", {hd});
            h->add_mat(p + ".resample.1.weight", MDPT_PACK_CONV3, hd, hd, hp, 9 * hp, 3);
            h->add_vec(p + ".resample.1.bias", hd, hp);
        }
        h->add_spec(p + ".fuse_proj.weight", {C, hd, 3, 3});
        h->add_mat(p + ".fuse_proj.weight", MDPT_PACK_CONV3, C, hd, h->Cp, 9 * hp, 3);
    }
    return build_inventory_decoder(h);
}

// fusion + head parameters (same structure in every family; attribute names differ, see rcu_seq / proj_seq)
int build_inventory_decoder(mdpt_handle* h) {
    const int C = h->C;
    for (int b = 0; b < 4; ++b) {
        char pb[64];
        snprintf(pb, sizeof(pb), "fusion.blocks.%d", b);
        std::vector<std::string> units;
        if (b < 3) units.push_back(std::string(pb) + ".conv_reassembly");
        units.push_back(std::string(pb) + "." + proj_seq(h) + ".0");
        for (const std::string& u : units)
            for (const char* idx : {"1", "3"}) {
                const std::string n = u + "." + rcu_seq(h) + "." + idx;
                h->add_spec(n + ".weight", {C, C, 3, 3});
                h->add_spec(n + ".bias", {C});
                h->add_mat(n + ".weight", MDPT_PACK_CONV3, C, C, h->Cp, 9 * h->Cp, 3);
                h->add_vec(n + ".bias", C, h->Cp);
            }
        const std::string o = std::string(pb) + "." + proj_seq(h) + ".2";
        h->add_spec(o + ".weight", {C, C, 1, 1});
        h->add_spec(o + ".bias", {C});
        h->add_mat(o + ".weight", MDPT_PACK_LINEAR, C, C, h->Cp, h->Cp, 0);
        h->add_vec(o + ".bias", C, h->Cp);
    }

    h->add_spec("head.spatial_upsampler.0.weight", {h->C2, C, 3, 3});
    h->add_spec("head.spatial_upsampler.0.bias", {h->C2});
    h->add_spec("head.proj_1ch.0.weight", {32, h->C2, 3, 3});
    h->add_spec("head.proj_1ch.0.bias", {32});
    h->add_spec("head.proj_1ch.2.weight", {1, 32, 1, 1});
    h->add_spec("head.proj_1ch.2.bias", {1});
    h->add_mat("head.spatial_upsampler.0.weight", MDPT_PACK_CONV3, h->C2, C, h->C2p, 9 * h->Cp, 3);
    h->add_vec("head.spatial_upsampler.0.bias", h->C2, h->C2p);
    h->add_mat("head.proj_1ch.0.weight", MDPT_PACK_CONV3, 32, h->C2, 32, 9 * h->C2p, 3);
    if (head_tail_fused(h))  // LDS image of the same weights for the fused head tail (head.hip)
        h->add_mat("head.proj_1ch.0.weight@kc32", MDPT_PACK_CONV3_KC32, 32, h->C2, 32, 9 * h->C2p, 3);
    h->add_vec("head.proj_1ch.0.bias", 32, 32);
    h->add_vec("head.proj_1ch.2.weight", 32, 32);
    h->add_vec("head.proj_1ch.2.bias", 1, 4);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// workspace planning
// ------------------------------------------------------------------------------------------------------------
// f8mode (fm() of the CONSUMING class): 0 = 16-bit residue plane, 1 = e5m2 residue bytes, 2 = + the e5m2 plane of the values (three terms);
// the buffer is the same 2 bytes per element either way (Planes, f8_cross.h)
void take_planes(Bump& bump, bool x3, size_t elems, size_t out[3], int f8mode = 0) {
    out[0] = bump.take(elems * 2);
    out[1] = x3 ? bump.take(elems * 2) : SIZE_MAX;
    out[2] = x3 && f8mode ? (elems | (f8mode == 2 ? (size_t)1 << 63 : 0)) : 0;
}
static int fm(const mdpt_handle* h, int cls) { return h->f8(cls) ? (h->terms(cls) == 3 ? 2 : 1) : 0; }

// reassembly outputs, fusion and head buffers; p.Np / p.gh / p.gw = the "noscale" level (1/Pv of the image)
void plan_decoder(Bump& bump, const mdpt_handle* h, Plan& p, size_t min_scratch_floats) {
    // lo planes exist where the CONSUMING class reads one (2 or 3 passes): the reassembly maps of levels 0..2 and a1 feed the conv_reassembly
    // units (CLS_FUSION_IN), level 3's map, x and b1 the projection path's 3x3 convs (CLS_FUSION), b2 the 1x1 projection (CLS_FUSION_PROJ)
    const bool x3 = h->alo(CLS_FUSION), x3i = h->alo(CLS_FUSION_IN), x3p = h->alo(CLS_FUSION_PROJ), x3h = h->alo(CLS_HEAD);
    const int B = p.B;
    const size_t px[4] = {(size_t)16 * p.Np, (size_t)4 * p.Np, (size_t)p.Np, (size_t)p.Np / 4};
    for (int i = 0; i < 4; ++i) {
        const size_t e = (size_t)B * px[i] * h->Cp;
        p.r_f32[i] = bump.take(e * 4);
        take_planes(bump, i == 3 ? x3 : x3i, e, p.r_bf[i], fm(h, i == 3 ? CLS_FUSION : CLS_FUSION_IN));
        take_planes(bump, x3i, e, p.a1[i], fm(h, CLS_FUSION_IN));
        p.x_f32[i] = bump.take(e * 4);
        take_planes(bump, x3, e, p.x_bf[i], fm(h, CLS_FUSION));
        take_planes(bump, x3, e, p.b1[i], fm(h, CLS_FUSION));
        take_planes(bump, x3p, e, p.b2[i], fm(h, CLS_FUSION_PROJ));
        p.flo[i] = bump.take(e * 4);
    }
    const size_t fpx = (size_t)64 * p.Np;  // (8gh)*(8gw)
    take_planes(bump, x3h, (size_t)B * fpx * h->Cp, p.fused, fm(h, CLS_HEAD));
    // bf16 mode with the fused head tail (run_head): conv 1 writes a bf16 map and the full-resolution upsampled map never exists (ViT-L,
    // 504x504, batch 32: 2.1 GB + 0.7 GB of workspace that used to be reserved and never touched)
    const bool bf16_head = head_tail_fused(h) && mdpt_head_tail_scale_ok(8 * p.gh, 8 * p.gw, p.H, p.W);
    p.h1 = bump.take((size_t)B * fpx * h->C2p * (bf16_head && h->terms(CLS_HEAD_TAIL) == 1 ? 2 : 4));  // one 16-bit plane | hi + lo planes | fp32 map
    if (bf16_head) p.h1u[0] = p.h1u[1] = SIZE_MAX;
    else take_planes(bump, h->alo(CLS_HEAD_TAIL), (size_t)B * p.H * p.W * h->C2p, p.h1u);
    p.scratch_floats = (size_t)B * fpx * h->Cp;
    if (min_scratch_floats > p.scratch_floats) p.scratch_floats = min_scratch_floats;
    p.scratch = bump.take(p.scratch_floats * 4);
    p.probe = bump.take(256);  // the side-stream probe's two flag words: nothing else ever lives here (mdpt_api.cpp ensure_side_stream)
    p.poison = bump.take((size_t)p.B * 4);  // per-image non-finite flags (patchify_kernel sets, poison_depth_kernel reads: mdpt_api.cpp forward_one)
}

int make_plan_swin(const mdpt_handle* h, int B, int H, int W, Plan* pl);

int make_plan(const mdpt_handle* h, int B, int H, int W, Plan* pl) {
    if (B <= 0 || H <= 0 || W <= 0) return fail(MDPT_E_INVALID, "bad batch/size B=%d H=%d W=%d", B, H, W);
    if (h->swin) return make_plan_swin(h, B, H, W, pl);
    if (H % h->P || W % h->P)
        return fail(MDPT_E_INVALID, "image size %dx%d must be divisible by the patch size %d (reference patch_embed.py:159-163)", H, W, h->P);
    const int gh = H / h->P, gw = W / h->P;
    if ((gh & 1) || (gw & 1))
        return fail(MDPT_E_GRID, "patch grid %dx%d must be even in both dimensions (the reference fails in fusion_model.py:151)", gh, gw);
    const int F = h->F;
    Plan& p = *pl;
    p.B = B; p.H = H; p.W = W; p.gh = gh; p.gw = gw;
    p.Np = gh * gw; p.N = p.Np + 1; p.npad = rup(p.N, 8); p.npadv = rup(p.N, 64);
    Bump bump;
    const size_t rows = (size_t)B * p.npad;
    take_planes(bump, h->alo(CLS_PATCH), (size_t)B * p.Np * h->Kpatch, p.im2col);
    p.pos = bump.take((size_t)p.Np * F * 4);
    p.resid = bump.take(rows * F * 4);
    take_planes(bump, h->alo(CLS_QKV) || h->alo(CLS_FC1), rows * F, p.xn);
    take_planes(bump, h->x3c(CLS_ATTN), (size_t)B * h->heads * p.npad * 64, p.q);
    take_planes(bump, h->x3c(CLS_ATTN), (size_t)B * h->heads * p.npad * 64, p.k);
    take_planes(bump, h->x3c(CLS_ATTN), (size_t)B * h->heads * 64 * p.npadv, p.vt);
    take_planes(bump, h->alo(CLS_PROJ) || h->x3c(CLS_ATTN), rows * F, p.att);  // the 3-pass attention kernel always writes its lo plane
    take_planes(bump, h->alo(CLS_FC2), rows * 4 * F, p.hbuf);
    p.swi = h->gh_hidden ? bump.take(rows * 2 * h->gh_hidden * 4) : SIZE_MAX;
    p.kspart = fc2_ksplit_fits((int)rows, F) ? bump.take(rows * F * 4 * 3) : SIZE_MAX;  // three partial-sum planes (a split in four); reserved whatever the latency switch says: it may flip later
    p.wrc_mean = h->wrc_maxk ? bump.take((size_t)B * h->wrc_maxk * 2) : SIZE_MAX;
    p.wrc_tab = h->wrc_maxn ? bump.take((size_t)B * h->wrc_maxn * 4) : SIZE_MAX;
    const bool x3 = h->alo(CLS_REASM);
    const int fr = fm(h, CLS_REASM);
    for (int i = 0; i < 4; ++i) take_planes(bump, x3, rows * F, p.tap[i], fr);
    p.tapf32 = bump.take(rows * F * 4);
    const size_t px[4] = {(size_t)16 * p.Np, (size_t)4 * p.Np, (size_t)p.Np, (size_t)p.Np / 4};
    for (int i = 0; i < 4; ++i) take_planes(bump, x3, (size_t)B * p.Np * h->hidp[i], p.t[i], fr);
    take_planes(bump, x3, (size_t)B * px[0] * h->hidp[0], p.u0, fr);
    take_planes(bump, x3, (size_t)B * px[1] * h->hidp[1], p.u1, fr);
    take_planes(bump, x3, (size_t)B * px[3] * h->hidp[3], p.d3, fr);
    plan_decoder(bump, h, p, rows * F);
    p.tokr[0] = p.tokr[1] = p.cbuf = p.relpos_lut = p.relpos_tq = p.relpos_tk = SIZE_MAX;
    if (is_beit(h)) {
        take_planes(bump, x3, (size_t)B * p.Np * F, p.tokr, fr);
        p.cbuf = bump.take((size_t)B * F * 4);
        p.relpos_lut = bump.take((size_t)h->heads * mdpt_beit_relpos_elen(gh, gw) * 4 * (h->nblocks <= 32 ? h->nblocks : 1));  // one table per block
        p.relpos_tq = bump.take((size_t)p.npadv * 4);
        p.relpos_tk = bump.take((size_t)p.npadv * 4);
    }
    p.total = bump.off;
    return 0;
}

#include "mdpt_swin_plan.inc"

int check_ws(const mdpt_handle* h, const Plan& p, const void* ws, size_t bytes) {
    if (!h->finalized) return fail(MDPT_E_STATE, "mdpt_finalize() has not been called");
    if (!ws || bytes < p.total) return fail(MDPT_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", p.total, bytes);
    if (((uintptr_t)ws) & 255) return fail(MDPT_E_WORKSPACE, "workspace must be 256-byte aligned");
    return 0;
}

int make_ctx(mdpt_handle* h, int B, int H, int W, void* ws, size_t ws_bytes, void* stream, Ctx* c) {
    Plan p;
    CHK(make_plan(h, B, H, W, &p));
    CHK(check_ws(h, p, ws, ws_bytes));
    c->h = h; c->p = p; c->ws = (char*)ws; c->s = (hipStream_t)stream;
    h->cache_clear();  // whoever builds a context may write the constant regions for another grid (stage-level calls): mdpt_forward re-validates its own
    return 0;
}

}  // namespace mdpt

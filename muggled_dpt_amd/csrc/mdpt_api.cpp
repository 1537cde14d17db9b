// libmdpt: the C ABI of include/mdpt.h (entry points only; inventory / plan: mdpt_inventory.cpp, stage drivers: mdpt_stages.cpp,
// test hooks: mdpt_debug.cpp).
#include "mdpt_internal.h"

// =====================================================================================================================
// C ABI
// =====================================================================================================================
extern "C" {

int mdpt_abi_version(void) { return MDPT_ABI_VERSION; }
const char* mdpt_last_error(void) { return g_err.c_str(); }

static constexpr int WRC_ALL = (1 << CLS_QKV) | (1 << CLS_PROJ) | (1 << CLS_FC1) | (1 << CLS_FC2);

int mdpt_create(const mdpt_config* cfg, mdpt_handle** out) {
    if (!cfg || !out) return fail(MDPT_E_INVALID, "null argument");
    *out = nullptr;
    if (cfg->is_giant && cfg->family != MDPT_FAMILY_DAV2) return fail(MDPT_E_INVALID, "is_giant (SwiGLU MLP) exists for Depth-Anything V2 only");
    if (cfg->family < MDPT_FAMILY_DAV2 || cfg->family > MDPT_FAMILY_SWINV2) return fail(MDPT_E_INVALID, "unknown model family %d", cfg->family);
    const bool swin = cfg->family == MDPT_FAMILY_SWINV2;
    if (cfg->features_per_token <= 0 || cfg->features_per_token % (swin ? 32 : 64))
        return fail(MDPT_E_INVALID, "features_per_token must be a positive multiple of %d, got %d", swin ? 32 : 64, cfg->features_per_token);
    if (swin) {
        if (cfg->patch_size_px != 4) return fail(MDPT_E_UNSUPPORTED, "SwinV2 DPT needs patch_size_px = 4 (head upsample x2 of a 1/2-resolution map), got %d", cfg->patch_size_px);
        if (cfg->swin_window_h <= 0 || cfg->swin_window_w <= 0) return fail(MDPT_E_INVALID, "bad SwinV2 window size");
        if (cfg->features_per_token != cfg->reassembly_features[0]) return fail(MDPT_E_INVALID, "SwinV2: features_per_token must equal features_per_stage[0]");
        for (int i = 0; i < 4; ++i) {
            // swin2_tiny_256 has a 96-wide first stage (3 heads of 32): operand planes are padded to a multiple of 64 columns
            if (cfg->reassembly_features[i] <= 0 || cfg->reassembly_features[i] % 32)
                return fail(MDPT_E_INVALID, "SwinV2 features_per_stage[%d] must be a positive multiple of 32, got %d", i, cfg->reassembly_features[i]);
            if (cfg->swin_heads[i] * 32 != cfg->reassembly_features[i])
                return fail(MDPT_E_UNSUPPORTED, "SwinV2 head dim must be 32 (stage %d: heads=%d, features=%d)", i, cfg->swin_heads[i], cfg->reassembly_features[i]);
            if (cfg->swin_layers[i] <= 0 || cfg->swin_layers[i] % 2) return fail(MDPT_E_INVALID, "SwinV2 layers_per_stage[%d] must be a positive even number", i);
            if (cfg->swin_pretrained_window[i] < 0) return fail(MDPT_E_INVALID, "bad SwinV2 pretrained window size");
        }
    } else {
        if (cfg->num_heads * 64 != cfg->features_per_token)
            return fail(MDPT_E_UNSUPPORTED, "head dim must be 64 (heads=%d, features=%d)", cfg->num_heads, cfg->features_per_token);
        if (cfg->num_blocks <= 0 || cfg->num_blocks % 4) return fail(MDPT_E_INVALID, "num_blocks must be a positive multiple of 4, got %d", cfg->num_blocks);
    }
    if (cfg->fusion_channels <= 0 || cfg->fusion_channels % 8) return fail(MDPT_E_INVALID, "fusion_channels must be a multiple of 8");
    if (cfg->patch_size_px <= 0 || cfg->patch_size_px % 2) return fail(MDPT_E_INVALID, "patch_size_px must be even (head scale = patch/8)");
    if (cfg->base_patch_grid_h <= 0 || cfg->base_patch_grid_w <= 0) return fail(MDPT_E_INVALID, "bad base patch grid");
    if (cfg->precision < MDPT_PREC_BF16 || cfg->precision > MDPT_PREC_MIXED) return fail(MDPT_E_INVALID, "unknown precision %d", cfg->precision);
    for (int i = 0; i < 4; ++i)
        if (cfg->reassembly_features[i] <= 0 || cfg->reassembly_features[i] % 4) return fail(MDPT_E_INVALID, "reassembly_features[%d] must be a multiple of 4", i);
    mdpt_handle* h = new mdpt_handle();
    h->cfg = *cfg;
    h->F = cfg->features_per_token; h->heads = cfg->num_heads; h->nblocks = cfg->num_blocks; h->bps = cfg->num_blocks / 4;
    h->P = cfg->patch_size_px; h->C = cfg->fusion_channels; h->Cp = rup(h->C, 64);
    h->C2 = h->C / 2; h->C2p = rup(h->C2, 64);
    h->Kpatch = rup(3 * h->P * h->P, 64);
    for (int i = 0; i < 4; ++i) { h->hid[i] = cfg->reassembly_features[i]; h->hidp[i] = rup(h->hid[i], 64); }
    h->swin = swin;
    h->Pv = swin ? 16 : h->P;
    // components/misc_helpers.py:164-165: 2/3 of the 4x MLP width, rounded up to a multiple of 8
    h->gh_hidden = cfg->is_giant ? 8 * (((int)((long)(4 * h->F) * 2 / 3) + 7) / 8) : 0;
    h->gh_hidden_p = rup(h->gh_hidden, 64);
    for (int i = 0; i < 4; ++i) { h->sH[i] = cfg->swin_heads[i]; h->sL[i] = cfg->swin_layers[i]; h->spre[i] = cfg->swin_pretrained_window[i]; }
    h->swh = cfg->swin_window_h; h->sww = cfg->swin_window_w;
    h->f16 = cfg->precision == MDPT_PREC_FP16 || cfg->precision == MDPT_PREC_FP16X3 || cfg->precision == MDPT_PREC_MIXED;
    {
        // MDPT_PREC_MIXED: the table of mdpt_default_mixed_passes_for where the configuration can run the fp8 forms it names, the 16-bit-plane table of
        // round 5 for a class that cannot (the small encoders' reassembly widths, fusion widths below 128, SwinV2): there "3 terms on fp8" would
        // silently become three fp16 passes where two were the measured choice
        int32_t mixed[NCLS], mixed16[NCLS];
        mdpt_default_mixed_passes_for(cfg->family, mixed);
        mdpt_default_mixed_passes_r05(cfg->family, mixed16);
        compute_f8ok(h);
        const bool all3 = cfg->precision == MDPT_PREC_BF16X3 || cfg->precision == MDPT_PREC_FP16X3;
        for (int i = 0; i < NCLS; ++i) h->np[i] = cfg->precision == MDPT_PREC_MIXED ? ((mixed[i] >= MDPT_PASSES_2F8 && !h->f8ok[i]) ? mixed16[i] : mixed[i]) : (all3 ? 3 : 1);
    }
    // token-mean compensation: on where the encoder's single-pass Linears are what is left of the error (mixed). With a single-pass decoder behind
    // them (MDPT_PREC_FP16) the map does not get better - encoder taps -25 %, map rms +10 ... +37 % on BEiT-L / SwinV2-L, -15 % on ViT-L,
    // profiles/r04_wrc_by_family.txt - for 3.5 % of the step: off there unless asked for
    h->wrc_on = cfg->precision == MDPT_PREC_MIXED;
    h->wrc_mask = WRC_ALL;
    h->gemm_tile = MDPT_TILE_AUTO;
    h->finalized = false;
    h->has_last = false;
    h->zero_page = nullptr;
    h->dbg_block = h->dbg_step = -1;
    h->ks_min_ktiles = 16; h->ks_big_ktiles = 1 << 30;
    h->split_min = 8;
    h->latency_mode = 0;
    h->overlap_reasm = 1;
    h->side_prio = 0;
    h->side_probe = 1;
    h->side_ncand = h->side_rejected = h->side_unresolved = 0;
    h->side_nfor = 0;
    h->grid_cache = 0;
    h->gen = 0;
    h->cache_clear();
    h->side_stream = nullptr;
    h->ev_fork = h->ev_join = nullptr;
    build_inventory(h);
    *out = h;
    return 0;
}

void mdpt_destroy(mdpt_handle* h) { delete h; }

// MDPT_PREC_MIXED: which classes pay for more than one pass. From the per-class error budget (profiles/r05_precision_budget.md; the CPU
// emulation tests/precision_budget/emulate_operand_rounding.py reproduces it): the decoder's convs feed the depth map directly - no
// LayerNorm or residual stream between them and the output averages their operand rounding away. Round 5 split the rounding of every
// decoder layer group by OPERAND: in the fusion blocks' 3x3 convs and in both head convs it is the ACTIVATIONS' rounding that reaches the map
// (weights rounded once cost 3e-5 rms each on top of 7.4e-5), so they run the two-pass activation-split form; the reassembly convs and the
// 1x1 fusion projections need their weights split too (three passes).
// The two-pass rows hold for the Depth-Anything families (ViT-L 504^2: worst checked image 9.1e-4, rms 9.1e-5 against 7.2e-5 with three passes).
// The MiDaS v3.1 families keep three passes for the whole projection path: on their reference fixtures the activation-split form reads
// 1.07e-3 (BEiT-L) / 1.08e-3 with twice the rms (SwinV2-L) against 8.4e-4 / 6.8e-4 (tools/probes/gpu_family_class_budget.py,
// profiles/r05_family_class_budget.txt) - their decoders' weight rounding is not the smaller half of the budget.
// Round 6: the cross terms of the decoder classes on fp8 planes (MDPT_PASSES_2F8 / _3F8, csrc/f8_cross.h) - 1.5 / 2 pass-equivalents where the table
// above paid 2 / 3. Measured on ViT-L 504^2 batch 32, one box, images 0 / 7 / 13 / 31 (tests/precision_budget/measure_on_gpu.py, profiles/r06_precision_budget_f8.md):
//   round 5's table                                   worst 9.05e-4  rms 9.12e-5  52.40 ms
//   the same table with fp8 cross terms               worst 9.26e-4  rms 9.15e-5  49.73 ms  (the accuracy of the fp16 cross terms, as the emulation said)
//   + fusion at THREE terms (its weights split too)   worst 7.97e-4  rms 8.38e-5  50.64 ms  (8.61e-4 after an equally valid re-rounding of the compensation's sums)
//   + head at three terms                             worst 7.60e-4  rms 7.96e-5  51.46 ms  <- shipped: most of the gain is spent on margin
// A configuration that cannot run the fp8 forms (mdpt_get_class_f8) keeps round 5's table (mdpt_create).
void mdpt_default_mixed_passes_for(int32_t family, int32_t passes[MDPT_NUM_CLASSES]) {
    const bool midas = family == MDPT_FAMILY_BEIT || family == MDPT_FAMILY_SWINV2;
    for (int i = 0; i < NCLS; ++i) passes[i] = 1;
    passes[CLS_PATCH] = 3;  // 0.13 % of the FLOPs
    passes[CLS_REASM] = MDPT_PASSES_3F8;
    passes[CLS_FUSION] = MDPT_PASSES_3F8;
    passes[CLS_FUSION_PROJ] = MDPT_PASSES_3F8;
    passes[CLS_HEAD] = MDPT_PASSES_3F8;  // (Depth-Anything ran it at 2F8 for a while: 7.97e-4 / 8.61e-4 worst image under two equally valid re-roundings; three terms 7.6e-4)
    passes[CLS_HEAD_TAIL] = midas ? 3 : 2;  // (its own kernel, VALU-bound in the halo interpolation: fp16 planes)
    passes[CLS_FUSION_IN] = 1;  // 2 % of the decoder's squared error for a quarter of its FLOPs (profiles/r04_precision_budget.md)
    if (family == MDPT_FAMILY_BEIT) {
        // BEiT-L's reference fixture sits closest to the bound, and which side of it depends on how the fp16 weight scale was rounded
        // (mdpt_debug_set_wscale_policy: 7.8e-4 under the shipped rule, 1.11e-3 with every folded matrix scaled) - and on any other re-rounding: with
        // ONE of the two MLP Linears at three passes the same fixture read 5.95e-4 / 7.51e-4 before and 1.016e-3 / 6.18e-4 after a change of the
        // attention kernel's summation order, the RMS of the error moving between 1.2e-4 and 2.9e-4 (a coherent error component of random
        // amplitude that the single-pass MLP carries). Both MLP Linears at three passes: 4.6e-4 ... 5.7e-4 in four such samples, rms 0.86e-4 ... 1.4e-4,
        // for 4 ms of 16.7 at batch 16 (profiles/r06_beitl_class_budget.txt). SwinV2-L reads 6.8e-4 ... 6.9e-4 under every rounding tried and keeps its table.
        passes[CLS_FC1] = 3;
        passes[CLS_FC2] = 3;
        passes[CLS_FUSION_IN] = MDPT_PASSES_2F8;
    }
}

// round 5's table (16-bit planes only): what a configuration without the fp8 forms runs
void mdpt_default_mixed_passes_r05(int32_t family, int32_t passes[MDPT_NUM_CLASSES]) {
    const bool midas = family == MDPT_FAMILY_BEIT || family == MDPT_FAMILY_SWINV2;
    for (int i = 0; i < NCLS; ++i) passes[i] = 1;
    passes[CLS_PATCH] = 3;
    passes[CLS_REASM] = 3;
    passes[CLS_FUSION] = midas ? 3 : 2;
    passes[CLS_FUSION_PROJ] = 3;
    passes[CLS_HEAD] = midas ? 3 : 2;
    passes[CLS_HEAD_TAIL] = midas ? 3 : 2;
    passes[CLS_FUSION_IN] = 1;
}

void mdpt_default_mixed_passes(int32_t passes[MDPT_NUM_CLASSES]) { mdpt_default_mixed_passes_for(MDPT_FAMILY_DAV2, passes); }

int mdpt_get_class_passes(const mdpt_handle* h, int32_t op_class, int32_t* passes) {
    if (!h || !passes || op_class < 0 || op_class >= NCLS) return fail(MDPT_E_INVALID, "bad argument");
    *passes = h->np[op_class];
    return 0;
}

int mdpt_get_class_f8(const mdpt_handle* h, int32_t op_class, int32_t* on) {
    if (!h || !on || op_class < 0 || op_class >= NCLS) return fail(MDPT_E_INVALID, "bad argument");
    *on = h->f8(op_class) ? 1 : 0;
    return 0;
}

static void rebuild_inventory_keeping_bindings(mdpt_handle* h);

int mdpt_set_class_passes(mdpt_handle* h, int32_t op_class, int32_t passes) {
    if (!h || op_class < 0 || op_class >= NCLS) return fail(MDPT_E_INVALID, "bad op class %d", op_class);
    if (passes < 1 || passes > MDPT_PASSES_3F8) return fail(MDPT_E_INVALID, "passes must be 1, 2 (activations split), 3, MDPT_PASSES_2F8 (4) or MDPT_PASSES_3F8 (5), got %d", passes);
    if (passes == 2 && op_class == CLS_ATTN) return fail(MDPT_E_INVALID, "the attention kernel's operands are both activations: 1 or 3 passes");
    if (passes >= MDPT_PASSES_2F8 && op_class != CLS_REASM && op_class != CLS_FUSION && op_class != CLS_FUSION_IN && op_class != CLS_FUSION_PROJ && op_class != CLS_HEAD)
        return fail(MDPT_E_INVALID, "fp8 cross terms (MDPT_PASSES_2F8 / _3F8) exist for the reasm, fusion, fusion_in, fusion_proj and head classes, not for class %d", op_class);
    if (h->np[op_class] == passes) return 0;
    h->np[op_class] = passes;
    rebuild_inventory_keeping_bindings(h);  // the packed-weight inventory depends on the pass counts (lo planes)
    return 0;
}

static void rebuild_inventory_keeping_bindings(mdpt_handle* h) {
    std::vector<WeightSpec> bound = h->specs;
    h->specs.clear(); h->spec_index.clear(); h->mats.clear(); h->mat_index.clear(); h->vecs.clear(); h->vec_index.clear();
    build_inventory(h);
    for (const WeightSpec& b : bound) {
        auto it = h->spec_index.find(b.name);
        if (it != h->spec_index.end()) { h->specs[it->second].ptr = b.ptr; h->specs[it->second].dtype = b.dtype; }
    }
    h->finalized = false;
    h->has_last = false;
}

int mdpt_set_weight_rounding_compensation(mdpt_handle* h, int32_t on) {
    if (!h) return fail(MDPT_E_INVALID, "null handle");
    if (on && !h->f16) return fail(MDPT_E_UNSUPPORTED, "the token-mean compensation exists for the fp16 operand modes (MDPT_PREC_FP16 / _MIXED / _FP16X3 with single-pass classes)");
    if (on < 0 || (on > 1 && (on & ~WRC_ALL))) return fail(MDPT_E_INVALID, "0, 1 or a mask of (1 << MDPT_CLASS_QKV | _PROJ | _FC1 | _FC2), got %d", on);
    const int mask = on > 1 ? on : WRC_ALL;
    if (h->wrc_on == (on != 0) && (!on || h->wrc_mask == mask)) return 0;
    h->wrc_on = on != 0;
    if (on) h->wrc_mask = mask;
    rebuild_inventory_keeping_bindings(h);
    return 0;
}

int mdpt_debug_set_operand_format(int32_t fp16) {
    g_debug_f16 = fp16 ? 1 : 0;
    return 0;
}

int mdpt_num_weights(const mdpt_handle* h) { return h ? (int)h->specs.size() : 0; }

const char* mdpt_weight_name(const mdpt_handle* h, int index) {
    if (!h || index < 0 || index >= (int)h->specs.size()) return nullptr;
    return h->specs[index].name.c_str();
}

int mdpt_weight_shape(const mdpt_handle* h, int index, int32_t* ndim, int64_t shape[4]) {
    if (!h || index < 0 || index >= (int)h->specs.size() || !ndim || !shape) return fail(MDPT_E_INVALID, "bad weight index");
    *ndim = h->specs[index].ndim;
    for (int i = 0; i < 4; ++i) shape[i] = h->specs[index].shape[i];
    return 0;
}

int mdpt_bind_weight(mdpt_handle* h, const char* name, const void* dev_ptr, int32_t dtype, int32_t ndim, const int64_t* shape) {
    if (!h || !name || !dev_ptr || !shape) return fail(MDPT_E_INVALID, "null argument");
    if (dtype != MDPT_DTYPE_F32 && dtype != MDPT_DTYPE_BF16 && dtype != MDPT_DTYPE_F16) return fail(MDPT_E_INVALID, "bad dtype %d for \"%s\"", dtype, name);
    auto it = h->spec_index.find(name);
    if (it == h->spec_index.end()) return fail(MDPT_E_INVALID, "unexpected parameter \"%s\" (not part of this model config)", name);
    WeightSpec& s = h->specs[it->second];
    bool ok = ndim == s.ndim;
    for (int i = 0; ok && i < ndim; ++i) ok = shape[i] == s.shape[i];
    if (!ok) {
        std::string got, want;
        for (int i = 0; i < ndim; ++i) got += (i ? "x" : "") + std::to_string(shape[i]);
        for (int i = 0; i < s.ndim; ++i) want += (i ? "x" : "") + std::to_string(s.shape[i]);
        return fail(MDPT_E_SHAPE, "size mismatch for %s: got %s, model expects %s", name, got.c_str(), want.c_str());
    }
    s.ptr = dev_ptr;
    s.dtype = dtype;
    h->finalized = false;
    return 0;
}

int mdpt_packed_bytes(const mdpt_handle* h, size_t* bytes) {
    if (!h || !bytes) return fail(MDPT_E_INVALID, "null argument");
    *bytes = h->packed_total;
    return 0;
}

int mdpt_finalize(mdpt_handle* h, void* packed_dev, size_t bytes, void* stream) {
    if (!h || !packed_dev) return fail(MDPT_E_INVALID, "null argument");
    if (bytes < h->packed_total) return fail(MDPT_E_WORKSPACE, "packed buffer too small: need %zu bytes, got %zu", h->packed_total, bytes);
    if (((uintptr_t)packed_dev) & 255) return fail(MDPT_E_WORKSPACE, "packed buffer must be 256-byte aligned");
    for (const WeightSpec& s : h->specs)
        if (!s.ptr) return fail(MDPT_E_MISSING, "missing parameter \"%s\" (strict load)", s.name.c_str());
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)packed_dev;
    CHK(hipMemsetAsync(base + h->zero_off, 0, 256, st));
    h->zero_page = (op_t*)(base + h->zero_off);
    for (Mat& m : h->mats) {
        m.hi = (op_t*)(base + m.off_hi);
        m.lo = m.off_lo == SIZE_MAX ? nullptr : (op_t*)(base + m.off_lo);
        std::string src_name = m.src;
        int src_ld = 0, src_col0 = 0;
        const size_t kc = src_name.find("@kc32");
        if (kc != std::string::npos) src_name = src_name.substr(0, kc);
        const size_t at = src_name.find("@cls");
        if (src_name.find(".readout_proj.1.weight") != std::string::npos) {  // [F, 2F] split into token / cls halves
            src_ld = 2 * h->F;
            if (at != std::string::npos) { src_col0 = h->F; src_name = src_name.substr(0, at); }
        }
        const WeightSpec& sp = h->specs[h->spec_index.at(src_name)];
        const WeightSpec* rs = m.row_scale.empty() ? nullptr : &h->specs[h->spec_index.at(m.row_scale)];
        m.wscale = nullptr;
        if (m.off_scale != SIZE_MAX) {
            float* sc = (float*)(base + m.off_scale);
            CHK(OPLH(mdpt_launch_weight_scale, sp.ptr, sp.dtype, m.N, m.K, src_ld, src_col0, rs ? rs->ptr : nullptr, rs ? rs->dtype : 0, sc, st, h->wscale_all));
            m.wscale = sc;
        }
        CHK(OPLH(mdpt_launch_pack_weight, sp.ptr, sp.dtype, m.hi, m.lo, m.kind, m.N, m.K, m.Np, m.Kp, m.ksz, st, src_ld, src_col0, rs ? rs->ptr : nullptr,
                                    rs ? rs->dtype : 0, m.wscale));
        m.w8 = m.wlo8 = m.s8 = m.slo8 = nullptr;
        if (m.off_w8 != SIZE_MAX) {  // the fp8 planes of an F8 class (f8_cross.h): e4m3 of W_hi (and of W - W_hi), per-row E8M0 scales
            unsigned char* w8 = (unsigned char*)(base + m.off_w8);
            unsigned char* wlo8 = m.off_wlo8 == SIZE_MAX ? nullptr : (unsigned char*)(base + m.off_wlo8);
            unsigned char* s8 = (unsigned char*)(base + m.off_s8);
            CHK(mdpt_launch_pack_weight_f8_f16(sp.ptr, sp.dtype, w8, wlo8, s8, m.kind, m.N, m.K, m.Np, m.Kp, m.ksz, st, src_ld, src_col0, rs ? rs->ptr : nullptr,
                                               rs ? rs->dtype : 0, m.wscale));
            m.w8 = w8; m.wlo8 = wlo8; m.s8 = s8; m.slo8 = s8 + m.Np;
        }
    }
    for (Vec& v : h->vecs) {
        v.ptr = (float*)(base + v.off);
        const size_t at = v.src.find(".attn.qkv.bias@qv");
        if (at != std::string::npos) {  // [q_bias (heads*d = F), zeros(F), v_bias (F)]: the k projection has no bias
            const std::string blk = v.src.substr(0, at);
            const int Fq = v.np / 3;
            CHK(hipMemsetAsync(v.ptr, 0, (size_t)v.np * 4, st));
            const WeightSpec& qb = h->specs[h->spec_index.at(blk + ".attn.q_bias")];
            const WeightSpec& vb = h->specs[h->spec_index.at(blk + ".attn.v_bias")];
            CHK(OPLH(mdpt_launch_pad_copy_f32, qb.ptr, qb.dtype, v.ptr, Fq, Fq, st));
            CHK(OPLH(mdpt_launch_pad_copy_f32, vb.ptr, vb.dtype, v.ptr + 2 * Fq, Fq, Fq, st));
            continue;
        }
        const size_t ls = v.src.find("@ls");
        if (ls != std::string::npos) {  // bias * layer scale (see build_inventory)
            const WeightSpec& bs = h->specs[h->spec_index.at(v.src.substr(0, ls))];
            const WeightSpec& sc = h->specs[h->spec_index.at(v.scale)];
            CHK(OPLH(mdpt_launch_pad_copy_f32, bs.ptr, bs.dtype, v.ptr, v.n, v.np, st, sc.ptr, sc.dtype));
            continue;
        }
        const WeightSpec& vs = h->specs[h->spec_index.at(v.src)];
        // SwinV2's per-head logit scale is packed times log2(e): the window attention's scores are in log2 units (attention.hip LOG2)
        const bool ls2 = v.src.size() > 17 && v.src.compare(v.src.size() - 17, 17, ".attn.logit_scale") == 0;
        CHK(OPLH(mdpt_launch_pad_copy_f32, vs.ptr, vs.dtype, v.ptr, v.n, v.np, st, nullptr, 0, ls2 ? 1.4426950408889634f : 1.0f));
    }
    h->finalized = true;
    h->has_last = false;
    ++h->gen;  // cached per-grid constants were derived from the previous weights
    return 0;
}

int mdpt_workspace_bytes(const mdpt_handle* h, int32_t B, int32_t H, int32_t W, size_t* bytes) {
    if (!h || !bytes) return fail(MDPT_E_INVALID, "null argument");
    Plan p;
    CHK(make_plan(h, B, H, W, &p));
    *bytes = p.total;
    if (h->split_min > 0 && B >= h->split_min && B >= 2) {  // two half-batch plans side by side (mdpt_forward)
        Plan p0, p1;
        CHK(make_plan(h, B / 2, H, W, &p0));
        CHK(make_plan(h, B - B / 2, H, W, &p1));
        if (rup256(p0.total) + p1.total > *bytes) *bytes = rup256(p0.total) + p1.total;
    }
    return 0;
}

int mdpt_set_batch_split(mdpt_handle* h, int32_t min_batch) {
    if (!h || min_batch < 0) return fail(MDPT_E_INVALID, "bad argument");
    h->split_min = min_batch == 1 ? 2 : min_batch;
    return 0;
}

int mdpt_set_grid_cache(mdpt_handle* h, int32_t on) {
    if (!h) return fail(MDPT_E_INVALID, "null handle");
    h->grid_cache = on ? 1 : 0;
    h->cache_clear();
    return 0;
}

int mdpt_set_latency_mode(mdpt_handle* h, int32_t on) {
    if (!h) return fail(MDPT_E_INVALID, "null handle");
    h->latency_mode = on ? 1 : 0;
    return 0;
}

int mdpt_set_nonfinite_propagation(mdpt_handle* h, int32_t on) {
    if (!h) return fail(MDPT_E_INVALID, "null handle");
    h->nonfinite_prop = on ? 1 : 0;
    return 0;
}

int mdpt_set_gemm_tile(mdpt_handle* h, int32_t tile) {
    if (!h || tile < 0 || tile > 6 || tile == 3) return fail(MDPT_E_INVALID, "tile must be 0 (auto), 1 (128x128), 2 (256x256 lockstep), 4 (256x128x32), 5 (8-phase 256x256) or 6 (64x64)");
    h->gemm_tile = tile;
    return 0;
}

static int forward_one(mdpt_handle* h, const Ctx& c, const void* image_bchw, int image_dtype, void* depth_bhw, int depth_dtype);

// The handle's internal side stream + two events, created on first use. WHICH stream: one that the GPU really runs beside the caller's stream
// (stream_probe.hip has the why and the measurement): up to four candidates of the default priority class, the first that passes the probe is kept
// for this caller stream. The probe costs a few launches and ONE host wait, once per (handle, caller stream) - the only host synchronisation of
// the library, never inside a stream capture (there the current choice, or the first candidate, is used unprobed; the graph keeps no stream).
// `scratch`: the plan's own 256 probe bytes of the caller's workspace (Plan::probe - no activation ever lives there, so the probe may run after
// kernels of this forward were queued). A probe in which EVERY candidate was rejected (the waiter spins 150 us at most: a GPU busy with other work
// can starve the setter) does not pin its fallback: the caller stream is probed again by the next forward, three times at most.
// Caveats: the host wait invalidates a global-mode stream capture another thread may have open at that moment (probe once before capturing, or
// switch the probe off, mdpt_debug_set_side_stream_probe); a stream handle the runtime recycles after hipStreamDestroy keeps the earlier pick.
static int ensure_side_stream(mdpt_handle* h, hipStream_t s0, void* scratch) {
    if (!h->ev_fork) {
        CHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        CHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    auto candidate = [&](int i) -> int {
        while (h->side_ncand <= i) {
            int least = 0, greatest = 0;
            CHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
            const int prio = h->side_prio > 0 ? least : h->side_prio < 0 ? greatest : 0;
            CHK(hipStreamCreateWithPriority(&h->side_cand[h->side_ncand], hipStreamNonBlocking, prio));
            ++h->side_ncand;
        }
        return 0;
    };
    if (h->side_stream && !h->side_probe) return 0;
    for (int k = 0; k < (h->side_nfor < 4 ? h->side_nfor : 4); ++k)
        if (h->side_for[k] == s0) { h->side_stream = h->side_cand[h->side_pick[k]]; return 0; }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s0 && hipStreamIsCapturing(s0, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (!h->side_probe || cap != hipStreamCaptureStatusNone) {
        if (!h->side_stream) { CHK(candidate(0)); h->side_stream = h->side_cand[0]; }
        return 0;
    }
    int chosen = -1;
    for (int i = 0; i < 4; ++i) {
        CHK(candidate(i));
        unsigned* words = (unsigned*)scratch;
        unsigned seen = 0;
        CHK(mdpt_launch_queue_probe(words, words + 1, s0, h->side_cand[i], h->ev_fork));
        CHK(hipMemcpyAsync(&seen, words + 1, sizeof(seen), hipMemcpyDeviceToHost, s0));
        CHK(hipStreamSynchronize(s0));
        CHK(hipStreamSynchronize(h->side_cand[i]));
        if (seen) { chosen = i; break; }
        ++h->side_rejected;
    }
    if (chosen < 0) {  // nobody ran beside the caller's stream this time
        h->side_stream = h->side_cand[0];
        if (++h->side_unresolved <= 3) return 0;  // not recorded: probed again next time
        chosen = 0;
    }
    h->side_stream = h->side_cand[chosen];
    h->side_for[h->side_nfor & 3] = s0;
    h->side_pick[h->side_nfor & 3] = chosen;
    ++h->side_nfor;
    return 0;
}

static inline size_t dtype_bytes(int dt) { return dt == MDPT_DTYPE_F32 ? 4 : 2; }

int mdpt_forward(mdpt_handle* h, const void* image_bchw, int32_t image_dtype, int32_t B, int32_t H, int32_t W, void* depth_bhw,
                 int32_t depth_dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !image_bchw || !depth_bhw) return fail(MDPT_E_INVALID, "null argument");
    for (int dt : {image_dtype, depth_dtype})
        if (dt != MDPT_DTYPE_F32 && dt != MDPT_DTYPE_BF16 && dt != MDPT_DTYPE_F16) return fail(MDPT_E_INVALID, "bad tensor dtype %d", dt);
    if (h->split_min > 0 && B >= h->split_min && B >= 2 && h->dbg_block < 0) {
        // two half batches, one on the caller's stream, one on the side stream; joined before returning to the caller's stream
        const int B0 = B / 2, B1 = B - B0;
        Plan p0, p1;
        CHK(make_plan(h, B0, H, W, &p0));
        CHK(make_plan(h, B1, H, W, &p1));
        CHK(check_ws(h, p0, workspace, workspace_bytes));
        const size_t off1 = rup256(p0.total);
        if (workspace_bytes < off1 + p1.total) return fail(MDPT_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", off1 + p1.total, workspace_bytes);
        hipStream_t s0 = (hipStream_t)stream;
        CHK(ensure_side_stream(h, s0, (char*)workspace + p0.probe));
        CHK(hipEventRecord(h->ev_fork, s0));
        CHK(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
        Ctx c0, c1;
        c0.h = h; c0.p = p0; c0.ws = (char*)workspace; c0.s = s0; c0.split = true;
        c1.h = h; c1.p = p1; c1.ws = (char*)workspace + off1; c1.s = h->side_stream; c1.split = true;
        c0.consts_cached = h->cache_hit(c0.ws, B0, H, W);
        c1.consts_cached = h->cache_hit(c1.ws, B1, H, W);
        h->cache_clear();
        const size_t in_stride = (size_t)3 * H * W * dtype_bytes(image_dtype), out_stride = (size_t)H * W * dtype_bytes(depth_dtype);
        // Whatever happens after the fork, the side stream is joined back into the caller's stream before returning: kernels already
        // queued there keep using the second half of the workspace and the caller's tensors, which the caller may free or reuse on its
        // own stream as soon as this call returns (also on the error path).
        int rc = forward_one(h, c0, image_bchw, image_dtype, depth_bhw, depth_dtype);
        if (rc == 0) rc = forward_one(h, c1, (const char*)image_bchw + in_stride * B0, image_dtype, (char*)depth_bhw + out_stride * B0, depth_dtype);
        const hipError_t ej = hipEventRecord(h->ev_join, h->side_stream);
        const hipError_t ew = ej == hipSuccess ? hipStreamWaitEvent(s0, h->ev_join, 0) : ej;
        if (ew != hipSuccess) hipStreamSynchronize(h->side_stream);  // last resort: never leave the side stream running un-joined
        h->has_last = false;  // taps live in two half-batch plans: mdpt_export_tap is for unsplit (small) batches
        if (rc != 0) return rc;
        CHK(ew);
        if (h->grid_cache && h->dbg_block < 0) { h->cache_store(0, c0.ws, B0, H, W); h->cache_store(1, c1.ws, B1, H, W); }
        return 0;
    }
    const bool hit = h->cache_hit(workspace, B, H, W);
    Ctx c;
    CHK(make_ctx(h, B, H, W, workspace, workspace_bytes, stream, &c));
    c.consts_cached = hit;
    CHK(forward_one(h, c, image_bchw, image_dtype, depth_bhw, depth_dtype));
    if (h->grid_cache && h->dbg_block < 0) h->cache_store(0, workspace, B, H, W);
    return 0;
}

// ---- DPTModel.inference's device half (reference dpt_model.py:87-109: prepare_image_bgr -> forward), SURVEY 8(f) row 1 "fused with patchify"
int mdpt_forward_bgr(mdpt_handle* h, const void* bgr_u8_hwc, int32_t in_h, int32_t in_w, int32_t image_dtype, int32_t H, int32_t W, const float rgb_mean[3],
                     const float rgb_std[3], int32_t interpolation, void* depth_hw, int32_t depth_dtype, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !bgr_u8_hwc || !depth_hw || !rgb_mean || !rgb_std) return fail(MDPT_E_INVALID, "null argument");
    for (int dt : {image_dtype, depth_dtype})
        if (dt != MDPT_DTYPE_F32 && dt != MDPT_DTYPE_BF16 && dt != MDPT_DTYPE_F16) return fail(MDPT_E_INVALID, "bad tensor dtype %d", dt);
    if (interpolation != MDPT_INTERP_BILINEAR && interpolation != MDPT_INTERP_BICUBIC)
        return fail(MDPT_E_UNSUPPORTED, "interpolation %d: antialiased resize exists for bilinear and bicubic only (as in torch)", interpolation);
    if (in_h <= 0 || in_w <= 0) return fail(MDPT_E_INVALID, "bad image size %dx%d", in_h, in_w);
    const bool hit = h->cache_hit(workspace, 1, H, W);
    Ctx c;
    CHK(make_ctx(h, 1, H, W, workspace, workspace_bytes, stream, &c));
    c.consts_cached = hit;
    c.bgr.ptr = (const unsigned char*)bgr_u8_hwc; c.bgr.ih = in_h; c.bgr.iw = in_w; c.bgr.round_dtype = image_dtype; c.bgr.interp = interpolation;
    for (int i = 0; i < 3; ++i) { c.bgr.mean[i] = rgb_mean[i]; c.bgr.inv_std[i] = 1.0f / rgb_std[i]; }  // patch_embed.py:38-39,62
    CHK(forward_one(h, c, nullptr, image_dtype, depth_hw, depth_dtype));
    if (h->grid_cache && h->dbg_block < 0) h->cache_store(0, workspace, 1, H, W);
    return 0;
}

static int forward_body(mdpt_handle* h, const Ctx& c, const void* image_bchw, int image_dtype, void* depth_bhw, int depth_dtype);

// Non-finite propagation (mdpt_set_nonfinite_propagation, default on): the reference's forward (dpt_model.py:61-83) turns an image with a NaN / inf
// pixel into an all-NaN depth map - the value reaches every token through the first attention and torch's ReLU keeps it. Here the saturating fp16
// operand converts (v_med3) and the v_max ReLUs would return a finite, plausible map instead. The im2col kernel, which reads every pixel anyway,
// flags such images in B words of the plan and one small launch behind the head writes their maps as NaN: two stream-ordered operations per
// forward (graph-capturable, nothing on the host). uint8 sources (mdpt_forward_bgr) cannot hold a non-finite value and skip both.
static int forward_one(mdpt_handle* h, const Ctx& c0, const void* image_bchw, int image_dtype, void* depth_bhw, int depth_dtype) {
    if (!h->nonfinite_prop || !image_bchw || h->dbg_block >= 0) return forward_body(h, c0, image_bchw, image_dtype, depth_bhw, depth_dtype);
    Ctx c = c0;
    c.poison = c.at<unsigned>(c.p.poison);
    CHK(hipMemsetAsync(c.poison, 0, (size_t)c.p.B * 4, c.s));
    CHK(forward_body(h, c, image_bchw, image_dtype, depth_bhw, depth_dtype));
    CHK(OPLC(mdpt_launch_poison_depth, depth_bhw, depth_dtype, c.poison, c.p.B, (size_t)c.p.H * c.p.W, c.s));
    return 0;
}

static int forward_body(mdpt_handle* h, const Ctx& c, const void* image_bchw, int image_dtype, void* depth_bhw, int depth_dtype) {
    if (h->swin) {
        CHK(run_patch_embed_swin(c, image_bchw, image_dtype, nullptr));
        h->last_plan = c.p;
        h->has_last = true;
        CHK(run_encoder_swin(c, nullptr));
        CHK(run_reassemble_swin(c));
        CHK(run_fusion(c, true));
        CHK(run_head(c, depth_bhw, depth_dtype, head_upsamples_bf16(h)));
        return 0;
    }
    CHK(run_patch_embed_fused(c, image_bchw, image_dtype));
    h->last_plan = c.p;
    h->has_last = true;
    // measured on one box (tools/probes/b1_overlap_ab.py, profiles/r05_b1_overlap_ab.txt): wide encoders in the default mode gain (ViT-L 3.89 -> 3.82 ms,
    // BEiT-L 3.34 -> 3.23 ms), ViT-S loses (1.13 -> 1.16 ms: its blocks leave no idle CUs to fill, the fork / join events only add), latency mode
    // loses or ties (its K-split decoder convs own the encoder's partial-sum planes and stand down on the side stream) - hence the rule
    const bool overlap = h->overlap_reasm == 2 || (h->overlap_reasm == 1 && !h->latency_mode && h->F >= 1024);
    if (!c.split && overlap && h->dbg_block < 0) {
        // unsplit (small-batch) forward: reassembly branches run on the side stream beside the encoder (run_encoder, Ctx::tap_stream); whatever
        // happens in between, the side stream is joined back into the caller's stream before anything else is queued or returned
        CHK(ensure_side_stream(h, c.s, c.ws + c.p.probe));
        Ctx ce = c;
        ce.tap_stream = h->side_stream; ce.tap_event = h->ev_fork;
        const int rc = run_encoder(ce, nullptr);
        const hipError_t ej = hipEventRecord(h->ev_join, h->side_stream);
        const hipError_t ew = ej == hipSuccess ? hipStreamWaitEvent(c.s, h->ev_join, 0) : ej;
        if (ew != hipSuccess) hipStreamSynchronize(h->side_stream);
        if (rc != 0) return rc;
        CHK(ew);
        Ctx cf = c;
        cf.a1_done = true;
        CHK(run_fusion(cf, true));
        CHK(run_head(c, depth_bhw, depth_dtype, head_upsamples_bf16(h)));
        return 0;
    }
    CHK(run_encoder(c, nullptr));
    if (h->dbg_block >= 0) return 0;  // test hook: encoder truncated, skip the decoder
    CHK(run_reassemble(c));
    CHK(run_fusion(c, true));
    CHK(run_head(c, depth_bhw, depth_dtype, head_upsamples_bf16(h)));
    return 0;
}

int mdpt_patch_embed(mdpt_handle* h, const void* image_bchw, int32_t B, int32_t H, int32_t W, void* tokens_bnf, void* workspace,
                     size_t workspace_bytes, void* stream) {
    if (!h || !image_bchw || !tokens_bnf) return fail(MDPT_E_INVALID, "null argument");
    if (B <= 0 || H <= 0 || W <= 0 || H % h->P || W % h->P)
        return fail(MDPT_E_INVALID, "image size %dx%d must be divisible by the patch size %d", H, W, h->P);
    // PatchEmbed alone accepts odd grids (the reference only fails later, in fusion): plan with an even-rounded size
    Ctx c;
    if (h->swin) {
        CHK(make_ctx(h, B, rup(H, 32), rup(W, 32), workspace, workspace_bytes, stream, &c));
        c.p.sw.g0h = H / h->P; c.p.sw.g0w = W / h->P;
        CHK(run_patch_embed_swin(c, image_bchw, MDPT_DTYPE_F32, (float*)tokens_bnf));
        h->has_last = false;
        return 0;
    }
    const int He = rup(H, 2 * h->P), We = rup(W, 2 * h->P);
    CHK(make_ctx(h, B, He, We, workspace, workspace_bytes, stream, &c));
    const int Np = (H / h->P) * (W / h->P);
    Planes im = c.pl(c.p.im2col);
    CHK(OPLC(mdpt_launch_patchify, image_bchw, MDPT_DTYPE_F32, im.hi, im.lo, B, H, W, h->P, h->Kpatch, c.s));
    GemmParams g = base_params(c, h->M("patch_embed.proj.weight"), im, B * Np, h->Kpatch);
    g.bias = h->V("patch_embed.proj.bias");
    g.out_f32 = (float*)tokens_bnf; g.ldc = h->F;
    CHK(OPLC(mdpt_launch_gemm, g, c.s));
    h->has_last = false;
    return 0;
}

int mdpt_encoder(mdpt_handle* h, const void* tokens_bnf, int32_t B, int32_t gh, int32_t gw, void* const stage_out[4], void* workspace,
                 size_t workspace_bytes, void* stream) {
    if (!h || !tokens_bnf || !stage_out) return fail(MDPT_E_INVALID, "null argument");
    for (int i = 0; i < 4; ++i)
        if (!stage_out[i]) return fail(MDPT_E_INVALID, "null stage output %d", i);
    if (gh <= 0 || gw <= 0) return fail(MDPT_E_INVALID, "bad grid");
    Ctx c;
    if (h->swin) {  // tokens = PatchEmbed output [B, gh*gw, F0]; stage s output is [B, (gh>>s)*(gw>>s), F_s]
        CHK(make_ctx(h, B, gh * h->P, gw * h->P, workspace, workspace_bytes, stream, &c));
        const size_t n = (size_t)B * gh * gw * h->F;
        Planes xn = c.pl(c.p.sw.xn);
        CHK(hipMemcpyAsync(c.at<float>(c.p.sw.resid[0]), tokens_bnf, n * 4, hipMemcpyDeviceToDevice, c.s));
        CHK(swin_zero_pad_planes(c, B * gh * gw));
        CHK(OPLC(mdpt_launch_f32_to_planes, (const float*)tokens_bnf, xn.hi, xn.lo, (size_t)B * gh * gw, h->F, rup(h->F, 64), c.s));
        CHK(run_encoder_swin(c, stage_out));
        h->has_last = false;
        return 0;
    }
    CHK(make_ctx(h, B, rup(gh, 2) * h->P, rup(gw, 2) * h->P, workspace, workspace_bytes, stream, &c));
    // the encoder itself does not need an even grid: re-derive token counts for the true grid
    c.p.gh = gh; c.p.gw = gw; c.p.Np = gh * gw; c.p.N = c.p.Np + 1;
    if (rup(c.p.N, 8) > c.p.npad) return fail(MDPT_E_INVALID, "internal: plan too small");
    c.p.npad = rup(c.p.N, 8); c.p.npadv = rup(c.p.N, 64);
    if (is_beit(h)) {
        CHK(OPLC(mdpt_launch_memset_f32, c.at<float>(c.p.pos), 0.0f, (size_t)c.p.Np * h->F, c.s));
    } else {
        CHK(run_pos(c));
    }
    CHK(OPLC(mdpt_launch_init_tokens, c.at<float>(c.p.resid), h->V("imgencoder.cls_token"), is_beit(h) ? nullptr : h->V("imgencoder.posenc.cls_embedding"),
                                B, c.p.N, c.p.npad, h->F, c.s));
    CHK(OPLC(mdpt_launch_tokens_to_resid, (const float*)tokens_bnf, c.at<float>(c.p.pos), c.at<float>(c.p.resid), B, c.p.Np, c.p.npad, h->F, c.s));
    CHK(run_encoder(c, stage_out));
    h->has_last = false;
    return 0;
}

// mdpt_encoder + explicit attention weights of selected blocks (enable_optimizations=False semantics of the reference: the
// nn.Softmax output of every block is observable, experiments/attention_visualization.py:325-332). attn_out has num_blocks
// entries; a non-null entry receives that block's softmax(q k^T / sqrt(d) [+ bias]) as fp32 [B, heads, N, N].
int mdpt_encoder_probe(mdpt_handle* h, const void* tokens_bnf, int32_t B, int32_t gh, int32_t gw, void* const stage_out[4],
                       void* const* attn_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!attn_out) return fail(MDPT_E_INVALID, "null argument");
    return mdpt_encoder_probe_blocks(h, tokens_bnf, B, gh, gw, stage_out, attn_out, nullptr, workspace, workspace_bytes, stream);
}

// ... and / or the output tokens of selected blocks (what a forward hook on a TransformerBlock sees: demo_helpers/model_capture.py:54-59
// used by experiments/block_norm_visualization.py:282)
int mdpt_encoder_probe_blocks(mdpt_handle* h, const void* tokens_bnf, int32_t B, int32_t gh, int32_t gw, void* const stage_out[4],
                              void* const* attn_out, void* const* block_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !tokens_bnf || !stage_out) return fail(MDPT_E_INVALID, "null argument");
    for (int i = 0; i < 4; ++i)
        if (!stage_out[i]) return fail(MDPT_E_INVALID, "null stage output %d", i);
    if (gh <= 0 || gw <= 0) return fail(MDPT_E_INVALID, "bad grid");
    Ctx c;
    if (h->swin) {  // as mdpt_encoder, plus the window-attention weights of the listed blocks (stage-major block order)
        CHK(make_ctx(h, B, gh * h->P, gw * h->P, workspace, workspace_bytes, stream, &c));
        c.attn_dump = attn_out; c.block_dump = block_out;
        const size_t n = (size_t)B * gh * gw * h->F;
        Planes xn = c.pl(c.p.sw.xn);
        CHK(hipMemcpyAsync(c.at<float>(c.p.sw.resid[0]), tokens_bnf, n * 4, hipMemcpyDeviceToDevice, c.s));
        CHK(swin_zero_pad_planes(c, B * gh * gw));
        CHK(OPLC(mdpt_launch_f32_to_planes, (const float*)tokens_bnf, xn.hi, xn.lo, (size_t)B * gh * gw, h->F, rup(h->F, 64), c.s));
        CHK(run_encoder_swin(c, stage_out));
        h->has_last = false;
        return 0;
    }
    CHK(make_ctx(h, B, rup(gh, 2) * h->P, rup(gw, 2) * h->P, workspace, workspace_bytes, stream, &c));
    c.p.gh = gh; c.p.gw = gw; c.p.Np = gh * gw; c.p.N = c.p.Np + 1;
    if (rup(c.p.N, 8) > c.p.npad) return fail(MDPT_E_INVALID, "internal: plan too small");
    c.p.npad = rup(c.p.N, 8); c.p.npadv = rup(c.p.N, 64);
    c.attn_dump = attn_out; c.block_dump = block_out;
    if (is_beit(h)) {
        CHK(OPLC(mdpt_launch_memset_f32, c.at<float>(c.p.pos), 0.0f, (size_t)c.p.Np * h->F, c.s));
    } else {
        CHK(run_pos(c));
    }
    CHK(OPLC(mdpt_launch_init_tokens, c.at<float>(c.p.resid), h->V("imgencoder.cls_token"), is_beit(h) ? nullptr : h->V("imgencoder.posenc.cls_embedding"),
                                B, c.p.N, c.p.npad, h->F, c.s));
    CHK(OPLC(mdpt_launch_tokens_to_resid, (const float*)tokens_bnf, c.at<float>(c.p.pos), c.at<float>(c.p.resid), B, c.p.Np, c.p.npad, h->F, c.s));
    CHK(run_encoder(c, stage_out));
    h->has_last = false;
    return 0;
}

int mdpt_attn_probe_shape(const mdpt_handle* h, int32_t B, int32_t gh, int32_t gw, int32_t block, int64_t shape[4]) {
    if (!h || !shape || B <= 0 || gh <= 0 || gw <= 0 || block < 0 || block >= h->nblocks) return fail(MDPT_E_INVALID, "bad argument");
    if (!h->swin) {
        const int64_t n = (int64_t)gh * gw + 1;
        shape[0] = B; shape[1] = h->heads; shape[2] = n; shape[3] = n;
        return 0;
    }
    int s = 0, l = block;
    while (s < 4 && l >= h->sL[s]) { l -= h->sL[s]; ++s; }
    if (s >= 4) return fail(MDPT_E_INVALID, "block %d out of range", block);
    SwinStageGeom g;
    CHK(swin_geom(h, gh, gw, s, &g));
    shape[0] = (int64_t)B * g.nw; shape[1] = g.heads; shape[2] = g.wa; shape[3] = g.wa;
    return 0;
}

int mdpt_reassemble(mdpt_handle* h, const void* const stage_in[4], int32_t B, int32_t gh, int32_t gw, void* const maps_out[4],
                    void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !stage_in || !maps_out) return fail(MDPT_E_INVALID, "null argument");
    Ctx c;
    CHK(make_ctx(h, B, gh * h->P, gw * h->P, workspace, workspace_bytes, stream, &c));
    const Plan& p = c.p;
    for (int i = 0; i < 4; ++i)
        if (!stage_in[i] || !maps_out[i]) return fail(MDPT_E_INVALID, "null stage tensor %d", i);
    if (h->swin) {  // gh x gw = stage-0 patch grid; maps come out at 1, 1/2, 1/4, 1/8 of it
        for (int i = 0; i < 4; ++i) {
            Planes tp = c.pl(p.tap[i]);
            if (i == 0) CHK(swin_zero_pad_planes(c, B * gh * gw));
            CHK(OPLC(mdpt_launch_f32_to_planes, (const float*)stage_in[i], tp.hi, tp.lo, (size_t)B * (gh >> i) * (gw >> i), h->hid[i], h->hidp[i], c.s));
        }
        CHK(run_reassemble_swin(c));
        for (int i = 0; i < 4; ++i)
            CHK(OPLC(mdpt_launch_nhwc_to_nchw, c.at<float>(p.r_f32[i]), nullptr, nullptr, (float*)maps_out[i], B, gh >> i, gw >> i, h->C, h->Cp, c.s));
        h->has_last = false;
        return 0;
    }
    for (int i = 0; i < 4; ++i) {
        Planes tp = c.pl(p.tap[i]);
        CHK(OPLC(mdpt_launch_tokens_import, (const float*)stage_in[i], tp.hi, tp.lo, B, p.N, p.npad, h->F, c.s, tp.lo ? tp.f8 : 0, tp.f8_a8));
    }
    CHK(run_reassemble(c));
    const int sh[4] = {4 * gh, 2 * gh, gh, gh / 2}, sw[4] = {4 * gw, 2 * gw, gw, gw / 2};
    for (int i = 0; i < 4; ++i)
        CHK(OPLC(mdpt_launch_nhwc_to_nchw, c.at<float>(p.r_f32[i]), nullptr, nullptr, (float*)maps_out[i], B, sh[i], sw[i], h->C, h->Cp, c.s));
    h->has_last = false;
    return 0;
}

int mdpt_fusion(mdpt_handle* h, const void* const maps_in[4], int32_t B, int32_t gh, int32_t gw, void* fused_out, void* workspace,
                size_t workspace_bytes, void* stream) {
    if (!h || !maps_in || !fused_out) return fail(MDPT_E_INVALID, "null argument");
    Ctx c;
    CHK(make_ctx(h, B, gh * h->Pv, gw * h->Pv, workspace, workspace_bytes, stream, &c));
    const Plan& p = c.p;
    const int sh[4] = {4 * gh, 2 * gh, gh, gh / 2}, sw[4] = {4 * gw, 2 * gw, gw, gw / 2};
    for (int i = 0; i < 4; ++i) {
        if (!maps_in[i]) return fail(MDPT_E_INVALID, "null map %d", i);
        Planes rb = c.pl(p.r_bf[i]);
        CHK(OPLC(mdpt_launch_nchw_to_nhwc, (const float*)maps_in[i], c.at<float>(p.r_f32[i]), rb.hi, rb.lo, 1, B, sh[i], sw[i], h->C, h->Cp, c.s, rb.lo ? rb.f8 : 0, rb.f8_a8));
    }
    if (head_upsamples_bf16(h)) {
        // single-pass head: the fused forward hands the head the 16-bit output of the last projection and upsamples THAT (run_fusion(c, true) +
        // up_bf16.h arithmetic, inside head conv 1 or stand-alone: same bits). The stage-level call returns exactly that map, so a pipeline
        // driven sub-module by sub-module (hooks registered, simple_examples/internal_features.py) predicts the same bits as DPTModel.forward
        CHK(run_fusion(c, true));
        Planes fu = c.pl(p.fused);
        CHK(OPLC(mdpt_launch_upsample_bf16, c.at<op_t>(p.flo[0]), fu.hi, B, sh[0], sw[0], 2 * sh[0], 2 * sw[0], h->Cp, c.s));
        CHK(OPLC(mdpt_launch_nhwc_to_nchw, nullptr, fu.hi, nullptr, (float*)fused_out, B, 8 * gh, 8 * gw, h->C, h->Cp, c.s));
        h->has_last = false;
        return 0;
    }
    CHK(run_fusion(c));
    float* tmp = c.at<float>(p.scratch);
    CHK(OPLC(mdpt_launch_upsample, c.at<float>(p.flo[0]), nullptr, nullptr, tmp, B, sh[0], sw[0], 2 * sh[0], 2 * sw[0], h->Cp, c.s));
    CHK(OPLC(mdpt_launch_nhwc_to_nchw, tmp, nullptr, nullptr, (float*)fused_out, B, 8 * gh, 8 * gw, h->C, h->Cp, c.s));
    h->has_last = false;
    return 0;
}

// FusionModel.blocks[index] on its own (reference fusion_model.py:89-114 top-most block, :148-154 regular block; used by
// experiments/fusion_scaling.py:330-334): reasm_in [B,C,sh,sw] (+ prior_in [B,C,sh,sw], the previous block's output; must be NULL for
// index 3) -> out [B,C,2sh,2sw].
int mdpt_fusion_block(mdpt_handle* h, int32_t index, const void* reasm_in, const void* prior_in, int32_t B, int32_t sh, int32_t sw,
                      void* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !reasm_in || !out) return fail(MDPT_E_INVALID, "null argument");
    if (index < 0 || index > 3) return fail(MDPT_E_INVALID, "fusion block index %d out of range", index);
    if ((index == 3) != (prior_in == nullptr))
        return fail(MDPT_E_INVALID, "fusion block %d takes %s", index, index == 3 ? "one input (top-most block)" : "two inputs (reassembly map, previous fusion output)");
    // level `index` has spatial size (4, 2, 1, 1/2) x the virtual patch grid
    int gh, gw;
    if (index == 3) { gh = 2 * sh; gw = 2 * sw; }
    else {
        const int f = 4 >> index;
        if (sh % f || sw % f) return fail(MDPT_E_INVALID, "fusion block %d input %dx%d is not a multiple of %d", index, sh, sw, f);
        gh = sh / f; gw = sw / f;
    }
    Ctx c;
    CHK(make_ctx(h, B, gh * h->Pv, gw * h->Pv, workspace, workspace_bytes, stream, &c));
    const Plan& p = c.p;
    const int i = index;
    char pb[64];
    snprintf(pb, sizeof(pb), "fusion.blocks.%d", i);
    const std::string blk = pb;
    const size_t elems = (size_t)B * sh * sw * h->Cp;
    Planes rb = c.pl(p.r_bf[i]);
    CHK(OPLC(mdpt_launch_nchw_to_nhwc, (const float*)reasm_in, c.at<float>(p.r_f32[i]), rb.hi, rb.lo, 1, B, sh, sw, h->C, h->Cp, c.s, rb.lo ? rb.f8 : 0, rb.f8_a8));
    const float* x_f32 = c.at<float>(p.r_f32[i]);
    Planes x_bf = rb;
    if (i != 3) {
        // skip term of the reassembly RCU plus the previous fusion output: (r + prior), added in the second conv's epilogue
        float* skip = c.at<float>(p.scratch);
        CHK(OPLC(mdpt_launch_nchw_to_nhwc, (const float*)prior_in, skip, nullptr, nullptr, 0, B, sh, sw, h->C, h->Cp, c.s));
        CHK(OPLC(mdpt_launch_add_f32, skip, c.at<float>(p.r_f32[i]), elems, c.s));
        Planes a1 = c.pl(p.a1[i]);
        CHK(rcu_conv(c, blk + ".conv_reassembly." + rcu_seq(h) + ".1", rb, sh, sw, nullptr, nullptr, 0, 0, nullptr, a1, 1));
        x_bf = c.pl(p.x_bf[i]);
        CHK(rcu_conv(c, blk + ".conv_reassembly." + rcu_seq(h) + ".3", a1, sh, sw, skip, nullptr, 0, 0, c.at<float>(p.x_f32[i]), x_bf, 1));
        x_f32 = c.at<float>(p.x_f32[i]);
    }
    Planes b1 = c.pl(p.b1[i]), b2 = c.pl(p.b2[i]);
    CHK(rcu_conv(c, blk + "." + proj_seq(h) + ".0." + rcu_seq(h) + ".1", x_bf, sh, sw, nullptr, nullptr, 0, 0, nullptr, b1, 1));
    CHK(rcu_conv(c, blk + "." + proj_seq(h) + ".0." + rcu_seq(h) + ".3", b1, sh, sw, x_f32, nullptr, 0, 0, nullptr, b2, 0));
    const bool to_head16 = i == 0 && head_upsamples_bf16(h);  // the last block's output is the head's input: same 16-bit map as the fused path
    {
        GemmParams g = base_params(c, h->M(blk + "." + proj_seq(h) + ".2.weight"), b2, B * sh * sw, h->Cp);
        g.bias = h->V(blk + "." + proj_seq(h) + ".2.bias");
        if (to_head16) g.out_hi = c.at<op_t>(p.flo[0]);
        else g.out_f32 = c.at<float>(p.flo[i]);
        g.ldc = h->Cp;
        CHK(OPLC(mdpt_launch_gemm, g, c.s));
    }
    float* tmp = c.at<float>(p.scratch);
    if (to_head16) {
        CHK(OPLC(mdpt_launch_upsample_bf16, c.at<op_t>(p.flo[0]), (op_t*)tmp, B, sh, sw, 2 * sh, 2 * sw, h->Cp, c.s));
        CHK(OPLC(mdpt_launch_nhwc_to_nchw, nullptr, (const op_t*)tmp, nullptr, (float*)out, B, 2 * sh, 2 * sw, h->C, h->Cp, c.s));
        h->has_last = false;
        return 0;
    }
    CHK(OPLC(mdpt_launch_upsample, c.at<float>(p.flo[i]), nullptr, nullptr, tmp, B, sh, sw, 2 * sh, 2 * sw, h->Cp, c.s));
    CHK(OPLC(mdpt_launch_nhwc_to_nchw, tmp, nullptr, nullptr, (float*)out, B, 2 * sh, 2 * sw, h->C, h->Cp, c.s));
    h->has_last = false;
    return 0;
}

int mdpt_head(mdpt_handle* h, const void* fused_in, int32_t B, int32_t gh, int32_t gw, void* depth_bhw, void* workspace,
              size_t workspace_bytes, void* stream) {
    if (!h || !fused_in || !depth_bhw) return fail(MDPT_E_INVALID, "null argument");
    Ctx c;
    CHK(make_ctx(h, B, gh * h->Pv, gw * h->Pv, workspace, workspace_bytes, stream, &c));
    Planes fu = c.pl(c.p.fused);
    CHK(OPLC(mdpt_launch_nchw_to_nhwc, (const float*)fused_in, nullptr, fu.hi, fu.lo, 0, B, 8 * gh, 8 * gw, h->C, h->Cp, c.s, fu.lo ? fu.f8 : 0, fu.f8_a8));
    CHK(run_head(c, (float*)depth_bhw));
    h->has_last = false;
    h->stage_plan = c.p; h->has_stage_plan = true;  // (mdpt_debug_read "h1" / "fused" / "h1u" of this call: tools/probes/gpu_f8_h1_check.py)
    return 0;
}

int mdpt_export_tap(mdpt_handle* h, int32_t which, void* out_f32, void* workspace, size_t workspace_bytes, void* stream) {
    if (!h || !out_f32) return fail(MDPT_E_INVALID, "null argument");
    if (!h->has_last) return fail(MDPT_E_STATE, "mdpt_export_tap needs a preceding mdpt_forward on this workspace");
    Ctx c;
    c.h = h; c.p = h->last_plan; c.ws = (char*)workspace; c.s = (hipStream_t)stream;
    CHK(check_ws(h, c.p, workspace, workspace_bytes));
    const Plan& p = c.p;
    const int sh[4] = {4 * p.gh, 2 * p.gh, p.gh, p.gh / 2}, sw[4] = {4 * p.gw, 2 * p.gw, p.gw, p.gw / 2};
    if (which >= 0 && which < 4 && h->swin) {
        const size_t n = (size_t)p.B * (p.sw.g0h >> which) * (p.sw.g0w >> which) * h->hid[which];
        CHK(hipMemcpyAsync(out_f32, c.at<float>(p.sw.resid[which]), n * 4, hipMemcpyDeviceToDevice, c.s));
    } else if (which >= 0 && which < 4) {
        Planes tp = c.pl(p.tap[which]);
        CHK(OPLC(mdpt_launch_tokens_export, tp.hi, tp.lo, nullptr, (float*)out_f32, p.B, p.N, p.npad, h->F, 0, c.s, tp.lo ? tp.f8 : 0));
    } else if (which >= 4 && which < 8) {
        const int i = which - 4;
        CHK(OPLC(mdpt_launch_nhwc_to_nchw, c.at<float>(p.r_f32[i]), nullptr, nullptr, (float*)out_f32, p.B, sh[i], sw[i], h->C, h->Cp, c.s));
    } else if (which == 8) {
        float* tmp = c.at<float>(p.scratch);
        if (head_upsamples_bf16(h)) {  // the forward left the last projection as a bf16 map (run_fusion(c, true)): same upsample as the head's
            CHK(OPLC(mdpt_launch_upsample_bf16, c.at<op_t>(p.flo[0]), (op_t*)tmp, p.B, sh[0], sw[0], 2 * sh[0], 2 * sw[0], h->Cp, c.s));
            CHK(OPLC(mdpt_launch_nhwc_to_nchw, nullptr, (const op_t*)tmp, nullptr, (float*)out_f32, p.B, 8 * p.gh, 8 * p.gw, h->C, h->Cp, c.s));
        } else {
            CHK(OPLC(mdpt_launch_upsample, c.at<float>(p.flo[0]), nullptr, nullptr, tmp, p.B, sh[0], sw[0], 2 * sh[0], 2 * sw[0], h->Cp, c.s));
            CHK(OPLC(mdpt_launch_nhwc_to_nchw, tmp, nullptr, nullptr, (float*)out_f32, p.B, 8 * p.gh, 8 * p.gw, h->C, h->Cp, c.s));
        }
    } else {
        return fail(MDPT_E_INVALID, "unknown tap %d", which);
    }
    return 0;
}

// ---- PatchEmbed.prepare_image (reference v2_depthanything/patch_embed.py:103-145): resize + BGR->RGB + normalise on the GPU
int mdpt_prepare_image(const void* bgr_u8_hwc, int32_t in_h, int32_t in_w, void* out_chw, int32_t out_dtype, int32_t out_h, int32_t out_w,
                       const float rgb_mean[3], const float rgb_std[3], int32_t interpolation, void* stream) {
    if (!bgr_u8_hwc || !out_chw || !rgb_mean || !rgb_std) return fail(MDPT_E_INVALID, "null argument");
    if (out_dtype != MDPT_DTYPE_F32 && out_dtype != MDPT_DTYPE_BF16 && out_dtype != MDPT_DTYPE_F16) return fail(MDPT_E_INVALID, "bad tensor dtype %d", out_dtype);
    if (interpolation != MDPT_INTERP_BILINEAR && interpolation != MDPT_INTERP_BICUBIC)
        return fail(MDPT_E_UNSUPPORTED, "interpolation %d: antialiased resize exists for bilinear and bicubic only (as in torch)", interpolation);
    if (in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return fail(MDPT_E_INVALID, "bad image size %dx%d -> %dx%d", in_h, in_w, out_h, out_w);
    const float inv_std[3] = {1.0f / rgb_std[0], 1.0f / rgb_std[1], 1.0f / rgb_std[2]};  // patch_embed.py:38-39,62
    CHK(mdpt_launch_prepare_image_bf16((const unsigned char*)bgr_u8_hwc, out_chw, out_dtype, in_h, in_w, out_h, out_w, rgb_mean, inv_std, interpolation, (hipStream_t)stream));
    return 0;
}

// ---- depth post-processing (demo_helpers/postprocess.py, run_3dviewer.py:576-590)
int mdpt_post_minmax(const void* in_f32, size_t count, void* minmax_out, void* scratch8, void* stream) {
    if (!in_f32 || !minmax_out || !scratch8 || count == 0) return fail(MDPT_E_INVALID, "null argument / empty input");
    CHK(mdpt_launch_post_minmax((const float*)in_f32, count, (float*)minmax_out, (unsigned*)scratch8, (hipStream_t)stream));
    return 0;
}

int mdpt_post_scale_prediction(const void* in_bhw_f32, int32_t B, int32_t in_h, int32_t in_w, void* out_bhw_f32, int32_t out_h,
                               int32_t out_w, void* minmax_out, void* scratch8, void* stream) {
    if (!in_bhw_f32 || !out_bhw_f32 || (minmax_out && !scratch8)) return fail(MDPT_E_INVALID, "null argument");
    if (B <= 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return fail(MDPT_E_INVALID, "bad size %dx%dx%d -> %dx%d", B, in_h, in_w, out_h, out_w);
    CHK(mdpt_launch_post_scale((const float*)in_bhw_f32, (float*)out_bhw_f32, B, in_h, in_w, out_h, out_w, (float*)minmax_out,
                               (unsigned*)scratch8, (hipStream_t)stream));
    return 0;
}

int mdpt_post_normalize(const void* in_f32, size_t count, const void* minmax, void* out, int32_t mode, int32_t lossy, void* stream) {
    if (!in_f32 || !out || count == 0) return fail(MDPT_E_INVALID, "null argument / empty input");
    if (mode < MDPT_POST_F32 || mode > MDPT_POST_U24) return fail(MDPT_E_INVALID, "unknown post-processing mode %d", mode);
    CHK(mdpt_launch_post_normalize((const float*)in_f32, (const float*)minmax, out, count, mode, lossy, (hipStream_t)stream));
    return 0;
}

}  // extern "C"

// libmdpt: C ABI (include/mdpt.h) + host-side orchestration of the DPT forward path on one MI355X.
//
// What lives here: config validation, the parameter inventory (reference "new format" key names), the one-time
// weight repack plan, the activation workspace plan (bump allocation inside a caller-provided HBM buffer) and the
// launch sequence of the HIP kernels in gemm.hip / attention.hip / elementwise.hip. No device memory is allocated
// here and nothing synchronises: every launch goes on the caller's stream (reference contract: work is enqueued on
// the current torch stream, demo_helpers/misc.py:30-38).
//
// Stage structure mirrors DPTModel.forward (reference muggled_dpt/dpt_model.py:61-83):
//   patch_embed -> imgencoder (4 taps) -> reassemble -> fusion -> head
// Internal layouts: tokens are [B, npad, F] (npad = N rounded up to 8; pad rows stay finite and are never read by
// real rows), feature maps are NHWC with channels padded to 64 (pad channels are exactly zero).

#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/mdpt.h"
#include "mdpt_kernels.h"

// Every operand-format dependent launcher exists twice (op_types.h): mdpt_kernels.h declared the *_bf16 set, here is the *_f16 one.
// The host side never looks inside an operand plane - `op_t*` is an opaque 2-byte-element pointer here - and picks the set per handle.
#undef MDPT_FN
#define MDPT_FN(name) name##_f16
extern "C" {
#include "mdpt_launchers.inc"
}
#undef MDPT_FN
#define MDPT_FN(name) name##_bf16
#define OPL_(f16, fn, ...) ((f16) ? fn##_f16(__VA_ARGS__) : fn##_bf16(__VA_ARGS__))
#define OPLC(fn, ...) OPL_(c.h->f16, fn, __VA_ARGS__)   // inside a stage driver (a Ctx named c)
#define OPLH(fn, ...) OPL_(h->f16, fn, __VA_ARGS__)     // with only the handle in scope
#define OPLG(fn, ...) OPL_(g_debug_f16, fn, __VA_ARGS__)  // handle-less test hooks (mdpt_debug_set_operand_format)
// host-only predicates of the kernel files: the same answer in both builds
#define mdpt_head_tail_supported mdpt_head_tail_supported_bf16
#define mdpt_head_tail_scale_ok mdpt_head_tail_scale_ok_bf16
#define mdpt_beit_relpos_elen mdpt_beit_relpos_elen_bf16
#define mdpt_gemm_resolves_to_pp256 mdpt_gemm_resolves_to_pp256_bf16
#define mdpt_conv3h_supported mdpt_conv3h_supported_bf16

namespace {

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

#define CHK(expr)                                                                                                   \
    do {                                                                                                            \
        int e_ = (int)(expr);                                                                                       \
        if (e_ != 0) {                                                                                              \
            if (e_ > 0) return fail(e_, "%s: hip error %d (%s)", #expr, e_, hipGetErrorString((hipError_t)e_));      \
            return e_;                                                                                              \
        }                                                                                                           \
    } while (0)

inline int rup(int v, int m) { return (v + m - 1) / m * m; }
inline size_t rup256(size_t v) { return (v + 255) & ~(size_t)255; }

struct WeightSpec {
    std::string name;
    int ndim;
    int64_t shape[4];
    const void* ptr;
    int dtype;  // MDPT_DTYPE_* of the bound device tensor
    size_t numel() const {
        size_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
        return n;
    }
};

// Op classes: every contraction of the path belongs to one; a class runs 1 MFMA pass (operands rounded to one 16-bit plane) or 3 (hi + lo
// split planes). The uniform modes set all classes alike; MDPT_PREC_MIXED / mdpt_set_class_passes choose per class (include/mdpt.h).
enum { CLS_PATCH = MDPT_CLASS_PATCH, CLS_QKV = MDPT_CLASS_QKV, CLS_ATTN = MDPT_CLASS_ATTN, CLS_PROJ = MDPT_CLASS_PROJ, CLS_FC1 = MDPT_CLASS_FC1,
       CLS_FC2 = MDPT_CLASS_FC2, CLS_REASM = MDPT_CLASS_REASM, CLS_FUSION = MDPT_CLASS_FUSION, CLS_HEAD = MDPT_CLASS_HEAD,
       CLS_FUSION_IN = MDPT_CLASS_FUSION_IN, NCLS = MDPT_NUM_CLASSES };

int mat_class(const std::string& src) {
    if (src.compare(0, 12, "patch_embed.") == 0) return CLS_PATCH;
    if (src.compare(0, 11, "reassemble.") == 0) return CLS_REASM;
    if (src.compare(0, 7, "fusion.") == 0) return src.find(".conv_reassembly.") != std::string::npos ? CLS_FUSION_IN : CLS_FUSION;
    if (src.compare(0, 5, "head.") == 0) return CLS_HEAD;
    if (src.find(".attn.qkv.") != std::string::npos) return CLS_QKV;
    if (src.find(".attn.proj.") != std::string::npos) return CLS_PROJ;
    if (src.find(".mlp.layers.0.") != std::string::npos || src.find("inner_linear_doubled") != std::string::npos) return CLS_FC1;
    if (src.find(".mlp.layers.2.") != std::string::npos || src.find("outer_linear") != std::string::npos) return CLS_FC2;
    if (src.find("patch_merge_layers") != std::string::npos) return CLS_PROJ;  // SwinV2 patch merge: a token-mixing projection
    return CLS_PROJ;
}

struct Mat {  // packed operand panel [Np][Kp]
    std::string src;
    int cls;  // CLS_*
    std::string row_scale;  // name of a per-output-feature fp32 parameter folded into the rows at pack time ("" = none)
    int kind, N, K, Np, Kp, ksz;
    size_t off_hi, off_lo;
    op_t* hi;
    op_t* lo;
};

struct Vec {  // packed fp32 vector (zero padded)
    std::string src;
    std::string scale;  // name of a parameter multiplied in element-wise at pack time ("" = none); src then carries an "@..." suffix
    int n, np;
    size_t off;
    float* ptr;
};

struct Planes {
    op_t* hi = nullptr;
    op_t* lo = nullptr;
};

const char* kStageNames[4] = {"spatial_upx4", "spatial_upx2", "spatial_noscale", "spatial_downx2"};

// activation workspace layout for one (B, H, W)
struct Plan {
    int B, H, W, gh, gw, Np, N, npad, npadv;
    size_t total;
    // byte offsets (SIZE_MAX = absent)
    size_t im2col[2], pos, resid, xn[2], q[2], k[2], vt[2], att[2], hbuf[2], tap[4][2], tapf32;
    size_t t[4][2], u0[2], u1[2], d3[2];
    size_t r_f32[4], r_bf[4][2];
    size_t a1[4][2], x_f32[4], x_bf[4][2], b1[4][2], b2[4][2], flo[4];
    size_t fused[2], h1, h1u[2], scratch;
    size_t scratch_floats;
    size_t tokr[2], cbuf, relpos_lut, relpos_tq, relpos_tk;  // BEiT: readout-projected tokens, per-image cls term, bias LUT
    size_t wrc_mean, wrc_tab;                                 // [B, wrc_maxk] operand-format column means, fp32 [B, wrc_maxn] per-image bias table
    size_t swi;                                               // ViT-G: fp32 [rows, 2*hidden] output of the doubled inner linear
    // SwinV2: stage-0 patch grid, per-stage residual streams (fp32, = the taps), shared GEMM fp32 output, token planes,
    // window operands, window maps (plain / shifted) and the position-bias LUT
    struct {
        int g0h, g0w;
        size_t resid[4], x, xn[2], q[2], k[2], vt[2], att[2], hb[2], lut, lut_stride, tq, tk, rowmap[2], region[2], tokmap[2], vtokmap[2];
    } sw;
};

}  // namespace

struct mdpt_handle {
    mdpt_config cfg;
    int F, heads, nblocks, bps, P, C, Cp, C2, C2p, Kpatch;
    int hid[4], hidp[4];
    bool swin;
    int Pv;  // patch size seen by fusion/head: the finest reassembly map is (4H/Pv) x (4W/Pv); = P except SwinV2 (16)
    int gh_hidden, gh_hidden_p;  // ViT-G SwiGLU hidden width (and padded to 64), 0 otherwise
    int sH[4], sL[4], swh, sww, spre[4];  // SwinV2: heads / layers per stage, target window, pretrained window sizes (0 = None)
    bool f16;       // operand format: fp16 (v_mfma_*_f16, *_f16 launchers) instead of bf16
    int np[NCLS];   // MFMA passes per op class: 1 or 3
    bool x3c(int cls) const { return np[cls] == 3; }
    // token-mean compensation of the weight rounding (fp16 operand modes, single-pass encoder Linears): see wrc_bias() below
    bool wrc_on;
    bool wrc(int cls) const { return wrc_on && f16 && !swin && np[cls] == 1 && (cls == CLS_QKV || cls == CLS_PROJ || cls == CLS_FC1 || cls == CLS_FC2); }
    int wrc_maxn, wrc_maxk;  // widest compensated matrix (table / mean buffers of the plan)
    int gemm_tile;
    std::vector<WeightSpec> specs;
    std::map<std::string, int> spec_index;
    std::vector<Mat> mats;
    std::map<std::string, int> mat_index;
    std::vector<Vec> vecs;
    std::map<std::string, int> vec_index;
    size_t packed_total;
    size_t zero_off;
    op_t* zero_page;
    bool finalized;
    // last forward (for export taps)
    Plan last_plan;
    bool has_last;
    int dbg_block, dbg_step;  // test hook: stop the encoder after (block, step); -1 = off
    // batch split: batches >= split_min run as two halves on the caller's stream and an internal side stream (fork / join with
    // events, no host sync) so that one half's kernels fill the tile-quantisation tails and epilogue phases of the other's
    int split_min;
    int latency_mode;  // mdpt_set_latency_mode: small launches may use summation orders that are not batch-invariant
    hipStream_t side_stream;
    hipEvent_t ev_fork, ev_join;
    ~mdpt_handle() {
        if (side_stream) hipStreamDestroy(side_stream);
        if (ev_fork) hipEventDestroy(ev_fork);
        if (ev_join) hipEventDestroy(ev_join);
    }

    void add_spec(const std::string& name, std::initializer_list<int64_t> shape) {
        WeightSpec s;
        s.name = name;
        s.ndim = (int)shape.size();
        int i = 0;
        for (int64_t d : shape) s.shape[i++] = d;
        for (; i < 4; ++i) s.shape[i] = 1;
        s.ptr = nullptr;
        s.dtype = MDPT_DTYPE_F32;
        spec_index[name] = (int)specs.size();
        specs.push_back(s);
    }
    void add_mat(const std::string& src, int kind, int N, int K, int Np, int Kp, int ksz) {
        Mat m;
        m.src = src; m.kind = kind; m.N = N; m.K = K; m.Np = Np; m.Kp = Kp; m.ksz = ksz;
        m.cls = mat_class(src);
        m.off_hi = packed_total;
        packed_total += rup256((size_t)Np * Kp * 2);
        m.off_lo = SIZE_MAX;
        if (x3c(m.cls) || wrc(m.cls)) {  // (a compensated single-pass class keeps the lo plane as the weight residue fp(W - fp(W)))
            if (wrc(m.cls)) { if (Np > wrc_maxn) wrc_maxn = Np; if (Kp > wrc_maxk) wrc_maxk = Kp; } m.off_lo = packed_total; packed_total += rup256((size_t)Np * Kp * 2); }
        m.hi = m.lo = nullptr;
        mat_index[src] = (int)mats.size();
        mats.push_back(m);
    }
    void add_vec(const std::string& src, int n, int np) {
        Vec v;
        v.src = src; v.n = n; v.np = np;
        v.off = packed_total;
        packed_total += rup256((size_t)np * 4);
        v.ptr = nullptr;
        vec_index[src] = (int)vecs.size();
        vecs.push_back(v);
    }
    const Mat& M(const std::string& name) const { return mats[mat_index.at(name)]; }
    const float* V(const std::string& name) const { return vecs[vec_index.at(name)].ptr; }
};

namespace {

std::string blk_name(const mdpt_handle* h, int block) {
    char buf[96];
    if (h->cfg.family == MDPT_FAMILY_DAV1) snprintf(buf, sizeof(buf), "imgencoder.blocks.%d", block);
    else snprintf(buf, sizeof(buf), "imgencoder.stages.%d.blocks.%d", block / h->bps, block % h->bps);
    return buf;
}

inline bool is_beit(const mdpt_handle* h) { return h->cfg.family == MDPT_FAMILY_BEIT; }
inline bool is_midas(const mdpt_handle* h) { return h->cfg.family == MDPT_FAMILY_BEIT || h->cfg.family == MDPT_FAMILY_SWINV2; }
// reference attribute names differ between the families (v2: fusion_model.py:100,138 / v31_beit, v31_swinv2 fusion_model.py)
inline const char* rcu_seq(const mdpt_handle* h) { return is_midas(h) ? "conv_seq" : "resconv_seq"; }
inline const char* proj_seq(const mdpt_handle* h) { return is_midas(h) ? "proj_seq" : "scale_proj_seq"; }

int build_inventory_swin_encoder(mdpt_handle* h);

int build_inventory_decoder(mdpt_handle* h);

int build_inventory(mdpt_handle* h) {
    const int F = h->F, P = h->P, C = h->C;
    const int G = h->cfg.base_patch_grid_h * h->cfg.base_patch_grid_w;
    h->packed_total = 0;
    h->wrc_maxn = h->wrc_maxk = 0;
    h->zero_off = 0;
    h->packed_total += 256;

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
    if (!h->x3c(CLS_HEAD) && mdpt_head_tail_supported(h->C2p))  // LDS image of the same weights for the fused head tail (head.hip)
        h->add_mat("head.proj_1ch.0.weight@kc32", MDPT_PACK_CONV3_KC32, 32, h->C2, 32, 9 * h->C2p, 3);
    h->add_vec("head.proj_1ch.0.bias", 32, 32);
    h->add_vec("head.proj_1ch.2.weight", 32, 32);
    h->add_vec("head.proj_1ch.2.bias", 1, 4);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// workspace planning
// ------------------------------------------------------------------------------------------------------------
struct Bump {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t o = off;
        off += rup256(bytes);
        return o;
    }
};

void take_planes(Bump& bump, bool x3, size_t elems, size_t out[2]) {
    out[0] = bump.take(elems * 2);
    out[1] = x3 ? bump.take(elems * 2) : SIZE_MAX;
}

// reassembly outputs, fusion and head buffers; p.Np / p.gh / p.gw = the "noscale" level (1/Pv of the image)
void plan_decoder(Bump& bump, const mdpt_handle* h, Plan& p, size_t min_scratch_floats) {
    // lo planes exist where the CONSUMING class runs three passes: the reassembly maps of levels 0..2 and a1 feed the conv_reassembly
    // units (CLS_FUSION_IN), level 3's map and everything else the projection path (CLS_FUSION)
    const bool x3 = h->x3c(CLS_FUSION), x3i = h->x3c(CLS_FUSION_IN), x3h = h->x3c(CLS_HEAD);
    const int B = p.B;
    const size_t px[4] = {(size_t)16 * p.Np, (size_t)4 * p.Np, (size_t)p.Np, (size_t)p.Np / 4};
    for (int i = 0; i < 4; ++i) {
        const size_t e = (size_t)B * px[i] * h->Cp;
        p.r_f32[i] = bump.take(e * 4);
        take_planes(bump, i == 3 ? x3 : x3i, e, p.r_bf[i]);
        take_planes(bump, x3i, e, p.a1[i]);
        p.x_f32[i] = bump.take(e * 4);
        take_planes(bump, x3, e, p.x_bf[i]);
        take_planes(bump, x3, e, p.b1[i]);
        take_planes(bump, x3, e, p.b2[i]);
        p.flo[i] = bump.take(e * 4);
    }
    const size_t fpx = (size_t)64 * p.Np;  // (8gh)*(8gw)
    take_planes(bump, x3h, (size_t)B * fpx * h->Cp, p.fused);
    // bf16 mode with the fused head tail (run_head): conv 1 writes a bf16 map and the full-resolution upsampled map never exists (ViT-L,
    // 504x504, batch 32: 2.1 GB + 0.7 GB of workspace that used to be reserved and never touched)
    const bool bf16_head = !x3h && mdpt_head_tail_supported(h->C2p) && mdpt_head_tail_scale_ok(8 * p.gh, 8 * p.gw, p.H, p.W);
    p.h1 = bump.take((size_t)B * fpx * h->C2p * (bf16_head ? 2 : 4));
    if (bf16_head) p.h1u[0] = p.h1u[1] = SIZE_MAX;
    else take_planes(bump, x3h, (size_t)B * p.H * p.W * h->C2p, p.h1u);
    p.scratch_floats = (size_t)B * fpx * h->Cp;
    if (min_scratch_floats > p.scratch_floats) p.scratch_floats = min_scratch_floats;
    p.scratch = bump.take(p.scratch_floats * 4);
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
    take_planes(bump, h->x3c(CLS_PATCH), (size_t)B * p.Np * h->Kpatch, p.im2col);
    p.pos = bump.take((size_t)p.Np * F * 4);
    p.resid = bump.take(rows * F * 4);
    take_planes(bump, h->x3c(CLS_QKV) || h->x3c(CLS_FC1), rows * F, p.xn);
    take_planes(bump, h->x3c(CLS_ATTN), (size_t)B * h->heads * p.npad * 64, p.q);
    take_planes(bump, h->x3c(CLS_ATTN), (size_t)B * h->heads * p.npad * 64, p.k);
    take_planes(bump, h->x3c(CLS_ATTN), (size_t)B * h->heads * 64 * p.npadv, p.vt);
    take_planes(bump, h->x3c(CLS_PROJ) || h->x3c(CLS_ATTN), rows * F, p.att);  // the 3-pass attention kernel always writes its lo plane
    take_planes(bump, h->x3c(CLS_FC2), rows * 4 * F, p.hbuf);
    p.swi = h->gh_hidden ? bump.take(rows * 2 * h->gh_hidden * 4) : SIZE_MAX;
    p.wrc_mean = h->wrc_maxk ? bump.take((size_t)B * h->wrc_maxk * 2) : SIZE_MAX;
    p.wrc_tab = h->wrc_maxn ? bump.take((size_t)B * h->wrc_maxn * 4) : SIZE_MAX;
    const bool x3 = h->x3c(CLS_REASM);
    for (int i = 0; i < 4; ++i) take_planes(bump, x3, rows * F, p.tap[i]);
    p.tapf32 = bump.take(rows * F * 4);
    const size_t px[4] = {(size_t)16 * p.Np, (size_t)4 * p.Np, (size_t)p.Np, (size_t)p.Np / 4};
    for (int i = 0; i < 4; ++i) take_planes(bump, x3, (size_t)B * p.Np * h->hidp[i], p.t[i]);
    take_planes(bump, x3, (size_t)B * px[0] * h->hidp[0], p.u0);
    take_planes(bump, x3, (size_t)B * px[1] * h->hidp[1], p.u1);
    take_planes(bump, x3, (size_t)B * px[3] * h->hidp[3], p.d3);
    plan_decoder(bump, h, p, rows * F);
    p.tokr[0] = p.tokr[1] = p.cbuf = p.relpos_lut = p.relpos_tq = p.relpos_tk = SIZE_MAX;
    if (is_beit(h)) {
        take_planes(bump, x3, (size_t)B * p.Np * F, p.tokr);
        p.cbuf = bump.take((size_t)B * F * 4);
        p.relpos_lut = bump.take((size_t)h->heads * mdpt_beit_relpos_elen(gh, gw) * 4 * (h->nblocks <= 32 ? h->nblocks : 1));  // one table per block
        p.relpos_tq = bump.take((size_t)p.npadv * 4);
        p.relpos_tk = bump.take((size_t)p.npadv * 4);
    }
    p.total = bump.off;
    return 0;
}

struct Ctx {
    const mdpt_handle* h;
    Plan p;
    char* ws;
    hipStream_t s;
    bool split = false;  // this context is one half of a two-stream batch split
    void* const* attn_dump = nullptr;  // per block: where to write softmax(q k^T) as fp32 [B,H,N,N] (null entries: skip)
    void* const* block_dump = nullptr; // per block: where to write the block's output tokens as fp32 [B,N,F] (null entries: skip)
    template <class T> T* at(size_t off) const { return off == SIZE_MAX ? nullptr : (T*)(ws + off); }
    Planes pl(const size_t o[2]) const {
        Planes r;
        r.hi = at<op_t>(o[0]);
        r.lo = at<op_t>(o[1]);
        return r;
    }
};

GemmParams base_params(const Ctx& c, const Mat& w, Planes a, int M, int lda) {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.npass = c.h->np[w.cls];  // the class of the weight matrix decides; an A buffer shared with a 3-pass class may carry an unused lo plane
    g.A_hi = a.hi; g.A_lo = g.npass == 3 ? a.lo : nullptr;
    g.W_hi = w.hi; g.W_lo = g.npass == 3 ? w.lo : nullptr;
    g.M = M; g.N = w.Np; g.K = w.Kp; g.lda = lda;
    g.zero_page = c.h->zero_page;
    g.amode = MDPT_A_DENSE; g.ekind = MDPT_E_GENERIC; g.tile = c.h->gemm_tile;
    g.throughput_mode = c.split ? 1 : 0;
    g.ldc = w.Np; g.ldr = w.Np;
    return g;
}

void as_conv(GemmParams& g, int Hi, int Wi, int Cin, int Ho, int Wo, int stride) {
    g.amode = MDPT_A_CONV3;
    g.Hi = Hi; g.Wi = Wi; g.Cin = Cin; g.Ho = Ho; g.Wo = Wo; g.cstride = stride;
}

int check_ws(const mdpt_handle* h, const Plan& p, const void* ws, size_t bytes) {
    if (!h->finalized) return fail(MDPT_E_STATE, "mdpt_finalize() has not been called");
    if (!ws || bytes < p.total) return fail(MDPT_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", p.total, bytes);
    if (((uintptr_t)ws) & 255) return fail(MDPT_E_WORKSPACE, "workspace must be 256-byte aligned");
    return 0;
}

// ---- stage: patch embed (fused form: writes the residual stream incl. position embedding)
int run_pos(const Ctx& c) {
    const mdpt_handle* h = c.h;
    return OPLC(mdpt_launch_posembed, h->V("imgencoder.posenc.base_patch_embedding"), c.at<float>(c.p.pos), h->cfg.base_patch_grid_h,
                                h->cfg.base_patch_grid_w, c.p.gh, c.p.gw, h->F, c.s);
}

int run_patch_embed_fused(const Ctx& c, const void* image, int image_dtype) {
    const mdpt_handle* h = c.h;
    const Plan& p = c.p;
    Planes im = c.pl(p.im2col);
    CHK(OPLC(mdpt_launch_patchify, image, image_dtype, im.hi, im.lo, p.B, p.H, p.W, h->P, h->Kpatch, c.s));
    const bool beit = is_beit(h);
    if (!beit) CHK(run_pos(c));
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
int wrc_bias(const Ctx& c, GemmParams& g, const Mat& w, const float* bias) {
    const mdpt_handle* h = c.h;
    if (!h->wrc(w.cls) || !w.lo) return 0;
    const Plan& p = c.p;
    op_t* mean = c.at<op_t>(p.wrc_mean);
    float* tab = c.at<float>(p.wrc_tab);
    // every step-th token estimates the shared component as well as all of them (tests/precision_budget/); the step depends on the token
    // count only, so an image's table does not depend on the batch it is part of
    const int step = p.N >= 1024 ? 8 : (p.N >= 256 ? 4 : 1);
    CHK(OPLC(mdpt_launch_colmean, g.A_hi, g.lda, p.B, p.npad, p.N, step, w.Kp, mean, c.s));
    CHK(OPLC(mdpt_launch_wrc_table, mean, w.lo, bias, tab, p.B, w.Np, w.Kp, c.s));
    g.bias = tab; g.bias_img_stride = w.Np; g.bias_img_rows = p.npad;
    return 0;
}

// ---- stage: encoder. taps_f32 != null: also emit fp32 copies of the 4 out-normed taps (reference layout)
int run_encoder(const Ctx& c, void* const taps_f32[4]) {
    const mdpt_handle* h = c.h;
    const Plan& p = c.p;
    const int F = h->F, rows = p.B * p.npad;
    float* resid = c.at<float>(p.resid);
    Planes xn = c.pl(p.xn), q = c.pl(p.q), k = c.pl(p.k), vt = c.pl(p.vt), att = c.pl(p.att), hb = c.pl(p.hbuf);
    CHK(OPLC(mdpt_launch_zero_vt_pad, vt.hi, vt.lo, p.B * h->heads * 64, p.N, p.npadv, c.s));
    // a 3-pass projection behind a 1-pass attention kernel (which writes no lo plane): the plane is zero, i.e. the projection keeps
    // the rounding of its A operand and loses only that of its weights
    if (att.lo && !h->x3c(CLS_ATTN)) CHK(hipMemsetAsync(att.lo, 0, (size_t)rows * F * 2, c.s));
    const Planes xn_qkv = {xn.hi, h->x3c(CLS_QKV) ? xn.lo : nullptr}, xn_fc1 = {xn.hi, h->x3c(CLS_FC1) ? xn.lo : nullptr};
#define DBG_STOP(step) if (h->dbg_block == b && h->dbg_step == (step)) return 0
    const size_t relpos_stride = is_beit(h) ? (size_t)h->heads * mdpt_beit_relpos_elen(p.gh, p.gw) : 0;
    const bool relpos_batched = is_beit(h) && h->nblocks <= 32;
    if (relpos_batched) {  // every block's relative-position table, resized to the current grid: one launch per forward
        BeitRelposBatch rb;
        memset(&rb, 0, sizeof(rb));
        for (int b = 0; b < h->nblocks; ++b) rb.ref[b] = h->V(blk_name(h, b) + ".attn.relpos_enc.ref_bias_lut");
        rb.ext0 = c.at<float>(p.relpos_lut); rb.ext_stride = relpos_stride;
        rb.tq = c.at<int>(p.relpos_tq); rb.tk = c.at<int>(p.relpos_tk);
        rb.n = h->nblocks; rb.heads = h->heads; rb.Gh = h->cfg.base_patch_grid_h; rb.Gw = h->cfg.base_patch_grid_w;
        rb.gh = p.gh; rb.gw = p.gw; rb.N = p.N; rb.ntok_pad = p.npadv;
        CHK(OPLC(mdpt_launch_beit_relpos_batch, rb, c.s));
    }
    for (int b = 0; b < h->nblocks; ++b) {
        const std::string n = blk_name(h, b);
        CHK(OPLC(mdpt_launch_layernorm, resid, h->V(n + ".norm1.weight"), h->V(n + ".norm1.bias"), xn_qkv.hi, xn_qkv.lo, nullptr, rows, F, c.s));
        DBG_STOP(0);
        {
            GemmParams g = base_params(c, h->M(n + ".attn.qkv.weight"), xn, rows, F);
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
            g.bias = h->V(n + ".attn.proj.bias@ls");  // layer scale folded into W and the bias at pack time
            g.acc_init = 1;
            g.resid = resid; g.out_f32 = resid; g.ldr = F; g.ldc = F;
            CHK(wrc_bias(c, g, h->M(n + ".attn.proj.weight"), g.bias));
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
        }
        DBG_STOP(3);
        CHK(OPLC(mdpt_launch_layernorm, resid, h->V(n + ".norm2.weight"), h->V(n + ".norm2.bias"), xn_fc1.hi, xn_fc1.lo, nullptr, rows, F, c.s));
        DBG_STOP(4);
        if (h->gh_hidden) {  // ViT-G: (a | b) = x W12^T + b12 ; hidden = silu(a) * b
            GemmParams g = base_params(c, h->M(n + ".mlp.inner_linear_doubled.weight"), xn, rows, F);
            g.bias = h->V(n + ".mlp.inner_linear_doubled.bias");
            g.out_f32 = c.at<float>(p.swi); g.ldc = 2 * h->gh_hidden;
            CHK(wrc_bias(c, g, h->M(n + ".mlp.inner_linear_doubled.weight"), g.bias));
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
            CHK(OPLC(mdpt_launch_swiglu, c.at<float>(p.swi), hb.hi, hb.lo, (size_t)rows, h->gh_hidden, h->gh_hidden_p, c.s));
        } else {
            GemmParams g = base_params(c, h->M(n + ".mlp.layers.0.weight"), xn, rows, F);
            g.bias = h->V(n + ".mlp.layers.0.bias");
            g.act = MDPT_ACT_GELU;
            g.out_hi = hb.hi; g.out_lo = hb.lo; g.ldc = 4 * F;
            CHK(wrc_bias(c, g, h->M(n + ".mlp.layers.0.weight"), g.bias));
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
        }
        DBG_STOP(5);
        {
            const bool giant = h->gh_hidden != 0;
            GemmParams g = base_params(c, h->M(giant ? n + ".mlp.outer_linear.weight" : n + ".mlp.layers.2.weight"), hb, rows,
                                       giant ? h->gh_hidden_p : 4 * F);
            g.bias = h->V(giant ? n + ".mlp.outer_linear.bias@ls" : n + ".mlp.layers.2.bias@ls");
            g.acc_init = 1;
            g.resid = resid; g.out_f32 = resid; g.ldr = F; g.ldc = F;
            CHK(wrc_bias(c, g, h->M(giant ? n + ".mlp.outer_linear.weight" : n + ".mlp.layers.2.weight"), g.bias));
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
        }
        DBG_STOP(6);
        if (c.block_dump && c.block_dump[b])  // TransformerBlock output (transformer_block.py:61-62), pad rows dropped
            CHK(OPLC(mdpt_launch_tokens_export, nullptr, nullptr, resid, (float*)c.block_dump[b], p.B, p.N, p.npad, F, 0, c.s));
        const bool v1 = h->cfg.family == MDPT_FAMILY_DAV1;
        if (v1 ? b >= h->nblocks - 4 : (b + 1) % h->bps == 0) {
            const int st = v1 ? b - (h->nblocks - 4) : b / h->bps;
            Planes tp = c.pl(p.tap[st]);
            float* f32 = taps_f32 ? c.at<float>(p.tapf32) : nullptr;
            if (is_beit(h)) {  // BEiT taps the raw residual stream (no out-norm, v31_beit/image_encoder_model.py:84-91)
                CHK(OPLC(mdpt_launch_tokens_import, resid, tp.hi, tp.lo, p.B, p.npad, p.npad, F, c.s));
                if (taps_f32) CHK(OPLC(mdpt_launch_tokens_export, nullptr, nullptr, resid, (float*)taps_f32[st], p.B, p.N, p.npad, F, 0, c.s));
            } else {
                CHK(OPLC(mdpt_launch_layernorm, resid, h->V("imgencoder.outnorm.weight"), h->V("imgencoder.outnorm.bias"), tp.hi, tp.lo, f32, rows, F, c.s));
                if (taps_f32)
                    CHK(OPLC(mdpt_launch_tokens_export, nullptr, nullptr, f32, (float*)taps_f32[st], p.B, p.N, p.npad, F, 0, c.s));
            }
        }
    }
    return 0;
}

int conv3_to_fusion(const Ctx& c, const Mat& w, Planes in, int Cin, int sh, int sw, const float* bias, const float* skip, const float* up_src,
                    int Hu, int Wu, float* out_f32, Planes out, int relu_bf16);

// ---- stage: reassemble
int run_reassemble(const Ctx& c) {
    const mdpt_handle* h = c.h;
    const Plan& p = c.p;
    const int F = h->F, gh = p.gh, gw = p.gw;
    for (int i = 0; i < 4; ++i) {
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
                g.out_hi = tr.hi; g.out_lo = tr.lo; g.ldc = F;
                CHK(OPLC(mdpt_launch_gemm, g, c.s));
            }
            tp = tr;
            tokens_mode = false;
        }
        {   // 1x1 conv on the patch tokens (cls row skipped by the A-row generator)
            GemmParams g = base_params(c, h->M(n + ".resample.0.weight"), tp, p.B * p.Np, F);
            if (tokens_mode) { g.amode = MDPT_A_TOKENS; g.tok_np = p.Np; g.tok_stride = p.npad; }
            g.bias = h->V(n + ".resample.0.bias");
            g.out_hi = t.hi; g.out_lo = t.lo; g.ldc = hp;
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
            g.out_hi = u.hi; g.out_lo = u.lo;
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
            src = u; sh = gh * kk; sw = gw * kk;
        } else if (i == 3) {  // 3x3 stride-2
            Planes d = c.pl(p.d3);
            GemmParams g = base_params(c, h->M(n + ".resample.1.weight"), t, p.B * (gh / 2) * (gw / 2), hp);
            as_conv(g, gh, gw, hp, gh / 2, gw / 2, 2);
            g.bias = h->V(n + ".resample.1.bias");
            g.out_hi = d.hi; g.out_lo = d.lo; g.ldc = hp;
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
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
        const bool three = h->np[w.cls] == 3;  // the weight's class decides (an input buffer may carry a lo plane this conv does not use)
        q.in = in.hi; q.in_lo = three ? in.lo : nullptr; q.w = w.hi; q.w_lo = three ? w.lo : nullptr; q.bias = bias; q.skip = skip; q.up_src = up_src; q.Hu = Hu; q.Wu = Wu;
        q.out_f32 = out_f32; q.out_bf = out.hi; q.out_bf_lo = out.lo; q.relu_bf = relu_bf16;
        q.B = c.p.B; q.H = sh; q.W = sw; q.Cin = Cin; q.Cout = 256;
        const long tiles256 = ((long)c.p.B * sh * sw + 255) / 256;
        if (tiles256 >= conv3h_min_tiles(c) && mdpt_conv3h_supported(q)) return OPLC(mdpt_launch_conv3h, q, c.s);
    }
    GemmParams g = base_params(c, w, in, c.p.B * sh * sw, Cin);
    as_conv(g, sh, sw, Cin, sh, sw, 1);
    g.bias = bias;
    g.resid = skip; g.ldr = h->Cp;
    g.up_src = up_src; g.Hu = Hu; g.Wu = Wu;
    g.out_f32 = out_f32; g.out_hi = out.hi; g.out_lo = out.lo; g.relu_bf16 = relu_bf16; g.ldc = h->Cp;
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
bool head_upsamples_bf16(const mdpt_handle* h) { return !h->x3c(CLS_HEAD) && (h->Cp & 7) == 0; }

int run_fusion(const Ctx& c, bool for_head = false) {
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
            CHK(rcu_conv(c, blk + ".conv_reassembly." + rcu_seq(h) + ".1", c.pl(p.r_bf[i]), sh[i], sw[i], nullptr, nullptr, 0, 0, nullptr, a1, 1));
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
    CHK(OPLC(mdpt_launch_upsample, c.at<float>(p.flo[0]), fu.hi, fu.lo, nullptr, p.B, sh[0], sw[0], 2 * sh[0], 2 * sw[0], h->Cp, c.s));
    return 0;
}

// ---- stage: head
// from_flo0b: the head's input is still the bf16 output of the last fusion projection at half resolution (run_fusion(c, true))
int run_head(const Ctx& c, void* depth, int depth_dtype = MDPT_DTYPE_F32, bool from_flo0b = false) {
    const mdpt_handle* h = c.h;
    const Plan& p = c.p;
    const int fh = 8 * p.gh, fw = 8 * p.gw;
    bool fused_ready = !from_flo0b;
    auto materialise_fused = [&]() -> int {  // stand-alone bf16 upsample (small launches / shapes the fused kernel does not cover)
        if (!fused_ready) CHK(OPLC(mdpt_launch_upsample_bf16, c.at<op_t>(p.flo[0]), c.pl(p.fused).hi, p.B, fh / 2, fw / 2, fh, fw, h->Cp, c.s));
        fused_ready = true;
        return 0;
    };
    if (!h->x3c(CLS_HEAD) && mdpt_head_tail_supported(h->C2p) && mdpt_head_tail_scale_ok(fh, fw, p.H, p.W)) {
        // bf16 mode: the first conv writes bf16 (the buffer of the fp32 map is reused), everything behind it is ONE kernel that keeps the
        // upsampled map in LDS tiles: upsample + 3x3 conv + ReLU + 1x1 conv + ReLU | sigmoid (head.hip). The bf16x3 mode keeps the
        // unfused form below (its hi + lo operand planes do not fit the LDS tile next to the weights).
        op_t* h1b = c.at<op_t>(p.h1);
        bool done = false;
        if (h->C2p == 128 && conv3h_shape_ok(h, fh, fw, h->Cp) && h->gemm_tile == MDPT_TILE_AUTO) {  // halo-staged form, 128 output channels
            Conv3hParams q;
            memset(&q, 0, sizeof(q));
            q.w = h->M("head.spatial_upsampler.0.weight").hi; q.bias = h->V("head.spatial_upsampler.0.bias");
            q.out_bf = h1b; q.B = p.B; q.H = fh; q.W = fw; q.Cin = h->Cp; q.Cout = 128;
            const long tiles256 = ((long)p.B * fh * fw + 255) / 256;
            const bool big = tiles256 >= conv3h_min_tiles(c);
#ifndef MDPT_NO_UPIN  // (A/B builds: -DMDPT_NO_UPIN keeps the stand-alone upsample in front of the halo-staged conv)
            if (big && !fused_ready) {  // the x2 upsample folded into the conv's halo interpolation
                q.up_in = c.at<op_t>(p.flo[0]); q.Hs = fh / 2; q.Ws = fw / 2;
                if (mdpt_conv3h_supported(q)) {
                    CHK(OPLC(mdpt_launch_conv3h, q, c.s));
                    done = true;
                }
                q.up_in = nullptr;
            }
#endif
            if (big && !done) {
                CHK(materialise_fused());
                q.in = c.pl(p.fused).hi;
                if (mdpt_conv3h_supported(q)) {
                    CHK(OPLC(mdpt_launch_conv3h, q, c.s));
                    done = true;
                }
            }
        }
        if (!done) {
            CHK(materialise_fused());
            GemmParams g = base_params(c, h->M("head.spatial_upsampler.0.weight"), c.pl(p.fused), p.B * fh * fw, h->Cp);
            as_conv(g, fh, fw, h->Cp, fh, fw, 1);
            g.bias = h->V("head.spatial_upsampler.0.bias");
            g.out_hi = h1b; g.ldc = h->C2p;
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
        }
        HeadTailParams t;
        memset(&t, 0, sizeof(t));
        t.src = h1b; t.w_kc = h->M("head.proj_1ch.0.weight@kc32").hi;
        t.bias = h->V("head.proj_1ch.0.bias"); t.head_w = h->V("head.proj_1ch.2.weight"); t.head_b = h->V("head.proj_1ch.2.bias");
        t.out = depth; t.out_dtype = depth_dtype; t.sigmoid = h->cfg.is_metric;
        t.B = p.B; t.Hi = fh; t.Wi = fw; t.Ho = p.H; t.Wo = p.W;
        CHK(OPLC(mdpt_launch_head_tail, t, h->C2p, c.s));
        return 0;
    }
    CHK(materialise_fused());
    {
        bool done = false;
        if (h->C2p == 128 && conv3h_shape_ok(h, fh, fw, h->Cp) && h->gemm_tile == MDPT_TILE_AUTO) {  // halo-staged form, fp32 map out
            const Mat& w1 = h->M("head.spatial_upsampler.0.weight");
            Planes fu = c.pl(p.fused);
            Conv3hParams q;
            memset(&q, 0, sizeof(q));
            q.in = fu.hi; q.in_lo = fu.lo; q.w = w1.hi; q.w_lo = w1.lo; q.bias = h->V("head.spatial_upsampler.0.bias");
            q.out_f32 = c.at<float>(p.h1); q.B = p.B; q.H = fh; q.W = fw; q.Cin = h->Cp; q.Cout = 128;
            const long tiles256 = ((long)p.B * fh * fw + 255) / 256;
            if (tiles256 >= conv3h_min_tiles(c) && mdpt_conv3h_supported(q)) {
                CHK(OPLC(mdpt_launch_conv3h, q, c.s));
                done = true;
            }
        }
        if (!done) {
            GemmParams g = base_params(c, h->M("head.spatial_upsampler.0.weight"), c.pl(p.fused), p.B * fh * fw, h->Cp);
            as_conv(g, fh, fw, h->Cp, fh, fw, 1);
            g.bias = h->V("head.spatial_upsampler.0.bias");
            g.out_f32 = c.at<float>(p.h1); g.ldc = h->C2p;
            CHK(OPLC(mdpt_launch_gemm, g, c.s));
        }
    }
    Planes hu = c.pl(p.h1u);
    CHK(OPLC(mdpt_launch_upsample, c.at<float>(p.h1), hu.hi, hu.lo, nullptr, p.B, fh, fw, p.H, p.W, h->C2p, c.s));
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

#include "mdpt_swin.inc"

int make_ctx(mdpt_handle* h, int B, int H, int W, void* ws, size_t ws_bytes, void* stream, Ctx* c) {
    Plan p;
    CHK(make_plan(h, B, H, W, &p));
    CHK(check_ws(h, p, ws, ws_bytes));
    c->h = h; c->p = p; c->ws = (char*)ws; c->s = (hipStream_t)stream;
    return 0;
}

}  // namespace

// =====================================================================================================================
// C ABI
// =====================================================================================================================
extern "C" {

int mdpt_abi_version(void) { return MDPT_ABI_VERSION; }
const char* mdpt_last_error(void) { return g_err.c_str(); }

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
        int32_t mixed[NCLS];
        mdpt_default_mixed_passes(mixed);
        const bool all3 = cfg->precision == MDPT_PREC_BF16X3 || cfg->precision == MDPT_PREC_FP16X3;
        for (int i = 0; i < NCLS; ++i) h->np[i] = cfg->precision == MDPT_PREC_MIXED ? mixed[i] : (all3 ? 3 : 1);
    }
    h->wrc_on = cfg->precision == MDPT_PREC_FP16 || cfg->precision == MDPT_PREC_MIXED;
    h->gemm_tile = MDPT_TILE_AUTO;
    h->finalized = false;
    h->has_last = false;
    h->zero_page = nullptr;
    h->dbg_block = h->dbg_step = -1;
    h->split_min = 8;
    h->side_stream = nullptr;
    h->ev_fork = h->ev_join = nullptr;
    build_inventory(h);
    *out = h;
    return 0;
}

void mdpt_destroy(mdpt_handle* h) { delete h; }

// MDPT_PREC_MIXED: which classes pay for three passes. From the per-class error budget (profiles/r04_precision_budget.md; the CPU
// emulation tests/precision_budget/emulate_operand_rounding.py reproduces it): the decoder's convs feed the depth map directly - no
// LayerNorm or residual stream between them and the output averages their operand rounding away - and carry ~85 % of the squared error.
void mdpt_default_mixed_passes(int32_t passes[MDPT_NUM_CLASSES]) {
    for (int i = 0; i < NCLS; ++i) passes[i] = 1;
    passes[CLS_PATCH] = 3;  // 0.13 % of the FLOPs
    passes[CLS_REASM] = 3;
    passes[CLS_FUSION] = 3;
    passes[CLS_HEAD] = 3;
    passes[CLS_FUSION_IN] = 1;  // 2 % of the decoder's squared error for a quarter of its FLOPs (profiles/r04_precision_budget.md)
}

int mdpt_get_class_passes(const mdpt_handle* h, int32_t op_class, int32_t* passes) {
    if (!h || !passes || op_class < 0 || op_class >= NCLS) return fail(MDPT_E_INVALID, "bad argument");
    *passes = h->np[op_class];
    return 0;
}

static void rebuild_inventory_keeping_bindings(mdpt_handle* h);

int mdpt_set_class_passes(mdpt_handle* h, int32_t op_class, int32_t passes) {
    if (!h || op_class < 0 || op_class >= NCLS) return fail(MDPT_E_INVALID, "bad op class %d", op_class);
    if (passes != 1 && passes != 3) return fail(MDPT_E_INVALID, "passes must be 1 or 3, got %d", passes);
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
    if (h->wrc_on == (on != 0)) return 0;
    h->wrc_on = on != 0;
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
        CHK(OPLH(mdpt_launch_pack_weight, sp.ptr, sp.dtype, m.hi, m.lo, m.kind, m.N, m.K, m.Np, m.Kp, m.ksz, st, src_ld, src_col0, rs ? rs->ptr : nullptr,
                                    rs ? rs->dtype : 0));
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
        CHK(OPLH(mdpt_launch_pad_copy_f32, vs.ptr, vs.dtype, v.ptr, v.n, v.np, st));
    }
    h->finalized = true;
    h->has_last = false;
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

int mdpt_set_latency_mode(mdpt_handle* h, int32_t on) {
    if (!h) return fail(MDPT_E_INVALID, "null handle");
    h->latency_mode = on ? 1 : 0;
    return 0;
}

int mdpt_set_gemm_tile(mdpt_handle* h, int32_t tile) {
    if (!h || tile < 0 || tile > 6 || tile == 3) return fail(MDPT_E_INVALID, "tile must be 0 (auto), 1 (128x128x64), 2 (256x256x64 lockstep), 4 (256x128x32) or 5 (256x256x32 ping-pong)");
    h->gemm_tile = tile;
    return 0;
}

static int forward_one(mdpt_handle* h, const Ctx& c, const void* image_bchw, int image_dtype, void* depth_bhw, int depth_dtype);

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
        if (!h->side_stream) {
            CHK(hipStreamCreateWithFlags(&h->side_stream, hipStreamNonBlocking));
            CHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
            CHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
        }
        hipStream_t s0 = (hipStream_t)stream;
        CHK(hipEventRecord(h->ev_fork, s0));
        CHK(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
        Ctx c0, c1;
        c0.h = h; c0.p = p0; c0.ws = (char*)workspace; c0.s = s0; c0.split = true;
        c1.h = h; c1.p = p1; c1.ws = (char*)workspace + off1; c1.s = h->side_stream; c1.split = true;
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
        return 0;
    }
    Ctx c;
    CHK(make_ctx(h, B, H, W, workspace, workspace_bytes, stream, &c));
    return forward_one(h, c, image_bchw, image_dtype, depth_bhw, depth_dtype);
}

static int forward_one(mdpt_handle* h, const Ctx& c, const void* image_bchw, int image_dtype, void* depth_bhw, int depth_dtype) {
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
        CHK(OPLC(mdpt_launch_tokens_import, (const float*)stage_in[i], tp.hi, tp.lo, B, p.N, p.npad, h->F, c.s));
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
        CHK(OPLC(mdpt_launch_nchw_to_nhwc, (const float*)maps_in[i], c.at<float>(p.r_f32[i]), rb.hi, rb.lo, 1, B, sh[i], sw[i], h->C, h->Cp, c.s));
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
    CHK(OPLC(mdpt_launch_nchw_to_nhwc, (const float*)reasm_in, c.at<float>(p.r_f32[i]), rb.hi, rb.lo, 1, B, sh, sw, h->C, h->Cp, c.s));
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
    CHK(OPLC(mdpt_launch_nchw_to_nhwc, (const float*)fused_in, nullptr, fu.hi, fu.lo, 0, B, 8 * gh, 8 * gw, h->C, h->Cp, c.s));
    CHK(run_head(c, (float*)depth_bhw));
    h->has_last = false;
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
        CHK(OPLC(mdpt_launch_tokens_export, tp.hi, tp.lo, nullptr, (float*)out_f32, p.B, p.N, p.npad, h->F, 0, c.s));
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
int mdpt_prepare_image(const void* bgr_u8_hwc, int32_t in_h, int32_t in_w, void* out_chw_f32, int32_t out_h, int32_t out_w,
                       const float rgb_mean[3], const float rgb_std[3], int32_t interpolation, void* stream) {
    if (!bgr_u8_hwc || !out_chw_f32 || !rgb_mean || !rgb_std) return fail(MDPT_E_INVALID, "null argument");
    if (interpolation != MDPT_INTERP_BILINEAR && interpolation != MDPT_INTERP_BICUBIC)
        return fail(MDPT_E_UNSUPPORTED, "interpolation %d: antialiased resize exists for bilinear and bicubic only (as in torch)", interpolation);
    if (in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return fail(MDPT_E_INVALID, "bad image size %dx%d -> %dx%d", in_h, in_w, out_h, out_w);
    const float inv_std[3] = {1.0f / rgb_std[0], 1.0f / rgb_std[1], 1.0f / rgb_std[2]};  // patch_embed.py:38-39,62
    CHK(mdpt_launch_prepare_image_bf16((const unsigned char*)bgr_u8_hwc, (float*)out_chw_f32, in_h, in_w, out_h, out_w, rgb_mean, inv_std, interpolation, (hipStream_t)stream));
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

// ---- test hooks (tests/ only): truncate the encoder after (block, step) and read raw internal buffers as fp32
int mdpt_debug_set_stop(mdpt_handle* h, int32_t block, int32_t step) {
    if (!h) return fail(MDPT_E_INVALID, "null handle");
    h->dbg_block = block; h->dbg_step = step;
    return 0;
}

int mdpt_debug_read(mdpt_handle* h, const char* name, void* out_f32, size_t out_floats, void* workspace, size_t workspace_bytes,
                    void* stream) {
    if (!h || !name || !out_f32) return fail(MDPT_E_INVALID, "null argument");
    if (!h->has_last) return fail(MDPT_E_STATE, "mdpt_debug_read needs a preceding mdpt_forward");
    if (h->swin) return fail(MDPT_E_UNSUPPORTED, "mdpt_debug_read: internal buffer names are defined for the ViT families only");
    Ctx c;
    c.h = h; c.p = h->last_plan; c.ws = (char*)workspace; c.s = (hipStream_t)stream;
    CHK(check_ws(h, c.p, workspace, workspace_bytes));
    const Plan& p = c.p;
    const size_t rows = (size_t)p.B * p.npad;
    const std::string n = name;
    const size_t* planes = nullptr;
    size_t f32_off = SIZE_MAX, elems = 0;
    size_t bf16_only[2] = {SIZE_MAX, SIZE_MAX};  // a bf16 map without a lo plane
    const bool bf16_head = !h->x3c(CLS_HEAD) && mdpt_head_tail_supported(h->C2p) && mdpt_head_tail_scale_ok(8 * p.gh, 8 * p.gw, p.H, p.W);
    const size_t px[4] = {(size_t)16 * p.Np, (size_t)4 * p.Np, (size_t)p.Np, (size_t)p.Np / 4};
    if (n == "im2col") { planes = p.im2col; elems = (size_t)p.B * p.Np * h->Kpatch; }
    else if (n == "pos") { f32_off = p.pos; elems = (size_t)p.Np * h->F; }
    else if (n == "resid") { f32_off = p.resid; elems = rows * h->F; }
    else if (n == "xn") { planes = p.xn; elems = rows * h->F; }
    else if (n == "q") { planes = p.q; elems = (size_t)p.B * h->heads * p.npad * 64; }
    else if (n == "k") { planes = p.k; elems = (size_t)p.B * h->heads * p.npad * 64; }
    else if (n == "vt") { planes = p.vt; elems = (size_t)p.B * h->heads * 64 * p.npadv; }
    else if (n == "att") { planes = p.att; elems = rows * h->F; }
    else if (n == "hbuf") { planes = p.hbuf; elems = rows * 4 * h->F; }
    else if (n == "h1") {
        // bf16 mode with the fused head tail: the first conv writes a bf16 map into the fp32 map's buffer (run_head)
        elems = (size_t)p.B * 64 * p.Np * h->C2p;
        if (bf16_head) { bf16_only[0] = p.h1; planes = bf16_only; } else { f32_off = p.h1; }
    }
    else if (n == "h1u") {
        if (bf16_head) return fail(MDPT_E_STATE, "h1u does not exist on the fused head-tail path (the upsampled map only ever lives in LDS tiles)");
        planes = p.h1u; elems = (size_t)p.B * p.H * p.W * h->C2p;
    }
    else if (n == "fused") {
        // bf16 mode: the forward may have folded the x2 upsample into the head's first conv; rebuild the map the head saw (same arithmetic)
        if (head_upsamples_bf16(h))
            CHK(OPLC(mdpt_launch_upsample_bf16, c.at<op_t>(p.flo[0]), c.pl(p.fused).hi, p.B, 4 * p.gh, 4 * p.gw, 8 * p.gh, 8 * p.gw, h->Cp, c.s));
        planes = p.fused; elems = (size_t)p.B * 64 * p.Np * h->Cp;
    }
    else if (n == "u0") { planes = p.u0; elems = (size_t)p.B * px[0] * h->hidp[0]; }
    else if (n == "u1") { planes = p.u1; elems = (size_t)p.B * px[1] * h->hidp[1]; }
    else if (n == "d3") { planes = p.d3; elems = (size_t)p.B * px[3] * h->hidp[3]; }
    else if (n.size() == 2 && n[0] == 't' && n[1] >= '0' && n[1] <= '3') { const int i = n[1] - '0'; planes = p.t[i]; elems = (size_t)p.B * p.Np * h->hidp[i]; }
    else if (n.size() == 4 && n.compare(0, 3, "flo") == 0 && n[3] >= '0' && n[3] <= '3') {
        const int i = n[3] - '0';
        elems = (size_t)p.B * px[i] * h->Cp;
        if (i == 0 && head_upsamples_bf16(h)) { bf16_only[0] = p.flo[0]; planes = bf16_only; } else { f32_off = p.flo[i]; }  // level 0: bf16 map (run_fusion(c, true))
    }
    else if (n.size() == 3 && n.compare(0, 2, "xf") == 0 && n[2] >= '0' && n[2] <= '3') { const int i = n[2] - '0'; f32_off = p.x_f32[i]; elems = (size_t)p.B * px[i] * h->Cp; }
    else if (n.size() == 3 && n.compare(0, 2, "a1") == 0 && n[2] >= '0' && n[2] <= '3') { const int i = n[2] - '0'; planes = p.a1[i]; elems = (size_t)p.B * px[i] * h->Cp; }
    else if (n.size() == 3 && n.compare(0, 2, "b2") == 0 && n[2] >= '0' && n[2] <= '3') { const int i = n[2] - '0'; planes = p.b2[i]; elems = (size_t)p.B * px[i] * h->Cp; }
    else return fail(MDPT_E_INVALID, "unknown debug buffer \"%s\"", name);
    if (out_floats < elems) return fail(MDPT_E_WORKSPACE, "debug buffer %s needs %zu floats, got %zu", name, elems, out_floats);
    // reuse the token exporter as a flat converter: B=1, N=npad=elems/F' with F'=4 keeps indices simple
    if (planes) {
        Planes pl = c.pl(planes);
        CHK(OPLC(mdpt_launch_tokens_export, pl.hi, pl.lo, nullptr, (float*)out_f32, 1, (int)(elems / 4), (int)(elems / 4), 4, 0, c.s));
    } else {
        CHK(OPLC(mdpt_launch_tokens_export, nullptr, nullptr, c.at<float>(f32_off), (float*)out_f32, 1, (int)(elems / 4), (int)(elems / 4), 4, 0, c.s));
    }
    return 0;
}

// ---- test/bench hook: the plain dense GEMM kernel on caller-provided bf16 operands (out_f32[M,N] = A[M,K] W[N,K]^T)
int mdpt_debug_gemm(const void* a_bf16, const void* w_bf16, void* out_f32, void* out_bf16, int32_t M, int32_t N, int32_t K,
                    int32_t tile, int32_t iters, void* stream, void* dbg_times) {
    if (!a_bf16 || !w_bf16 || (!out_f32 && !out_bf16)) return fail(MDPT_E_INVALID, "null argument");
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.A_hi = (const op_t*)a_bf16; g.W_hi = (const op_t*)w_bf16;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.npass = 1;
    g.zero_page = (const op_t*)w_bf16;
    g.amode = MDPT_A_DENSE; g.ekind = MDPT_E_GENERIC; g.tile = tile & 255;
    g.act = (tile >> 8) & 3;  // bits 8-9 of `tile`: epilogue activation (MDPT_ACT_*), for epilogue-cost measurements
    g.out_f32 = (float*)out_f32; g.out_hi = (op_t*)out_bf16; g.ldc = N; g.ldr = N;
    g.dbg_times = (unsigned long long*)dbg_times;
    if ((tile >> 10) & 1) {  // bit 10: in-place residual epilogue (proj / fc2 form); bias and gamma are read from the out_bf16 buffer
        if (!out_f32 || !out_bf16) return fail(MDPT_E_INVALID, "residual mode needs both output buffers");
        g.bias = (const float*)out_bf16; g.gamma = (const float*)out_bf16 + N; g.resid = (const float*)out_f32; g.out_hi = nullptr;
    }
    if ((tile >> 11) & 1) {  // bit 11: residual-initialised accumulators (the encoder's proj / fc2 form): out = (out + A W^T) + bias, in place
        if (!out_f32 || !out_bf16) return fail(MDPT_E_INVALID, "residual mode needs both output buffers");
        g.bias = (const float*)out_bf16; g.resid = (const float*)out_f32; g.out_hi = nullptr; g.acc_init = 1;
    }
    for (int i = 0; i < iters; ++i) CHK(OPLG(mdpt_launch_gemm, g, (hipStream_t)stream));
    return 0;
}

// ---- test/bench hook: one 3x3 stride-1 conv Cin -> Cout (256 | 128) on caller-provided operands (bf16 NHWC input, MDPT_PACK_CONV3 weights [Cout][9 Cin]):
//      path 0 = the implicit-GEMM kernels of gemm.hip (tile = MDPT_TILE_*), path 1 = the halo-staged kernel of conv3h.hip.
//      out = [skip +] conv + [bias] [+ up2(up)] -> out_f32 (optional) and out_bf16 (ReLU'd if relu_bf16); both paths use the same arithmetic
int mdpt_debug_conv3(const void* in_bf16, const void* w_packed_bf16, const void* bias_f32, const void* skip_f32, const void* up_f32, int32_t Hu,
                     int32_t Wu, void* out_f32, void* out_bf16, int32_t relu_bf16, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                     int32_t path, int32_t tile, int32_t iters, void* stream, void* dbg_times, const void* in_lo_bf16, const void* w_lo_bf16,
                     void* out_lo_bf16) {
    if (Cout != 256 && Cout != 128) return fail(MDPT_E_INVALID, "Cout must be 256 or 128");
    if ((in_lo_bf16 != nullptr) != (w_lo_bf16 != nullptr)) return fail(MDPT_E_INVALID, "bf16x3 needs the lo planes of input and weights");
    if (!in_bf16 || !w_packed_bf16 || (!out_bf16 && !out_f32)) return fail(MDPT_E_INVALID, "null argument");
    static op_t* zero_page = nullptr;  // test hook only: allocated once, never freed
    if (!zero_page) {
        if (hipMalloc((void**)&zero_page, 256) != hipSuccess || hipMemset(zero_page, 0, 256) != hipSuccess) return fail(MDPT_E_STATE, "zero page allocation failed");
    }
    if (path == 2 || path == 3) {
        // upsampled input: in_bf16 is the SOURCE map [B, Hu, Wu, Cin]; the conv runs on its bilinear upsample to H x W.
        // path 2 = interpolated inside the halo-staged kernel; path 3 = stand-alone bf16 upsample into out_lo_bf16 (scratch [B, H, W, Cin])
        // followed by the implicit-GEMM conv (tile) - the two must agree bit for bit
        if (Cout != 128 || !out_bf16 || Hu < 2 || Wu < 2) return fail(MDPT_E_INVALID, "upsampled-input form: 128 output channels, bf16 output");
        if (path == 2) {
            Conv3hParams q;
            memset(&q, 0, sizeof(q));
            q.up_in = (const op_t*)in_bf16; q.Hs = Hu; q.Ws = Wu; q.w = (const op_t*)w_packed_bf16; q.bias = (const float*)bias_f32;
            q.out_bf = (op_t*)out_bf16; q.B = B; q.H = H; q.W = W; q.Cin = Cin; q.Cout = 128;
            q.dbg_times = (unsigned long long*)dbg_times;
            if (!mdpt_conv3h_supported(q)) return fail(MDPT_E_UNSUPPORTED, "conv3h does not cover this combination");
            for (int i = 0; i < iters; ++i) CHK(OPLG(mdpt_launch_conv3h, q, (hipStream_t)stream));
            return 0;
        }
        if (!out_lo_bf16) return fail(MDPT_E_INVALID, "path 3 needs a scratch map in out_lo_bf16");
        for (int i = 0; i < iters; ++i) {
            CHK(OPLG(mdpt_launch_upsample_bf16, (const op_t*)in_bf16, (op_t*)out_lo_bf16, B, Hu, Wu, H, W, Cin, (hipStream_t)stream));
            GemmParams g;
            memset(&g, 0, sizeof(g));
            g.A_hi = (const op_t*)out_lo_bf16; g.W_hi = (const op_t*)w_packed_bf16;
            g.M = B * H * W; g.N = 128; g.K = 9 * Cin; g.lda = Cin; g.npass = 1; g.zero_page = zero_page;
            g.amode = MDPT_A_CONV3; g.ekind = MDPT_E_GENERIC; g.tile = tile;
            g.Hi = H; g.Wi = W; g.Cin = Cin; g.Ho = H; g.Wo = W; g.cstride = 1;
            g.bias = (const float*)bias_f32; g.out_hi = (op_t*)out_bf16; g.ldc = 128; g.ldr = 128;
            CHK(OPLG(mdpt_launch_gemm, g, (hipStream_t)stream));
        }
        return 0;
    }
    if (path == 1) {
        Conv3hParams q;
        memset(&q, 0, sizeof(q));
        q.in = (const op_t*)in_bf16; q.w = (const op_t*)w_packed_bf16; q.bias = (const float*)bias_f32; q.skip = (const float*)skip_f32;
        q.in_lo = (const op_t*)in_lo_bf16; q.w_lo = (const op_t*)w_lo_bf16; q.out_bf_lo = (op_t*)out_lo_bf16;
        q.up_src = (const float*)up_f32; q.Hu = Hu; q.Wu = Wu; q.out_f32 = (float*)out_f32; q.out_bf = (op_t*)out_bf16; q.relu_bf = relu_bf16;
        q.B = B; q.H = H; q.W = W; q.Cin = Cin; q.Cout = Cout;
        q.dbg_times = (unsigned long long*)dbg_times;
        if (!mdpt_conv3h_supported(q)) return fail(MDPT_E_UNSUPPORTED, "conv3h does not cover this combination");
        for (int i = 0; i < iters; ++i) CHK(OPLG(mdpt_launch_conv3h, q, (hipStream_t)stream));
        return 0;
    }
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.A_hi = (const op_t*)in_bf16; g.W_hi = (const op_t*)w_packed_bf16; g.A_lo = (const op_t*)in_lo_bf16; g.W_lo = (const op_t*)w_lo_bf16;
    g.M = B * H * W; g.N = Cout; g.K = 9 * Cin; g.lda = Cin; g.npass = in_lo_bf16 ? 3 : 1;
    g.zero_page = zero_page;
    g.amode = MDPT_A_CONV3; g.ekind = MDPT_E_GENERIC; g.tile = tile;
    g.Hi = H; g.Wi = W; g.Cin = Cin; g.Ho = H; g.Wo = W; g.cstride = 1;
    g.bias = (const float*)bias_f32; g.resid = (const float*)skip_f32; g.ldr = Cout;
    g.up_src = (const float*)up_f32; g.Hu = Hu; g.Wu = Wu;
    g.out_f32 = (float*)out_f32; g.out_hi = (op_t*)out_bf16; g.out_lo = (op_t*)out_lo_bf16; g.relu_bf16 = relu_bf16; g.ldc = Cout;
    g.dbg_times = (unsigned long long*)dbg_times;
    for (int i = 0; i < iters; ++i) CHK(OPLG(mdpt_launch_gemm, g, (hipStream_t)stream));
    return 0;
}

// ---- test hook: the fused attention kernel on caller-provided head-major operands (bf16 mode, head dim 64, no bias):
//      Q, K [B, heads, npad, 64] (Q pre-scaled by 1/8), Vt [B, heads, 64, npadv] (pad columns zero) -> out [B * npad, heads * 64]
int mdpt_debug_attention(const void* q_bf16, const void* k_bf16, const void* vt_bf16, void* out_bf16, int32_t B, int32_t heads, int32_t N,
                         int32_t npad, int32_t npadv, int32_t iters, void* stream) {
    if (!q_bf16 || !k_bf16 || !vt_bf16 || !out_bf16) return fail(MDPT_E_INVALID, "null argument");
    AttnParams a;
    memset(&a, 0, sizeof(a));
    a.q_hi = (const op_t*)q_bf16; a.k_hi = (const op_t*)k_bf16; a.vt_hi = (const op_t*)vt_bf16; a.out_hi = (op_t*)out_bf16;
    a.B = B; a.heads = heads; a.N = N; a.npad = npad; a.npadv = npadv; a.F = heads * 64;
    for (int i = 0; i < iters; ++i) CHK(OPLG(mdpt_launch_attention, a, (hipStream_t)stream));
    return 0;
}

// ---- RCCL all-gather wrapper (resolved lazily so the library itself has no link-time dependency on RCCL)
int mdpt_allgather(void* comm, const void* send_dev, void* recv_dev, size_t count_per_rank, int32_t dtype, void* stream) {
    typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, void*);
    static allgather_fn fn = nullptr;
    if (!fn) {
        void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return fail(MDPT_E_STATE, "cannot load librccl.so: %s", dlerror());
        fn = (allgather_fn)dlsym(lib, "ncclAllGather");
        if (!fn) return fail(MDPT_E_STATE, "ncclAllGather not found in librccl.so");
    }
    // ncclDataType_t: ncclFloat16 = 6, ncclFloat32 = 7, ncclBfloat16 = 9 (rccl.h)
    if (dtype != MDPT_DTYPE_F32 && dtype != MDPT_DTYPE_BF16 && dtype != MDPT_DTYPE_F16) return fail(MDPT_E_INVALID, "bad dtype %d", dtype);
    const int nccl_type = dtype == MDPT_DTYPE_F32 ? 7 : (dtype == MDPT_DTYPE_BF16 ? 9 : 6);
    const int rc = fn(send_dev, recv_dev, count_per_rank, nccl_type, comm, stream);
    if (rc != 0) return fail(MDPT_E_STATE, "ncclAllGather failed with code %d", rc);
    return 0;
}

}  // extern "C"

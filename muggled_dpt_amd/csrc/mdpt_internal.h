// libmdpt internals shared by the host-side translation units (not part of the C ABI, which is include/mdpt.h):
//   mdpt_inventory.cpp  parameter inventory (reference key names), packed-weight layout, activation workspace plan
//   mdpt_stages.cpp     launch sequences of the five stages (patch embed, encoder, reassemble, fusion, head) for all families
//   mdpt_api.cpp        the C entry points
//   mdpt_debug.cpp      test / measurement hooks and the RCCL wrapper
//
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

#pragma once
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
// host-only predicates of the kernel files with the same answer in both builds (mdpt_conv3h_supported is NOT one of them: the fp16 build
// covers more output-plane combinations - call it through OPLC / OPLG)
#define mdpt_head_tail_supported mdpt_head_tail_supported_bf16
#define mdpt_head_tail_scale_ok mdpt_head_tail_scale_ok_bf16
#define mdpt_beit_relpos_elen mdpt_beit_relpos_elen_bf16

namespace mdpt {

extern thread_local std::string g_err;
extern int g_debug_f16;

int fail(int code, const char* fmt, ...);  // records the message for mdpt_last_error(), returns `code`

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

// Op classes: every contraction of the path belongs to one; a class runs 1 MFMA pass (operands rounded to one 16-bit plane), 3 (hi + lo
// split planes of both operands) or 2 (activations split, weights one plane). The uniform modes set all classes alike; MDPT_PREC_MIXED /
// mdpt_set_class_passes choose per class (include/mdpt.h).
enum { CLS_PATCH = MDPT_CLASS_PATCH, CLS_QKV = MDPT_CLASS_QKV, CLS_ATTN = MDPT_CLASS_ATTN, CLS_PROJ = MDPT_CLASS_PROJ, CLS_FC1 = MDPT_CLASS_FC1,
       CLS_FC2 = MDPT_CLASS_FC2, CLS_REASM = MDPT_CLASS_REASM, CLS_FUSION = MDPT_CLASS_FUSION, CLS_HEAD = MDPT_CLASS_HEAD,
       CLS_FUSION_IN = MDPT_CLASS_FUSION_IN, CLS_HEAD_TAIL = MDPT_CLASS_HEAD_TAIL, CLS_FUSION_PROJ = MDPT_CLASS_FUSION_PROJ, NCLS = MDPT_NUM_CLASSES };

int mat_class(const std::string& src);

struct Mat {  // packed operand panel [Np][Kp]
    std::string src;
    int cls;  // CLS_*
    std::string row_scale;  // name of a per-output-feature fp32 parameter folded into the rows at pack time ("" = none)
    int kind, N, K, Np, Kp, ksz;
    size_t off_hi, off_lo;
    op_t* hi;
    op_t* lo;
    size_t off_scale;      // fp16 handles, layer-scale-folded matrices: 2 floats {s, 1 / s} in the packed buffer (SIZE_MAX = none), GemmParams::wscale
    const float* wscale;
    // fp8 cross-term planes of an F8 class (f8_cross.h; SIZE_MAX = none): [Np][Kp] e4m3 bytes of W_hi (K order of the 128-wide fp8 K tiles),
    // of W - W_hi (3 terms only), and the per-row E8M0 scale bytes of both ([Np] each, the second with the activations' 2^16 folded in)
    size_t off_w8, off_wlo8, off_s8;
    const uint8_t* w8; const uint8_t* wlo8; const uint8_t* s8; const uint8_t* slo8;
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
    size_t f8 = 0;  // != 0: the CONSUMING class runs its cross terms on fp8 planes (f8_cross.h) and `lo` holds BYTES - the e5m2 residue plane
                    // [elems], then (consumers with three terms) the e5m2 plane of the values themselves at byte offset f8 = elems
    int f8_a8 = 0;  // the second plane is wanted
};

extern const char* const kStageNames[4];
extern const char* const kSwinStageNames[4];
constexpr int kCpbHidden = 512;  // SwinV2 position-bias MLP width, relative_positional_encoder.py:31

// activation workspace layout for one (B, H, W)
struct Plan {
    int B, H, W, gh, gw, Np, N, npad, npadv;
    size_t total;
    // byte offsets (SIZE_MAX = absent)
    // operand planes: [0] hi, [1] lo (SIZE_MAX = none), [2] fp8 form of the lo plane: 0 = 16-bit residue plane, else element count | (1 << 63 if the
    // consumer also reads the e5m2 plane of the values), see Planes
    size_t im2col[3], pos, resid, xn[3], q[3], k[3], vt[3], att[3], hbuf[3], tap[4][3], tapf32;
    size_t t[4][3], u0[3], u1[3], d3[3];
    size_t r_f32[4], r_bf[4][3];
    size_t a1[4][3], x_f32[4], x_bf[4][3], b1[4][3], b2[4][3], flo[4];
    size_t fused[3], h1, h1u[3], scratch;
    size_t scratch_floats;
    size_t tokr[3], cbuf, relpos_lut, relpos_tq, relpos_tk;  // BEiT: readout-projected tokens, per-image cls term, bias LUT
    size_t probe;                                             // 256 bytes of their own for the side-stream probe
    size_t poison;                                            // one word per image: the image holds a NaN / inf (mdpt_forward: its depth map becomes NaN)
    size_t wrc_mean, wrc_tab;                                 // [B, wrc_maxk] operand-format column means, fp32 [B, wrc_maxn] per-image bias table
    size_t swi;                                               // ViT-G: fp32 [rows, 2*hidden] output of the doubled inner linear
    size_t kspart;                                            // small batches: 3 x fp32 [rows, F] partial sums of the K-split proj / fc2 (latency mode), else absent
    // SwinV2: stage-0 patch grid, per-stage residual streams (fp32, = the taps), shared GEMM fp32 output, token planes,
    // window operands, window maps (plain / shifted) and the position-bias LUT
    struct {
        int g0h, g0w;
        size_t resid[4], x, xn[3], q[3], k[3], vt[3], att[3], hb[3], lut, lut_stride, tq, tk, rowmap[2], region[2], tokmap[2], vtokmap[2];
    } sw;
};

}  // namespace mdpt

using namespace mdpt;

struct mdpt_handle {
    mdpt_config cfg;
    int F, heads, nblocks, bps, P, C, Cp, C2, C2p, Kpatch;
    int hid[4], hidp[4];
    bool swin;
    int Pv;  // patch size seen by fusion/head: the finest reassembly map is (4H/Pv) x (4W/Pv); = P except SwinV2 (16)
    int gh_hidden, gh_hidden_p;  // ViT-G SwiGLU hidden width (and padded to 64), 0 otherwise
    int sH[4], sL[4], swh, sww, spre[4];  // SwinV2: heads / layers per stage, target window, pretrained window sizes (0 = None)
    bool f16;       // operand format: fp16 (v_mfma_*_f16, *_f16 launchers) instead of bf16
    int np[NCLS];   // per op class as set (mdpt_set_class_passes): 1, 2 (activations split), 3, MDPT_PASSES_2F8, MDPT_PASSES_3F8
    bool f8ok[NCLS];  // the class CAN run its cross terms on fp8 planes (fp16 operands, every contraction length a multiple of 128, not SwinV2): build_inventory
    int terms(int cls) const { return np[cls] >= MDPT_PASSES_2F8 ? np[cls] - 2 : np[cls]; }  // products per contraction: 1, 2 or 3
    bool f8(int cls) const { return np[cls] >= MDPT_PASSES_2F8 && f8ok[cls]; }  // cross terms on fp8 planes (f8_cross.h)
    bool x3c(int cls) const { return terms(cls) == 3 && !f8(cls); }   // the class's WEIGHTS have a 16-bit lo plane
    bool alo(int cls) const { return terms(cls) >= 2; }   // the class reads a lo plane of its ACTIVATIONS (its producers write one)
    // token-mean compensation of the weight rounding (fp16 operand modes, single-pass encoder Linears): see wrc_bias() below
    bool wrc_on;
    int wrc_mask;  // bit (1 << class): which of the four encoder Linear classes are compensated (mdpt_set_weight_rounding_compensation)
    bool wrc(int cls) const { return wrc_on && f16 && np[cls] == 1 && (cls == CLS_QKV || cls == CLS_PROJ || cls == CLS_FC1 || cls == CLS_FC2) && ((wrc_mask >> cls) & 1); }
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
    Plan stage_plan;      // test hook: plan of the last stage-level mdpt_head call (mdpt_debug_read of its buffers)
    bool has_stage_plan = false;
    int dbg_block, dbg_step;  // test hook: stop the encoder after (block, step); -1 = off
    // batch split: batches >= split_min run as two halves on the caller's stream and an internal side stream (fork / join with
    // events, no host sync) so that one half's kernels fill the tile-quantisation tails and epilogue phases of the other's
    int split_min;
    int nonfinite_prop = 1;  // mdpt_set_nonfinite_propagation: an image tensor with a NaN / inf gives a NaN depth map (as the reference does), in every mode
    int wscale_all = 0;  // test policy (mdpt_debug_set_wscale_policy): scale EVERY layer-scale-folded matrix of the fp16 build, not only those below 2^-5
    int latency_mode;  // mdpt_set_latency_mode: small launches may use summation orders that are not batch-invariant
    int side_prio;      // priority class of the side stream: 0 = the default class (default), 1 = lowest, -1 = highest (mdpt_debug_set_side_stream_priority; measured worse)
    int overlap_reasm;  // unsplit forwards queue the reassembly branches on the side stream beside the encoder: 0 never, 1 = rule in forward_one (default), 2 always
    // mdpt_set_grid_cache (the reference's enable_cache, position_encoder.py:152-227): per-grid constants - resized position embedding, BEiT's
    // relative-position tables, SwinV2's position-bias tables, the zero pads of operand planes - stay in the workspace of the last forward of
    // a (workspace, B, H, W) and are not recomputed by the next forward on the same workspace and shape (`gen` = finalize generation)
    int grid_cache;
    uint64_t gen;
    struct CacheSlot { const void* ws; int B, H, W; uint64_t gen; bool valid; } cache_slot[2];
    bool cache_hit(const void* ws, int B, int H, int W) const {
        if (!grid_cache) return false;
        for (const CacheSlot& s : cache_slot)
            if (s.valid && s.ws == ws && s.B == B && s.H == H && s.W == W && s.gen == gen) return true;
        return false;
    }
    void cache_store(int slot, const void* ws, int B, int H, int W) { cache_slot[slot] = CacheSlot{ws, B, H, W, gen, true}; }
    void cache_clear() { cache_slot[0].valid = cache_slot[1].valid = false; }
    int ks_min_ktiles, ks_big_ktiles;  // ... proj / fc2 split K in two (64x64 tile) from ks_min K tiles on, in four from ks_big on (mdpt_debug_set_ksplit_min)
    // the side stream: chosen among up to four candidates as one that really runs beside the caller's stream (stream_probe.hip), per caller stream
    hipStream_t side_stream, side_cand[4], side_for[4];  // side_for[k] -> side_cand[side_pick[k]]: the caller streams probed so far (ring of four)
    int side_pick[4], side_nfor;
    int side_unresolved;  // probes in which every candidate was rejected (a busy GPU can starve the setter): the choice is not pinned, the stream is probed again (three times at most)
    int side_ncand, side_rejected;  // candidates created; candidates found on the caller's hardware queue so far (mdpt_debug_side_stream_info)
    int side_probe;                 // 1 (default): probe; 0: take the first candidate unseen (mdpt_debug_set_side_stream_probe)
    hipEvent_t ev_fork, ev_join;
    ~mdpt_handle() {
        for (int i = 0; i < side_ncand; ++i) hipStreamDestroy(side_cand[i]);
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
        m.off_scale = SIZE_MAX; m.wscale = nullptr;
        m.off_w8 = m.off_wlo8 = m.off_s8 = SIZE_MAX;
        m.w8 = m.wlo8 = m.s8 = m.slo8 = nullptr;
        if (f8(m.cls) && kind != MDPT_PACK_CONV3_KC32) {
            m.off_w8 = packed_total; packed_total += rup256((size_t)Np * Kp);
            if (terms(m.cls) == 3) { m.off_wlo8 = packed_total; packed_total += rup256((size_t)Np * Kp); }
            m.off_s8 = packed_total; packed_total += rup256((size_t)Np * 2);
        }
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

namespace mdpt {

struct Bump {
    size_t off = 0;
    size_t take(size_t bytes) {
        const size_t o = off;
        off += rup256(bytes);
        return o;
    }
};

struct Ctx {
    const mdpt_handle* h;
    Plan p;
    char* ws;
    hipStream_t s;
    bool split = false;  // this context is one half of a two-stream batch split
    hipStream_t tap_stream = nullptr;  // small-batch forward: every reassembly branch is queued here as soon as its tap exists (tap_event orders it)
    hipEvent_t tap_event = nullptr;
    bool side = false;  // this context runs on that side stream, beside the encoder
    bool a1_done = false;  // ... where the first conv of every conv_reassembly unit was queued too: run_fusion skips it
    unsigned* poison = nullptr;  // mdpt_forward with non-finite propagation on: the plan's per-image words, cleared, for the im2col kernel to set
    bool consts_cached = false;  // the per-grid constants of this (workspace, shape) are in place (mdpt_set_grid_cache): skip the kernels that write them
    // mdpt_forward_bgr: the patch embedding's im2col kernel builds its rows from this uint8 BGR image (resize + normalise fused in) instead of an image tensor
    struct BgrSource { const unsigned char* ptr = nullptr; int ih = 0, iw = 0, round_dtype = 0, interp = 0; float mean[3] = {0, 0, 0}, inv_std[3] = {1, 1, 1}; } bgr;
    void* const* attn_dump = nullptr;  // per block: where to write softmax(q k^T) as fp32 [B,H,N,N] (null entries: skip)
    void* const* block_dump = nullptr; // per block: where to write the block's output tokens as fp32 [B,N,F] (null entries: skip)
    template <class T> T* at(size_t off) const { return off == SIZE_MAX ? nullptr : (T*)(ws + off); }
    Planes pl(const size_t o[3]) const {
        Planes r;
        r.hi = at<op_t>(o[0]);
        r.lo = at<op_t>(o[1]);
        r.f8 = o[1] == SIZE_MAX ? 0 : (o[2] & ~((size_t)1 << 63));
        r.f8_a8 = o[1] != SIZE_MAX && (o[2] >> 63);
        return r;
    }
};

struct SwinStageGeom {
    int gh, gw, N, F, heads, wh, ww, sh, sw, nw, wa, npad, npadv;
};

// ---- mdpt_inventory.cpp
std::string blk_name(const mdpt_handle* h, int block);
inline bool is_beit(const mdpt_handle* h) { return h->cfg.family == MDPT_FAMILY_BEIT; }
inline bool is_midas(const mdpt_handle* h) { return h->cfg.family == MDPT_FAMILY_BEIT || h->cfg.family == MDPT_FAMILY_SWINV2; }
// reference attribute names differ between the families (v2: fusion_model.py:100,138 / v31_beit, v31_swinv2 fusion_model.py)
inline const char* rcu_seq(const mdpt_handle* h) { return is_midas(h) ? "conv_seq" : "resconv_seq"; }
inline const char* proj_seq(const mdpt_handle* h) { return is_midas(h) ? "proj_seq" : "scale_proj_seq"; }
void compute_f8ok(mdpt_handle* h);
int build_inventory(mdpt_handle* h);
int make_plan(const mdpt_handle* h, int B, int H, int W, Plan* pl);
int check_ws(const mdpt_handle* h, const Plan& p, const void* ws, size_t bytes);
int make_ctx(mdpt_handle* h, int B, int H, int W, void* ws, size_t ws_bytes, void* stream, Ctx* c);
int swin_geom(const mdpt_handle* h, int g0h, int g0w, int s, SwinStageGeom* g);
std::string swin_blk(int s, int l);

// ---- mdpt_stages.cpp
GemmParams base_params(const Ctx& c, const Mat& w, Planes a, int M, int lda);
void as_conv(GemmParams& g, int Hi, int Wi, int Cin, int Ho, int Wo, int stride);
int run_pos(const Ctx& c);
int run_patchify(const Ctx& c, const void* image, int image_dtype, const Planes& im, int H, int W);
int run_patch_embed_fused(const Ctx& c, const void* image, int image_dtype);
int wrc_bias(const Ctx& c, GemmParams& g, const Mat& w, const float* bias, int rows_per_img = 0, int nreal = 0);
bool fc2_ksplit_fits(int rows, int F);  // batch small enough for the K-split form of fc2 (the 64x64 tile's range): the plan then holds kspart
int run_encoder(const Ctx& c, void* const taps_f32[4]);
int run_reassemble(const Ctx& c);
int run_reassemble_stage(const Ctx& c, int i);
int rcu_conv(const Ctx& c, const std::string& wname, Planes in, int sh, int sw, const float* skip, const float* up_src, int Hu, int Wu,
             float* out_f32, Planes out, int relu_bf16);
bool head_upsamples_bf16(const mdpt_handle* h);
bool head_tail_fused(const mdpt_handle* h);
int run_fusion(const Ctx& c, bool for_head = false);
int fusion_rcu_a_first(const Ctx& c, int i);
int run_head(const Ctx& c, void* depth, int depth_dtype = MDPT_DTYPE_F32, bool from_flo0b = false);
int swin_zero_pad_planes(const Ctx& c, int rows0);
int run_patch_embed_swin(const Ctx& c, const void* image, int image_dtype, float* tokens_out);
int run_encoder_swin(const Ctx& c, void* const taps_f32[4]);
int run_reassemble_swin(const Ctx& c);

}  // namespace mdpt

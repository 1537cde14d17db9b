/* libmdpt - C ABI of the MI355X-native DPT depth-inference path (Depth-Anything-V2 family).
 *
 * The reference (heyoeyo/muggled_dpt) has NO FFI / operator / plugin layer: its boundary is the Python API
 *   make_dpt_from_state_dict()            muggled_dpt/make_dpt.py:21-72
 *   DPTModel.forward / .inference         muggled_dpt/dpt_model.py:61-83, :87-109
 *   model.patch_embed / .imgencoder / .reassemble / .fusion / .head   (dpt_model.py:50-54; called one by one in
 *                                          simple_examples/internal_features.py:38-45)
 * This header is therefore the NEW boundary a binding of that API sits on: plain pointers and
 * sizes, no torch types. Every entry point names the reference call it replaces.
 *
 * Conventions
 *   - return value: 0 = OK, otherwise a negative MDPT_E_* code or a positive hipError_t; mdpt_last_error() has text.
 *   - all `dev` pointers are device (HBM) pointers owned by the caller (PyTorch allocates them in our binding);
 *     the library allocates no device memory and does not synchronise with the host (one exception: the probe below).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream). Every kernel is ordered on the caller's stream. ONE
 *     exception, stated here because it is visible to tools: mdpt_forward lazily creates an internal non-blocking side stream + two events
 *     per handle. The first forward on a caller stream picks that side stream among up to four candidates by PROBING, on the GPU, that it
 *     runs beside the caller's stream and not behind it on a shared hardware queue (a few tiny launches and one host wait, once per handle
 *     and caller stream, skipped inside a stream capture; mdpt_debug_set_side_stream_probe). For batches >= 8 (mdpt_set_batch_split) the
 *     second half batch runs there; for smaller batches of the wide encoders (feature width >= 1024, default mode) the four reassembly
 *     branches do, each as soon as its encoder tap exists (mdpt_debug_set_reassemble_overlap). Either way the side stream is
 *     forked from and joined back into the caller's stream with events before the call returns (also on the error path) - stream-ordering
 *     semantics for the caller are unchanged, the call stays capturable into a hipGraph, results are bit-identical to the one-stream form.
 *   - tensors at this boundary use the REFERENCE layouts: images/maps BCHW, tokens B x N x F, depth B x H x W. The hot entry points
 *     (mdpt_bind_weight, mdpt_forward, mdpt_allgather) take a dtype tag per tensor (fp32, bf16 or fp16: whatever the caller's
 *     model dtype is - no cast kernels at the boundary); the stage-level / debug entry points are fp32.
 *   - a handle is not thread-safe; distinct handles are independent.
 */
#ifndef MDPT_H
#define MDPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDPT_ABI_VERSION 6

/* arithmetic modes (all accumulate in fp32; residual stream, LayerNorm and softmax statistics are fp32) */
#define MDPT_PREC_BF16 0   /* bf16 MFMA operands - the reference's GPU default dtype (demo_helpers/misc.py:73-77) */
#define MDPT_PREC_BF16X3 1 /* split-bf16 (hi+lo) operands, 3 MFMA passes: fp32-class accuracy (parity mode)        */
#define MDPT_PREC_FP16 2   /* fp16 MFMA operands (v_mfma_*_f16: the bf16 rate, 11 instead of 8 significand bits; converts saturate at
                              +-65504 - and turn a NaN into a finite value; a NaN / inf in the input IMAGE still gives a NaN depth map,
                              in every mode: mdpt_set_nonfinite_propagation) - what the reference's device policy hands the model when
                              bf16 is not preferred (demo_helpers/misc.py:61-77: float16). The layer-scale-folded matrices are packed
                              times a power of two that the GEMMs undo exactly, so checkpoints with gammas of 1e-2 ... 1e-5 keep their
                              hi / lo planes in fp16's normal range                                                   */
#define MDPT_PREC_FP16X3 3 /* split-fp16 (hi+lo) operands, 3 passes                                                 */
#define MDPT_PREC_MIXED 4  /* fp16 operands; the op classes listed in mdpt_default_mixed_passes() run 3 or 2 passes, the others 1:
                              the cheapest per-class assignment that keeps the depth map within 1e-3 of the fp32 reference
                              (profiles/r05_precision_budget.md)                                                     */

/* op classes of the path (what mdpt_set_class_passes / MDPT_PREC_MIXED address) */
#define MDPT_CLASS_PATCH 0  /* patch-embed projection                    patch_embed.py:92                         */
#define MDPT_CLASS_QKV 1    /* attention QKV projection                  transformer_block.py:160                  */
#define MDPT_CLASS_ATTN 2   /* q k^T and p v inside the attention kernel transformer_block.py:164                  */
#define MDPT_CLASS_PROJ 3   /* attention output projection (+ SwinV2 patch merge)   transformer_block.py:168       */
#define MDPT_CLASS_FC1 4    /* MLP first linear                          misc_helpers.py:111                       */
#define MDPT_CLASS_FC2 5    /* MLP second linear                         misc_helpers.py:115                       */
#define MDPT_CLASS_REASM 6  /* reassembly convs (incl. BEiT readout)     reassembly_model.py:139-149               */
#define MDPT_CLASS_FUSION 7 /* RefineNet fusion: the 3x3 convs of the projection path's residual conv unit  fusion_model.py:151-154,210-220 */
#define MDPT_CLASS_HEAD 8   /* depth head, first 3x3 conv (C -> C/2)     head_model.py:74-76                       */
#define MDPT_CLASS_FUSION_IN 9 /* the fusion blocks' conv_reassembly units (RCU on the reassembly map before the prior is added),
                                  fusion_model.py:148-150: the least error-sensitive convs of the decoder                  */
#define MDPT_CLASS_HEAD_TAIL 10   /* depth head behind its upsample: 3x3 conv C/2 -> 32 (the 32 -> 1 projection runs in fp32)  head_model.py:78-85 */
#define MDPT_CLASS_FUSION_PROJ 11 /* the 1x1 output projection of every fusion block                                            fusion_model.py:178-182 */
#define MDPT_NUM_CLASSES 12

#define MDPT_FAMILY_DAV2 0
#define MDPT_FAMILY_DAV1 1
#define MDPT_FAMILY_BEIT 2
#define MDPT_FAMILY_SWINV2 3

#define MDPT_E_INVALID (-1)    /* bad argument / shape                                  */
#define MDPT_E_STATE (-2)      /* call order (e.g. forward before finalize)             */
#define MDPT_E_MISSING (-3)    /* a weight required by the config was not bound         */
#define MDPT_E_SHAPE (-4)      /* bound weight has the wrong shape                      */
#define MDPT_E_WORKSPACE (-5)  /* workspace / packed buffer too small                   */
#define MDPT_E_UNSUPPORTED (-6)/* config outside this build (e.g. head dim != 64)        */
#define MDPT_E_GRID (-7)       /* odd patch grid: the reference raises RuntimeError at fusion_model.py:151 */

typedef struct mdpt_handle mdpt_handle;

/* == the 11-key config dict of the reference (state_dict_conversion/config_from_original_state_dict.py:29-41),
 *    minus the two Python-only flags, plus the arithmetic mode. */
typedef struct mdpt_config {
    int32_t features_per_token;
    int32_t num_heads;
    int32_t num_blocks;
    int32_t reassembly_features[4];
    int32_t base_patch_grid_h, base_patch_grid_w;
    int32_t fusion_channels;
    int32_t patch_size_px;
    int32_t is_giant;  /* ViT-G: SwiGLU FFN instead of the GELU MLP (components/misc_helpers.py:125-185); Depth-Anything V2 only */
    int32_t is_metric; /* sigmoid instead of the final ReLU (head_model.py:84) */
    int32_t precision; /* MDPT_PREC_* */
    int32_t family;    /* MDPT_FAMILY_DAV2: Depth-Anything V2 (encoder tapped after each quarter of the blocks, image_encoder_model.py:88-93)
                          MDPT_FAMILY_DAV1: Depth-Anything V1 (tapped after each of the last four blocks,
                                            v1_depthanything/image_encoder_model.py:55-61; parameters named imgencoder.blocks.N...)
                          MDPT_FAMILY_BEIT: MiDaS v3.1 BEiT (v31_beit/: relative-position-bias attention with q/v bias, no position
                                            embedding or out-norm, readout projection in the reassembly, patch 16)
                          MDPT_FAMILY_SWINV2: MiDaS v3.1 SwinV2 (v31_swinv2/: 4 stages of shifted-window cosine attention, head dim 32,
                                            post-norm blocks, patch merging between stages, patch 4). Uses the swin_* fields below;
                                            features per stage = reassembly_features[], features_per_token = reassembly_features[0] */
    /* SwinV2 only (make_swinv2_dpt.py:67-79); ignored by the other families */
    int32_t swin_heads[4];             /* heads_per_stage */
    int32_t swin_layers[4];            /* layers_per_stage (even: blocks come in plain/shifted pairs, image_encoder_model.py:155-161) */
    int32_t swin_window_h, swin_window_w;
    int32_t swin_pretrained_window[4]; /* pretrained_window_sizes_per_stage, 0 = None */
} mdpt_config;

int mdpt_abi_version(void);
const char* mdpt_last_error(void);

/* replaces make_depthanythingv2_dpt(**config) (make_depthanythingv2_dpt.py:67-138): validates the config and
 * derives the list of parameters the model needs. */
int mdpt_create(const mdpt_config* cfg, mdpt_handle** out);
void mdpt_destroy(mdpt_handle* h);

/* Per-class MFMA pass count on top of whatever mdpt_config.precision chose: 1 = one rounded 16-bit plane per operand; 3 = hi + lo planes
 * for both operands (A_lo W_hi + A_hi W_lo + A_hi W_hi, fp32-class); 2 = ACTIVATIONS split, weights one plane (A_lo W_hi + A_hi W_hi) -
 * for the decoder classes whose error is the rounding of their activations, two thirds of the cost of 3 (not for MDPT_CLASS_ATTN, whose
 * operands are both activations).
 * MDPT_PASSES_2F8 / MDPT_PASSES_3F8 (fp16 operand modes; MDPT_CLASS_REASM, _FUSION, _FUSION_IN, _FUSION_PROJ, _HEAD): the same two / three
 * products with the CROSS TERMS (A_lo W_hi, A_hi W_lo: 2^-11 of the main term) on fp8 operands through gfx950's block-scaled MFMA at twice
 * the fp16 rate - activations E5M2 with one constant power-of-two scale, weights E4M3 with one power-of-two scale per output row
 * (csrc/f8_cross.h) - i.e. 1.5 / 2 pass-equivalents instead of 2 / 3 at the accuracy of the fp16 cross terms
 * (tests/precision_budget/emulate_operand_rounding.py, format "sf8"). A class whose contraction lengths are not all multiples of 128 (the
 * small encoders' reassembly widths, fusion widths below 128), the SwinV2 family and the bf16 operand modes run the fp16-plane form of the
 * same term count instead (mdpt_get_class_f8 tells). Changes the packed-weight layout and the workspace
 * plan: call it after mdpt_create and BEFORE mdpt_packed_bytes / mdpt_finalize / mdpt_workspace_bytes (bound pointers are kept).
 * mdpt_get_class_passes reads the current assignment; mdpt_default_mixed_passes fills the table MDPT_PREC_MIXED uses. */
#define MDPT_PASSES_2F8 4 /* A_hi W_hi on fp16 planes + A_lo W_hi on fp8 planes */
#define MDPT_PASSES_3F8 5 /* A_hi W_hi on fp16 planes + A_lo W_hi + A_hi W_lo on fp8 planes */
int mdpt_set_class_passes(mdpt_handle* h, int32_t op_class, int32_t passes);
int mdpt_get_class_passes(const mdpt_handle* h, int32_t op_class, int32_t* passes);
int mdpt_get_class_f8(const mdpt_handle* h, int32_t op_class, int32_t* on); /* 1: the class's cross terms really run on fp8 planes */
void mdpt_default_mixed_passes(int32_t passes[MDPT_NUM_CLASSES]);                      /* the Depth-Anything families */
void mdpt_default_mixed_passes_for(int32_t family, int32_t passes[MDPT_NUM_CLASSES]);  /* per MDPT_FAMILY_*: the MiDaS v3.1 families keep three
                                                                                          terms on the decoder's whole projection path. Round 6: the
                                                                                          decoder classes name the fp8 forms (MDPT_PASSES_*F8); BEiT adds
                                                                                          fc1 and fc2 at 3 and fusion_in at 2F8 (margin under either rounding
                                                                                          of the fp16 weight scale, profiles/r06_beitl_class_budget.txt) */
void mdpt_default_mixed_passes_r05(int32_t family, int32_t passes[MDPT_NUM_CLASSES]);  /* the 16-bit-plane table of round 5: what MDPT_PREC_MIXED gives a class
                                                                                          of a configuration that cannot run the fp8 forms (mdpt_get_class_f8) */
/* Token-mean compensation of the weight rounding (fp16 operand modes; on by default in MDPT_PREC_MIXED, available in MDPT_PREC_FP16): a
 * single-pass Linear of the encoder (QKV, proj, fc1, fc2) computes A fp16(W)^T; what the weight rounding loses is dominated by the part all
 * tokens of an image share, mean_t(A) (W - fp16(W))^T, which two small kernels turn into a per-image bias table the GEMM epilogue adds
 * (transformer_block.py:160,168, misc_helpers.py:111-115 restated with that term). Same call-order rule as mdpt_set_class_passes.
 * on: 0 off, 1 all four classes, or a mask (1 << MDPT_CLASS_QKV) | (1 << MDPT_CLASS_PROJ) | (1 << MDPT_CLASS_FC1) | (1 << MDPT_CLASS_FC2) of the classes to compensate. */
int mdpt_set_weight_rounding_compensation(mdpt_handle* h, int32_t on);

/* Parameter inventory, named with the reference's converted ("new format") keys, prefixed by component:
 * "patch_embed.proj.weight", "imgencoder.stages.0.blocks.0.attn.qkv.weight", "reassemble.spatial_upx4.resample.1.weight",
 * "fusion.blocks.0.conv_reassembly.resconv_seq.1.weight", "head.proj_1ch.2.bias", ...
 * (state_dict_conversion/convert_original_state_dict_keys.py:15-86). */
int mdpt_num_weights(const mdpt_handle* h);
const char* mdpt_weight_name(const mdpt_handle* h, int index);
int mdpt_weight_shape(const mdpt_handle* h, int index, int32_t* ndim, int64_t shape[4]);

/* element types of the tensors handed to mdpt_bind_weight / mdpt_forward / mdpt_allgather */
#define MDPT_DTYPE_F32 0
#define MDPT_DTYPE_BF16 1
#define MDPT_DTYPE_F16 2

/* replaces <sub-module>.load_state_dict(...) (make_depthanythingv2_dpt.py:55-59). `dev_ptr` is a contiguous device tensor of
 * element type `dtype` (MDPT_DTYPE_*: the parameter as the caller holds it, e.g. bf16 after model.to(torch.bfloat16)) in the
 * PyTorch layout of that parameter; it is only read during mdpt_finalize(). */
int mdpt_bind_weight(mdpt_handle* h, const char* name, const void* dev_ptr, int32_t dtype, int32_t ndim, const int64_t* shape);

/* replaces model.to(device, dtype) (run_image.py:158): one-time repack of all bound weights into MFMA-friendly
 * bf16 (hi[/lo]) [N][K] panels inside the caller-provided `packed_dev` buffer (size from mdpt_packed_bytes).
 * Strict: fails with MDPT_E_MISSING if any parameter is unbound. */
int mdpt_packed_bytes(const mdpt_handle* h, size_t* bytes);
int mdpt_finalize(mdpt_handle* h, void* packed_dev, size_t bytes, void* stream);

/* Workspace (activations) needed for a batch of B images of H x W pixels. */
int mdpt_workspace_bytes(const mdpt_handle* h, int32_t B, int32_t H, int32_t W, size_t* bytes);

/* replaces DPTModel.forward (dpt_model.py:61-83): image [B,3,H,W] (RGB, normalised; element type image_dtype) -> depth [B,H,W]
 * (element type depth_dtype: the reference returns the model dtype, dpt_model.py:105-107). The patchify kernel reads the image in its
 * own dtype and the fused head epilogue writes the depth in the requested one: a bf16 model pays no cast kernels.
 * H, W multiples of patch_size_px with an even patch grid. */
int mdpt_forward(mdpt_handle* h, const void* image_bchw, int32_t image_dtype, int32_t B, int32_t H, int32_t W, void* depth_bhw,
                 int32_t depth_dtype, void* workspace, size_t workspace_bytes, void* stream);

/* Stage-level entry points == the five sub-module calls (simple_examples/internal_features.py:39-45). */
/* PatchEmbed.forward (v2_depthanything/patch_embed.py:77-99): -> tokens [B, (H/P)*(W/P), F] */
int mdpt_patch_embed(mdpt_handle* h, const void* image_bchw, int32_t B, int32_t H, int32_t W, void* tokens_bnf, void* workspace,
                     size_t workspace_bytes, void* stream);
/* DinoV2Model4Stages.forward (image_encoder_model.py:80-94): tokens [B,gh*gw,F] -> 4 x [B, 1+gh*gw, F] */
int mdpt_encoder(mdpt_handle* h, const void* tokens_bnf, int32_t B, int32_t gh, int32_t gw, void* const stage_out[4],
                 void* workspace, size_t workspace_bytes, void* stream);
/* ReassembleModel.forward (reassembly_model.py:61-94): 4 x [B,1+gh*gw,F] -> [B,C,4gh,4gw],[B,C,2gh,2gw],[B,C,gh,gw],[B,C,gh/2,gw/2] */
int mdpt_reassemble(mdpt_handle* h, const void* const stage_in[4], int32_t B, int32_t gh, int32_t gw, void* const maps_out[4],
                    void* workspace, size_t workspace_bytes, void* stream);
/* FusionModel.forward (fusion_model.py:55-80): the 4 maps above -> [B,C,8gh,8gw] */
int mdpt_fusion(mdpt_handle* h, const void* const maps_in[4], int32_t B, int32_t gh, int32_t gw, void* fused_out, void* workspace,
                size_t workspace_bytes, void* stream);
/* mdpt_encoder plus the explicit attention weights of selected blocks: what the reference's non-optimised attention exposes through its
 * nn.Softmax module (enable_optimizations=False, v2_depthanything/components/transformer_block.py:101,126-131; hooked by
 * experiments/attention_visualization.py:325-332). attn_out has num_blocks entries; every non-NULL entry receives that block's
 * softmax(q k^T / sqrt(d) [+ relative-position bias]) as fp32 [B, heads, N, N] (N = 1 + gh*gw) for the ViT / BEiT families.
 * SwinV2 (whose window attention always goes through a hookable nn.Softmax, v31_swinv2/components/windowed_attention.py:60-61,119):
 * blocks are numbered stage-major, an entry receives softmax(cosine attention * logit scale + position bias + shift mask) of every
 * window as fp32 [B * windows, heads_of_the_stage, Nw, Nw] (windows in the reference's partition order, tokens in window order).
 * mdpt_attn_probe_shape() gives the shape of a block's entry for a batch / grid. */
int mdpt_encoder_probe(mdpt_handle* h, const void* tokens_bnf, int32_t B, int32_t gh, int32_t gw, void* const stage_out[4],
                       void* const* attn_out, void* workspace, size_t workspace_bytes, void* stream);
int mdpt_attn_probe_shape(const mdpt_handle* h, int32_t B, int32_t gh, int32_t gw, int32_t block, int64_t shape[4]);
/* The same pass with, in addition or instead, the OUTPUT TOKENS of selected transformer blocks: what a forward hook on a block module of
 * the reference receives (demo_helpers/model_capture.py:54-59, used by experiments/block_norm_visualization.py:282 on
 * v2_depthanything/components/transformer_block.py:41-62; SwinV2: v31_swinv2/image_encoder_model.py:213-225). block_out has one entry
 * per block (SwinV2: stage-major); every non-NULL entry receives fp32 [B, 1 + gh*gw, F] (ViT / BEiT families, cls row first) or
 * [B, tokens of the block's stage, features of the stage] (SwinV2). attn_out and block_out may each be NULL. */
int mdpt_encoder_probe_blocks(mdpt_handle* h, const void* tokens_bnf, int32_t B, int32_t gh, int32_t gw, void* const stage_out[4],
                              void* const* attn_out, void* const* block_out, void* workspace, size_t workspace_bytes, void* stream);
/* FusionModel.blocks[index].forward (fusion_model.py:89-114 top-most block, :148-154 regular blocks; called one by one by
 * experiments/fusion_scaling.py:330-334): reassembly map [B,C,sh,sw] (+ the previous block's output [B,C,sh,sw]; NULL for index 3,
 * the top-most block) -> [B,C,2sh,2sw]. */
int mdpt_fusion_block(mdpt_handle* h, int32_t index, const void* reasm_in, const void* prior_in, int32_t B, int32_t sh, int32_t sw,
                      void* out, void* workspace, size_t workspace_bytes, void* stream);
/* MonocularDepthHead.forward (head_model.py:89-106): [B,C,8gh,8gw] -> [B, gh*P, gw*P] */
int mdpt_head(mdpt_handle* h, const void* fused_in, int32_t B, int32_t gh, int32_t gw, void* depth_bhw, void* workspace,
              size_t workspace_bytes, void* stream);

/* Batches of at least `min_batch` images (default 8; 0 = never) are run by mdpt_forward as two half batches, one on the caller's
 * stream and one on an internal side stream (event fork / join, no host synchronisation): the halves' kernels overlap and fill
 * each other's partially occupied last round of tiles. Results are bit-identical to the unsplit run. mdpt_workspace_bytes
 * accounts for the split. */
int mdpt_set_batch_split(mdpt_handle* h, int32_t min_batch);

/* The reference's `enable_cache` (make_*_dpt(..., enable_cache=True); v2_depthanything/components/position_encoder.py:152-227 GridCache,
 * v31_beit/components/relative_positional_encoder.py, v31_swinv2 default True; run_video.py:144 switches it on). Off (default): every forward
 * recomputes the per-grid constants - the bicubic-resized position embedding, BEiT's resized relative-position tables, SwinV2's
 * continuous-position-bias tables, the zero pads of operand planes. On: the first mdpt_forward of a (workspace, B, H, W) leaves them in the
 * workspace and the following mdpt_forward calls on the SAME workspace and shape skip the kernels that write them (same bits). The cached state is
 * dropped by a different shape on that workspace, by any stage-level call on the handle, and by mdpt_finalize. With the cache on the caller must
 * not write into the workspace between calls, and a caller that frees a workspace and allocates a new one (which may land at the same address)
 * calls mdpt_set_grid_cache(h, 1) again: setting the switch drops every cached slot (the Python engine does so on every workspace allocation). */
int mdpt_set_grid_cache(mdpt_handle* h, int32_t on);

/* Latency mode (default off). Off: every image's result is bit-identical whatever batch it is part of (the kernels that run do
 * not depend on the batch size in any way that changes the arithmetic). On: launches that are too small to fill the GPU (batch 1 of
 * the small models) may use forms that change the summation order - today the attention kernel splits the key loop over the four
 * waves of a workgroup and merges the partial softmax states (ViT-S, batch 1: +13 %) - so results can differ in the last bit from
 * the batch-invariant form. Accuracy against the fp32 oracle is unchanged. Round 4: fc2 of a small batch (long K on the small GEMM tile:
 * one serial chain of 64 K tiles per workgroup at ViT-L) splits K into two fixed halves, twice the workgroups in flight; the second half's
 * partial sums are folded in by the LayerNorm that follows (no reduction launch); the long-K 3x3 convs of the coarse decoder levels run as K
 * ranges that store partial planes plus a small finishing kernel. The splits are fixed per shape: results are reproducible run to run. */
int mdpt_set_latency_mode(mdpt_handle* h, int32_t on);

/* Non-finite propagation (default on; additive to ABI v6). The reference's forward (muggled_dpt/dpt_model.py:61-83) turns an image that holds a
 * NaN / inf into an all-NaN depth map: the value reaches every token of that image through the first attention, and torch's ReLU keeps it. The
 * kernels here would hide it (the fp16 operand converts saturate through v_med3, the ReLUs are v_max): so mdpt_forward has its im2col kernel flag
 * such images (it reads every pixel anyway) and one small launch behind the head writes their depth maps as NaN - in every arithmetic mode, the
 * other images of the batch untouched. Cost: one B-word memset and one launch per forward, both stream-ordered and graph-capturable. Off = the
 * earlier behaviour (a finite, meaningless map for such an image). mdpt_forward_bgr's uint8 source cannot hold a non-finite value; the stage-level
 * entry points (mdpt_patch_embed ...) return what their own arithmetic gives; non-finite WEIGHTS are the caller's to check. */
int mdpt_set_nonfinite_propagation(mdpt_handle* h, int32_t on);

/* PatchEmbed.prepare_image (v2_depthanything/patch_embed.py:103-145): uint8 [in_h,in_w,3] BGR on the device ->
 * [3,out_h,out_w] RGB in the element type out_dtype (MDPT_DTYPE_*: the model's dtype, what mdpt_forward takes next - no cast kernel in
 * between), antialiased-bilinear resized exactly like F.interpolate(..., antialias=True) in fp32 and normalised with the
 * given per-channel mean/std (ImageNet values for Depth-Anything, 0.5/0.5 for BEiT). The caller picks out_h/out_w with the reference's size rule (multiples of 2*patch).
 * `interpolation` = the reference's interpolation_mode argument (patch_embed.py:108,141): bilinear (its default) or bicubic; torch itself
 * rejects antialias=True for every other mode, and so does this entry point (MDPT_E_UNSUPPORTED). */
#define MDPT_INTERP_BILINEAR 0
#define MDPT_INTERP_BICUBIC 1
int mdpt_prepare_image(const void* bgr_u8_hwc, int32_t in_h, int32_t in_w, void* out_chw, int32_t out_dtype, int32_t out_h, int32_t out_w,
                       const float rgb_mean[3], const float rgb_std[3], int32_t interpolation, void* stream);

/* DPTModel.inference's device half in ONE call (dpt_model.py:87-109: prepare_image_bgr -> forward; SURVEY §8(f) row 1 "fused with patchify"):
 * uint8 [in_h,in_w,3] BGR on the device -> depth [H,W]. The patch embedding's im2col kernel computes every pixel of the [3,H,W] model tensor from
 * the uint8 image itself (the arithmetic of mdpt_prepare_image, rounded to image_dtype = the model's dtype) and stores it straight into its rows:
 * the normalised image never exists in memory, one launch and one HBM round trip fewer than mdpt_prepare_image + mdpt_forward, bit-identical to
 * that pair. One image per call (the reference's inference is per image); H, W by the reference's size rule; workspace as for mdpt_forward(B = 1). */
int mdpt_forward_bgr(mdpt_handle* h, const void* bgr_u8_hwc, int32_t in_h, int32_t in_w, int32_t image_dtype, int32_t H, int32_t W, const float rgb_mean[3],
                     const float rgb_std[3], int32_t interpolation, void* depth_hw, int32_t depth_dtype, void* workspace, size_t workspace_bytes, void* stream);

/* Depth post-processing on the device (reference muggled_dpt/demo_helpers/postprocess.py and
 * run_3dviewer.py:576-590). All buffers are device pointers; `minmax` is a 2-float device buffer {min, max} and
 * `scratch8` 8 bytes of device scratch - nothing is read back to the host, nothing synchronises.
 *   mdpt_post_minmax ............ data.min(), data.max() of an fp32 array (normalize_01, postprocess.py:72-73)
 *   mdpt_post_scale_prediction .. F.interpolate(pred[:, None], size=(out_h, out_w), mode="bilinear") (postprocess.py:22-29);
 *                                 minmax_out != NULL also reduces min/max of the OUTPUT in the same pass
 *   mdpt_post_normalize ......... mode MDPT_POST_F32: (x - min) / (max - min) -> fp32            (postprocess.py:74)
 *                                 mode MDPT_POST_U8:  (255 * norm).byte() -> uint8 (truncation)     (postprocess.py:91)
 *                                 mode MDPT_POST_U24: round(16777215 * norm) -> BGRA uint8x4, B/G/R = low/mid/high byte,
 *                                                     alpha 0; lossy != 0 keeps only the high byte (run_3dviewer.py:579-590)
 *                                 minmax == NULL skips the normalisation (metric models, run_3dviewer.py:577-578) */
#define MDPT_POST_F32 0
#define MDPT_POST_U8 1
#define MDPT_POST_U24 2
int mdpt_post_minmax(const void* in_f32, size_t count, void* minmax_out, void* scratch8, void* stream);
int mdpt_post_scale_prediction(const void* in_bhw_f32, int32_t B, int32_t in_h, int32_t in_w, void* out_bhw_f32, int32_t out_h,
                               int32_t out_w, void* minmax_out, void* scratch8, void* stream);
int mdpt_post_normalize(const void* in_f32, size_t count, const void* minmax, void* out, int32_t mode, int32_t lossy, void* stream);

/* Stage boundaries of the LAST mdpt_forward on `workspace`, converted to reference layouts (debug / parity taps):
 * which = 0..3 encoder taps [B,N,F]; 4..7 reassembly maps (BCHW); 8 fused map [B,C,8gh,8gw]. */
int mdpt_export_tap(mdpt_handle* h, int32_t which, void* out_f32, void* workspace, size_t workspace_bytes, void* stream);

/* Test hooks (used by tests/ only): stop the encoder of the next mdpt_forward calls after sub-step `step` of transformer
 * block `block` (0 LN1, 1 QKV, 2 attention, 3 proj+residual, 4 LN2, 5 fc1+GELU, 6 fc2+residual; block = -1 disables), and
 * read an internal activation buffer ("resid","xn","q","k","vt","att","hbuf","im2col","pos","t0".."t3","u0","u1","d3",
 * "xf0".."xf3","a10".."a13","b20".."b23","flo0".."flo3","fused","h1","h1u") of the last forward as flat fp32 in its internal layout. */
int mdpt_debug_set_stop(mdpt_handle* h, int32_t block, int32_t step);
/* Test hook: latency mode's K split of the residual GEMMs applies from `min_k_tiles` 64-wide K tiles on (two ranges) and from
 * `four_k_tiles` on as four ranges (toy models have 4 K tiles). */
int mdpt_debug_set_ksplit_min(mdpt_handle* h, int32_t min_k_tiles, int32_t four_k_tiles);
/* Test / A-B hook: unsplit forwards (batches below mdpt_set_batch_split's threshold, i.e. the reference's frame-by-frame workload) queue each
 * reassembly branch (reassembly_model.py:61-94: four independent branches, one per encoder tap) on the handle's internal side stream as soon
 * as its tap exists, beside the remaining encoder blocks, and join before the fusion stage - same kernels, same bits, stream-ordering
 * semantics for the caller unchanged (as for the batch split). `on`: 0 = never (branches behind the encoder on the caller's stream), 1 = the
 * library's rule (default: encoders of width >= 1024 outside latency mode, where it measured faster), 2 = always. */
int mdpt_debug_set_reassemble_overlap(mdpt_handle* h, int32_t on);
/* Test / A-B hooks of the internal side stream, to be set before the first forward creates it. The runtime multiplexes the streams of a process
 * onto a few hardware queues; a side stream on the caller's queue would run the two halves of a split batch one after the other, so the first
 * forward on a caller stream PROBES up to four candidate streams on the GPU and keeps one that really runs beside the caller's (one host wait,
 * once per handle and caller stream, never inside a stream capture; csrc/stream_probe.hip). `probe` 0 takes the first candidate unseen.
 * Caveats of the probe: its host wait would invalidate a GLOBAL-mode stream capture another thread has open at that moment (run one forward before
 * capturing, or switch the probe off); a probe in which every candidate was rejected (GPU busy with other work) is repeated by the next forward, three
 * times at most; a stream handle recycled by the runtime after hipStreamDestroy keeps the pick of its predecessor.
 * `prio`: priority class of the candidates, 0 = default class (default), 1 = the device's lowest, -1 = highest (own queue pool, but measured
 * slower: the two classes do not overlap). mdpt_debug_side_stream_info: candidates created / candidates found on the caller's queue so far. */
int mdpt_debug_set_wscale_policy(mdpt_handle* h, int32_t all);  /* 1: scale every layer-scale-folded matrix of the fp16 build (the other valid rounding) */
int mdpt_debug_set_side_stream_priority(mdpt_handle* h, int32_t prio);
int mdpt_debug_set_side_stream_probe(mdpt_handle* h, int32_t on);
int mdpt_debug_side_stream_info(mdpt_handle* h, int32_t* candidates, int32_t* rejected);
int mdpt_debug_read(mdpt_handle* h, const char* name, void* out_f32, size_t out_floats, void* workspace, size_t workspace_bytes,
                    void* stream);

/* The handle-less kernel hooks below (mdpt_debug_gemm / _attention / _conv3) take their 16-bit operands as bf16 (0, default) or fp16 (1):
 * process-wide switch, tests only. */
int mdpt_debug_set_operand_format(int32_t fp16);

/* Kernel micro-benchmark hook: `iters` launches of the dense GEMM kernel, out[M,N] = A[M,K] * W[N,K]^T (bf16 operands,
 * fp32 and/or bf16 output), tile as in mdpt_set_gemm_tile. */
int mdpt_debug_gemm(const void* a_bf16, const void* w_bf16, void* out_f32, void* out_bf16, int32_t M, int32_t N, int32_t K,
                    int32_t tile, int32_t iters, void* stream, void* dbg_times_or_null);
/* test/bench hook: the fused attention kernel (replaces F.scaled_dot_product_attention, v2_depthanything/components/transformer_block.py:164)
 * on caller-provided head-major bf16 operands: Q (pre-scaled by 1/sqrt(64)), K [B, heads, npad, 64]; Vt [B, heads, 64, npadv] with zero pad
 * columns; out [B * npad, heads * 64]. */
int mdpt_debug_attention(const void* q_bf16, const void* k_bf16, const void* vt_bf16, void* out_bf16, int32_t B, int32_t heads, int32_t N,
                         int32_t npad, int32_t npadv, int32_t iters, void* stream);
/* test/bench hook: one 3x3 stride-1 conv Cin -> Cout (256 = the decoder's fusion width, every epilogue form; 128 = the head's first conv,
 * bias only) on caller-provided operands, through the implicit-GEMM
 * kernels (path 0, tile = MDPT_TILE_*) or the halo-staged kernel (path 1). Replaces one nn.Conv2d(…, 3, padding=1) of the reference's
 * ResidualConv2D / fusion blocks (v2_depthanything/fusion_model.py:178-182,210-220) incl. its skip add and the x2-upsampled prior. */
int mdpt_debug_conv3(const void* in_bf16, const void* w_packed_bf16, const void* bias_f32, const void* skip_f32, const void* up_f32, int32_t Hu,
                     int32_t Wu, void* out_f32, void* out_bf16, int32_t relu_bf16, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                     int32_t path, int32_t tile, int32_t iters, void* stream, void* dbg_times, const void* in_lo_bf16_or_null,
                     const void* w_lo_bf16_or_null, void* out_lo_bf16_or_null);  /* lo planes: the bf16x3 (fp32-class) mode */

/* Per-launch HIP-event profiler (bench.py's roofline leg): events are recorded on the launch stream around every
 * kernel launch while enabled. mdpt_profile_report() waits for the recorded events and writes a JSON summary
 * {"kernels":[{"name","launches","total_ms","avg_us","gflop","tflops"}...]} (algorithmic flops: 2*M*N*K per GEMM). */
int mdpt_profile_enable(int on);
int mdpt_profile_report(char* json_buf, size_t capacity);

/* Tuning knob for benchmarks: force a GEMM tile (0 = auto, 1 = 128x128, 2 = 256x256). */
int mdpt_set_gemm_tile(mdpt_handle* h, int32_t tile);

/* Data-parallel output collective (north star: "RCCL all-gather of the output depth maps"): thin wrapper over
 * ncclAllGather on a communicator created by the caller (torch.distributed's RCCL comm is used in the binding, so
 * this entry point is for non-torch hosts). `comm` is an ncclComm_t, `dtype` the element type of the maps (MDPT_DTYPE_*: the depth
 * maps travel in the dtype mdpt_forward wrote them in). Loaded lazily from librccl.so. */
int mdpt_allgather(void* comm, const void* send_dev, void* recv_dev, size_t count_per_rank, int32_t dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDPT_H */

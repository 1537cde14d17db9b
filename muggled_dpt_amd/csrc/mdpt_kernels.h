// Internal kernel-launcher interface of libmdpt (gfx950 only). Not part of the public C ABI
// (that is include/mdpt.h); this header is shared by the .hip kernel files and the host-side files (mdpt_internal.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "op_types.h"

// element types of tensors that cross the C ABI (= MDPT_DTYPE_* of include/mdpt.h)
enum { MDPT_DT_F32 = 0, MDPT_DT_BF16 = 1, MDPT_DT_F16 = 2 };

// ------------------------------------------------------------------------------------------------
// GEMM / implicit-GEMM convolution family:  C[M,N] = A[M,K] * W[N,K]^T   (both operands K-contiguous)
// A and W are bf16 "hi" planes plus optional "lo" planes (x = hi + lo, bf16x3 split precision).
// ------------------------------------------------------------------------------------------------
enum { MDPT_A_DENSE = 0, MDPT_A_TOKENS = 1, MDPT_A_CONV3 = 2 };
enum { MDPT_E_GENERIC = 0, MDPT_E_QKV = 1, MDPT_E_PATCH = 2, MDPT_E_D2S = 3, MDPT_E_HEAD = 4, MDPT_E_SWQKV = 5 };
enum { MDPT_ACT_NONE = 0, MDPT_ACT_RELU = 1, MDPT_ACT_GELU = 2 };
enum { MDPT_TILE_AUTO = 0, MDPT_TILE_128x128 = 1, MDPT_TILE_256x256 = 2, MDPT_TILE_128x32 = 3, MDPT_TILE_256x128 = 4, MDPT_TILE_PP256 = 5, MDPT_TILE_64x64 = 6 };

struct GemmParams {
    // operands
    const op_t* A_hi; const op_t* A_lo;   // activations, row stride lda (elements)
    const op_t* W_hi; const op_t* W_lo;   // weights [N][K]
    int M, N, K;                              // K % 64 == 0 (conv: K = 9 * Cin)
    int ldw;                                  // row stride of W in elements (0 = K: the packed panels; > K: a K range of wider rows, see ksplit)
    int M_alg;                                // > 0: rows that are algorithmic work (token rows without the per-image pad rows): profiler FLOP accounting only
    int lda;
    int npass;                                // 1 = one rounded plane per operand, 3 = split planes (A_lo*W_hi + A_hi*W_lo + A_hi*W_hi),
                                              // 2 = activations split, weights one plane (A_lo*W_hi + A_hi*W_hi)
    const op_t* zero_page;                  // >= 256 B of zeros (source for padded conv taps)
    int amode, ekind, tile;
    // A_TOKENS: logical row m = (b, p) reads source row b*tok_stride + 1 + p (skips the cls row)
    int tok_np, tok_stride;
    // A_CONV3: input NHWC [B,Hi,Wi,Cin] (Cin % 64 == 0), 3x3, pad 1, stride cstride -> [B,Ho,Wo,N]
    int Hi, Wi, Cin, Ho, Wo, cstride;
    // generic epilogue: v = acc (+bias[n]) -> act -> (*gamma[n]) (+resid[m,n]) (+up2x(up_src)[m,n])
    const float* bias; const float* gamma; const float* resid; int ldr;
    int bias_img_stride, bias_img_rows;       // stride != 0: `bias` is a table, row m uses bias + (m / bias_img_rows) * bias_img_stride (per-image
                                              // bias: BEiT readout cls term; token-mean compensation of the weight rounding in the fp16 modes)
    const float* up_src; int Hu, Wu;          // fp32 NHWC [B,Hu,Wu,N] added through x2 bilinear (align_corners)
    int act;
    float* out_f32; op_t* out_hi; op_t* out_lo; int ldc;
    int relu_bf16;                            // apply ReLU to the bf16 planes only (fp32 copy stays raw)
    int acc_init;                             // 1: accumulators start at resid[m,n] (resid == out_f32, gamma folded into W / bias): out = (resid + A W^T) + bias
    const float* wscale;                      // fp16 build, generic epilogue: device {s, 1 / s}, s a power of two the packed weight planes were multiplied by (the
                                              // layer-scale-folded matrices, mdpt_launch_weight_scale): accumulators start at resid * s, v = acc * (1 / s) (+ bias ...)
                                              // - exact rescalings, so the planes' lo halves stay out of fp16's subnormal range whatever gamma is. null = 1
    // E_QKV: scatter to head-major Q (pre-scaled), K and transposed V
    op_t* q_hi; op_t* q_lo; op_t* k_hi; op_t* k_lo; op_t* vt_hi; op_t* vt_lo;
    int F, heads, npad, npadv; float qscale;
    // E_SWQKV, the SwinV2 QKV projection (its own kernel; 8-phase tile only, 2F % 256 == 0): Q / K column tiles are written as the window
    // attention's operands - q / max(|q|, 1e-12) * logit_scale[h], k / max(|k|, 1e-12), heads of 32, row (img*swin_img_rows +
    // swin_tokmap[t] + h*npad) for image token t - and V column tiles as fp32 into out_f32 (ldc = 3F) for swin_v_prep
    const int* swin_tokmap; const float* swin_logit_scale; int swin_N, swin_img_rows;
    // ... and, when swin_vtokmap != nullptr (token runs of 4 stay together: gw, ww, shift % 4 == 0), the V column tiles are written as the
    // transposed window operand Vt[(img*nw + w)*heads + h][d][npadv] directly: element img*swin_img_velems + swin_vtokmap[t] + (h*32 + d)*npadv
    const int* swin_vtokmap; int swin_img_velems;
    // E_PATCH: out_f32[(b*npad + 1 + p), n] = acc + bias[n] + pos[p, n]   (m = b*tok_np + p)
    const float* pos;
    // E_D2S: transposed conv k==s as GEMM: n = (ky*k + kx)*Cout + co ; out NHWC [B, Ho*k, Wo*k, Cout]
    int d2s_k, d2s_cout;
    // E_HEAD: N == 32: depth[m] = final( sum_n relu(acc+bias[n]) * head_w[n] + head_b )
    const float* head_w; const float* head_b; int head_sigmoid; void* head_out; int head_out_dtype;  // MDPT_DT_*
    // test hook: per-workgroup phase timestamps (s_memtime): [start, first barrier passed, main loop done, epilogue done]
    int throughput_mode;  // 1: another stream runs the other half batch concurrently -> pick tiles by CU-time efficiency, not latency
    // K split: ksplit > 1 splits K into ksplit equal ranges, one workgroup each (grid.y). Range 0 runs the normal epilogue; range z >= 1 stores its
    // bare fp32 partial sums to ks_part + (z - 1) * M * ldc, and the CONSUMER adds them in the order z = 1, 2, ... - the LayerNorm that follows
    // (mdpt_launch_layernorm_addp, mdpt_launch_ln_res). A fixed split: bits do not depend on the batch. Two users:
    //  * latency mode, proj / fc2 of a small batch (one 16 ... 64-K-tile serial chain per 64x64 workgroup): dense A, generic epilogue, 64x64 tile
    //    (four ranges on the 128x128 tile - half the operand traffic per flop - measured 7 % slower: profiles/r04_b1_ksplit_sweep.txt);
    //  * SwinV2 fc2 with K >= 3072 at EVERY batch size (stages of few, long-K tiles: 108 8-phase tiles at batch 16): fp32 output without
    //    residual (DM_F32 form of the 8-phase kernel from 140 workgroups on, the 64x64 tile below - same sums, same bits).
    // fp8 cross terms (f8_cross.h, fp16 build; npass 2 or 3): f8 != 0 -> A_lo points at the e5m2 RESIDUE plane of the activations (bytes, rows of lda
    // bytes) and the cross terms run on the block-scaled MFMA: A_lo8 W8^T, then (npass == 3) A8 W8_lo^T with the e5m2 plane of the values at
    // (bytes) A_lo + a8_off, then A_hi W_hi^T on the fp16 planes. K % 128 == 0; conv K order of the fp8 planes: k = (cb128 * 9 + tap) * 128 + c.
    int f8; size_t a8_off;
    const unsigned char* W8; const unsigned char* W8_lo;   // [N][K] e4m3 bytes (row stride K)
    const unsigned char* S8; const unsigned char* S8_lo;   // [N] E8M0 bytes: power-of-two scale of every weight row
    // output planes for an F8 consumer: out_f8 != 0 -> out_lo receives BYTES: the e5m2 residue plane (element index = out_hi's) and, out_a8 != 0,
    // the e5m2 plane of the values themselves at byte offset out_f8 (= the plane's element count)
    size_t out_f8; int out_a8;
    int ksplit; float* ks_part;
    int ks_all;  // 1: EVERY range (z = 0 too) stores its bare partial sums, plane z at ks_part + z * M * ldc, and nothing else is written: a finishing
                 // kernel (mdpt_launch_ksplit_finish) adds the planes in the order z = 0, 1, ... and applies bias / ReLU / the output planes.
                 // Dense rows or 3x3-conv rows, 64x64 tile. Latency mode: the long-K convs of the 18^2 / 36^2 levels at batch 1.
    unsigned long long* dbg_times;
};


// ------------------------------------------------------------------------------------------------
// halo-staged 3x3 convolution, stride 1, pad 1, Cout = 256 (conv3h.hip): the 256-channel convs of the DPT decoder at large batch.
//   out = ((conv3x3(in) [+ bias]) [+ x2 bilinear (align_corners) of up_src]) [+ skip]  ->  out_f32 (optional) and out_bf (ReLU'd if relu_bf)
// Same K order (MDPT_PACK_CONV3 weights) and epilogue arithmetic as the MDPT_A_CONV3 path of mdpt_launch_gemm (generic epilogue, resid = skip).
// ------------------------------------------------------------------------------------------------
struct Conv3hParams {
    const op_t* in;          // NHWC [B, H, W, Cin] bf16 (hi plane), Cin % 128 == 0
    const op_t* up_in; int Hs, Ws;  // instead of `in` (Cout = 128, bf16 only): bf16 NHWC [B, Hs, Ws, Cin]; the conv input is its bilinear
                               // (align_corners) upsample to H x W, interpolated inside the kernel (up_bf16.h arithmetic)
    const op_t* in_lo;       // lo plane of the input: non-null selects the bf16x3 mode (then w_lo and, with out_bf, out_bf_lo are required)
    const op_t* w;           // [Cout][9 * Cin] bf16, MDPT_PACK_CONV3 order (hi plane)
    const op_t* w_lo;
    const float* bias;         // [Cout] or null
    const float* skip;         // fp32 NHWC [B, H, W, 256] or null
    const float* up_src; int Hu, Wu;  // fp32 NHWC [B, Hu, Wu, 256] or null
    float* out_f32;            // fp32 NHWC [B, H, W, 256] or null
    op_t* out_bf;            // bf16 NHWC [B, H, W, Cout] (hi plane); Cout = 128 may write the fp32 map alone instead
    op_t* out_bf_lo;
    int relu_bf;
    int B, H, W, Cin;
    int Cout;                  // 256 (every epilogue form) or 128 (bias-only bf16 output: the head's first conv)
    // fp8 cross terms (GemmParams::f8 has the description): in_lo = e5m2 residue plane (bytes, NHWC), values' plane at in_lo + a8_off (three terms:
    // w8_lo != null), weights [Cout][9 * Cin] e4m3 bytes in 128-channel-block K order + per-row E8M0 scales; output planes as GemmParams::out_f8
    int f8; size_t a8_off;
    const unsigned char* w8; const unsigned char* w8_lo; const unsigned char* s8; const unsigned char* s8_lo;
    size_t out_f8; int out_a8;
    unsigned long long* dbg_times;  // test hook: per-workgroup s_memtime stamps [start, first barrier, loop done, stores acknowledged, stores issued, XCC id]
};

// ------------------------------------------------------------------------------------------------
// fused multi-head attention (head dim 64), Q/K head-major [B,H,npad,64], Vt [B,H,64,npadv]
// ------------------------------------------------------------------------------------------------
struct AttnParams {
    const op_t* q_hi; const op_t* q_lo; const op_t* k_hi; const op_t* k_lo;
    const op_t* vt_hi; const op_t* vt_lo;
    op_t* out_hi; op_t* out_lo;           // [B*npad, F] token-major, column h*64 + d
    int B, heads, N, npad, npadv, F;
    int x3;
    // additive relative-position bias (BEiT): per-head extended LUT [heads][bias_elen] fp32, s[q][k] += lut[tq[q] - tk[k]]
    const float* bias_lut; int bias_elen; const int* tq; const int* tk;
    // SwinV2 window attention (rowmap != null): head_dim 32, B = images * win_nw windows of N tokens each,
    // rowmap[win_nw*N] = image token of every window token, region[win_nw][region_ld] = shifted-window region ids (null: no mask)
    int head_dim;  // 0 = 64
    int win_nw; const int* rowmap; const int* region; int region_ld;
    int bias_run4;       // window width and tokens per window are multiples of 4: four consecutive keys have consecutive bias-table indices
    int bias_row;        // (bias_run4) row length 2 ww - 1 of the table and window width ww: the LDS image of the table is re-strided so that 32
    int bias_ww;         // consecutive query tokens read 32 different banks (attention.hip, round 6); 0 = the table's own row length
    const float* swin_ls; // (window attention) the packed logit scale per head = exp(clamped logit_scale) * log2(e): Q carries it, the kernel's fixed softmax
                          // reference point is built from it (attention.hip FIXREF)
    int out_ld;          // row stride of out_hi / out_lo in elements (0 = F); pad columns are the caller's
    int allow_split_kv;  // latency mode: small launches may split the key loop over the waves (not batch-invariant in the last bit)
    int tail_last;  // set by the launcher: dispatch nearly empty last q-tiles after all full ones
};

// ------------------------------------------------------------------------------------------------
// bandwidth-bound helpers
// ------------------------------------------------------------------------------------------------
// weight repack: source (fp32 / bf16 / fp16: src_dtype) in PyTorch layout -> bf16 hi (+lo) [Np][Kp] rows, zero padded. Layout kinds:
enum { MDPT_PACK_LINEAR = 0,   // src [N][K]
       MDPT_PACK_CONV3 = 1,    // src [Cout][Cin][3][3] -> k = (cb*9 + ky*3+kx)*64 + c, ci = cb*64 + c (64-channel block outer, tap inner)
       MDPT_PACK_CONVT = 2,    // src [Cin][Cout][k][k] -> row n = (ky*k+kx)*Coutp + co, col ci
       MDPT_PACK_CONV3_KC32 = 3 };  // src [32][Cin][3][3] -> [Kp/8][32][8] (k = (ky*3+kx)*Cinp + ci in 8-element chunks, Np = 32): the
                                    // LDS image of head_tail_kernel, where a 32-lane fragment read is 512 consecutive bytes
struct BeitRelposBatch {
    const float* ref[32]; float* ext0; size_t ext_stride;  // block l writes ext0 + l * ext_stride (elements)
    int* tq; int* tk;
    int n, heads, Gh, Gw, gh, gw, N, ntok_pad;
};

// ------------------------------------------------------------------------------------------------
// fused tail of the depth head (head.hip): x(P/8) bilinear upsample -> 3x3 conv (cin -> 32) + ReLU -> 1x1 (32 -> 1) + ReLU | sigmoid
// ------------------------------------------------------------------------------------------------
struct HeadTailParams {
    const op_t* src;    // [B, Hi, Wi, cin] bf16 NHWC: output of the head's first conv (pad channels zero)
    const op_t* src_lo; // lo plane of the same map (src = hi): non-null selects the two-pass form (activations split, weights one plane)
    const op_t* w_kc;   // MDPT_PACK_CONV3_KC32 image of the 3x3 conv weights
    const float* bias;    // [32]
    const float* head_w;  // [32] 1x1 conv weights
    const float* head_b;  // [1]
    void* out; int out_dtype;  // depth [B, Ho, Wo], MDPT_DT_*
    int sigmoid;
    int B, Hi, Wi, Ho, Wo;
    unsigned long long* dbg_times;  // test hook (MDPT_HEAD_DBG=1): per-workgroup s_memtime stamps of the phases of its 2nd tile
};

// ------------------------------------------------------------------------------------------------
// SwinV2 helpers (swin.hip)
// ------------------------------------------------------------------------------------------------
// the same for every block of the encoder in ONE launch (the LUTs depend on weights and window sizes only, not on activations)
struct SwinCpbBatch {
    const float* w1[32]; const float* b1[32]; const float* w2[32]; float* lut[32];
    int heads[32], wh[32], ww[32], pre[32];
    int n, hidden;
};

// ------------------------------------------------------------------------------------------------
// depth post-processing (postprocess.hip); scratch2 = 2 uints of device scratch for the min/max reduction
// ------------------------------------------------------------------------------------------------
int mdpt_launch_post_minmax(const float* in, size_t n, float* minmax_out, unsigned* scratch2, hipStream_t stream);
int mdpt_launch_post_scale(const float* in, float* out, int B, int ih, int iw, int oh, int ow, float* minmax_out,
                           unsigned* scratch2, hipStream_t stream);
int mdpt_launch_post_normalize(const float* in, const float* minmax, void* out, size_t n, int mode, int lossy,
                               hipStream_t stream);

// stream_probe.hip: does `candidate` run kernels beside `waiter_stream`? (*seen != 0 after synchronising with waiter_stream)
int mdpt_launch_queue_probe(unsigned* flag, unsigned* seen, hipStream_t waiter_stream, hipStream_t candidate, hipEvent_t ready);

// ------------------------------------------------------------------------------------------------
// launchers: one set per operand format (op_types.h). A kernel file sees its own set through MDPT_FN; the host side (mdpt_internal.h)
// includes mdpt_launchers.inc a second time for the other format and picks per handle (OPL). C linkage: the two
// builds of a file differ in what `op_t*` points at, which must not reach the symbol names.
// ------------------------------------------------------------------------------------------------
extern "C" {
#include "mdpt_launchers.inc"
}

// HBM-bound helper kernels of the SwinV2 encoder (MiDaS v3.1, reference muggled_dpt/v31_swinv2/*) for gfx950:
//   * post-norm LayerNorm with residual add      (image_encoder_model.py:213-225: t + LN(attn(t)), u + LN(mlp(u)))
//   * window bookkeeping: window -> image token map incl. the cyclic shift, shifted-window region ids, relative
//     position index terms                        (components/windowed_attention.py:171-260, :394-439)
//   * continuous position bias: 16*sigmoid(MLP(log-spaced offsets)) per head as a LUT
//                                                 (components/relative_positional_encoder.py:60-93, :122-150)
//   * cosine-attention operand preparation: q/k L2-normalised, q scaled by the (pre-exponentiated) logit scale, window
//     partition + roll, head-major Q/K and transposed V planes for attn_kernel<.., 2, 32>
//                                                 (windowed_attention.py:100-123)
//   * patch merge gather (TL, BL, TR, BR concat)  (components/patch_merge.py:49-103)
// All are coalesced streaming kernels; the matmuls around them run in gemm.hip / attention.hip.

#include "mdpt_kernels.h"
#include <algorithm>
#include "mdpt_prof.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

__device__ __forceinline__ void split_store4(op_t* hi, op_t* lo, size_t off, f32x4 v) {
    opx4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = to_op(v[e]);
    *(opx4*)(hi + off) = h;
    if (lo) {
        opx4 l;
#pragma unroll
        for (int e = 0; e < 4; ++e) l[e] = to_op(v[e] - (float)h[e]);
        *(opx4*)(lo + off) = l;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

inline int grid_for(size_t total, int block = 256) {
    size_t g = (total + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

// ---------------------------------------------------------------------------------------------------
// y = LN(x) (+ add); one wave per row held in registers. out_f32 may alias `add` (in-place residual update).
// ---------------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void ln_res_kernel(const float* __restrict__ x, const float* add, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, float* out_f32, op_t* out_hi,
                                                     op_t* out_lo, int rows, int F, int ldp, const float* __restrict__ part) {
    // part: partial sums of the second K range of a K-split GEMM (GemmParams::ksplit): the row normalised is x + part
    // ldp = row stride of the bf16 planes (>= F; pad columns are zeroed once per forward by the caller, never written here)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (size_t)row * F;
    f32x4 v[NV], av[NV];
    // every load of the row - x and the residual it is added to - is issued before the first use (the residual used to be fetched behind
    // both reductions: two dependent memory round trips per row at 3.8 TB/s; round 4)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < F) {
            v[i] = *(const f32x4*)(xr + c);
            if (add) av[i] = *(const f32x4*)(add + (size_t)row * F + c);
            if (part) v[i] += *(const f32x4*)(part + (size_t)row * F + c);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < F) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)F;
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < F) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                ss += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)F + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < F) {
            const f32x4 g = *(const f32x4*)(gamma + c), bt = *(const f32x4*)(beta + c);
            const size_t o = (size_t)row * F + c;
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + bt[e];
            if (add) {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] += av[i][e];
            }
            if (out_f32) *(f32x4*)(out_f32 + o) = y;
            if (out_hi) split_store4(out_hi, out_lo, (size_t)row * ldp + c, y);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Window bookkeeping for one (grid, window, shift): windows are numbered wy * nWx + wx, tokens inside a window iy * ww + ix
// (image_to_windows, windowed_attention.py:262-289). The reference rolls the image by (-sh, -sw) before partitioning, so the
// window token at rolled position (y', x') is image token ((y' + sh) mod gh, (x' + sw) mod gw).
// region id = 3 * hslice + wslice of the rolled position, slices as make_shift_mask builds them (:419-426); a zero shift in
// one dimension makes that dimension's LAST slice cover everything (Python slice(-0, None)).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swin_window_map_kernel(int* rowmap, int* region, int* tq, int* tk, int gh, int gw, int wh, int ww,
                                                              int sh, int sw, int region_ld, int ntok_pad, int* tokmap, int tok_stride,
                                                              int* vtokmap, int vtok_stride) {
    const int wa = wh * ww, nwx = gw / ww, nw = (gh / wh) * nwx;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < ntok_pad) {
        const int t = gid < wa ? gid : wa - 1;
        const int iy = t / ww, ix = t - iy * ww;
        tq[gid] = (iy + wh - 1) * (2 * ww - 1) + ix + ww - 1;
        tk[gid] = iy * (2 * ww - 1) + ix;
    }
    if (gid >= nw * wa) return;
    const int w = gid / wa, i = gid - w * wa;
    const int wy = w / nwx, wx = w - wy * nwx, iy = i / ww, ix = i - iy * ww;
    const int yr = wy * wh + iy, xr = wx * ww + ix;
    const int tok = ((yr + sh) % gh) * gw + (xr + sw) % gw;
    rowmap[gid] = tok;
    if (tokmap) tokmap[tok] = w * tok_stride + i;
    if (vtokmap) vtokmap[tok] = w * vtok_stride + i;
    const int rh = sh == 0 ? 2 : (yr < gh - wh ? 0 : (yr < gh - sh ? 1 : 2));
    const int rw = sw == 0 ? 2 : (xr < gw - ww ? 0 : (xr < gw - sw ? 1 : 2));
    region[(size_t)w * region_ld + i] = 3 * rh + rw;
}

// ---------------------------------------------------------------------------------------------------
// Continuous position bias LUT: one workgroup per relative offset (dy, dx); lut[h][e] = 16 * sigmoid(W2 relu(W1 c + b1))[h]
// with c = sign(v) * log2(|8 v| + 1) / log2(8), v = offset / max(divider - 1, 1), divider = pretrained window size if given
// else the current window size (relative_positional_encoder.py:122-150, :79-84).
// ---------------------------------------------------------------------------------------------------
constexpr int kCpbPos = 8;  // table positions per workgroup: 53k one-position workgroups were dispatch-rate bound (0.55 ms per forward)

__device__ __forceinline__ void swin_cpb_body(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                              float* __restrict__ lut, int heads, int hidden, int wh, int ww, int pre) {
    extern __shared__ float hid[];  // [kCpbPos][hidden]
    __shared__ float coord[kCpbPos][2];
    const int rw = 2 * ww - 1, R = (2 * wh - 1) * rw;
    const int e0 = blockIdx.x * kCpbPos;
    if (e0 >= R) return;  // batched launch: the grid covers the largest window
    if (threadIdx.x < kCpbPos) {
        const int e = min(e0 + (int)threadIdx.x, R - 1);
        const int dy = e / rw - (wh - 1), dx = e % rw - (ww - 1);
        const float div_h = (float)max((pre > 0 ? pre : wh) - 1, 1), div_w = (float)max((pre > 0 ? pre : ww) - 1, 1);
        float cy = (float)dy / div_h, cx = (float)dx / div_w;
        const float sy = cy > 0.0f ? 1.0f : (cy < 0.0f ? -1.0f : 0.0f), sx = cx > 0.0f ? 1.0f : (cx < 0.0f ? -1.0f : 0.0f);
        coord[threadIdx.x][0] = sy * (log2f(fabsf(cy * 8.0f) + 1.0f) / 3.0f);
        coord[threadIdx.x][1] = sx * (log2f(fabsf(cx * 8.0f) + 1.0f) / 3.0f);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < kCpbPos * hidden; idx += blockDim.x) {
        const int q = idx / hidden, j = idx - q * hidden;
        hid[idx] = fmaxf(w1[2 * j] * coord[q][0] + w1[2 * j + 1] * coord[q][1] + b1[j], 0.0f);
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int h = wave; h < heads; h += 4) {
        float acc[kCpbPos];
#pragma unroll
        for (int q = 0; q < kCpbPos; ++q) acc[q] = 0.0f;
        for (int j = lane; j < hidden; j += 64) {
            const float w = w2[(size_t)h * hidden + j];
#pragma unroll
            for (int q = 0; q < kCpbPos; ++q) acc[q] += w * hid[q * hidden + j];
        }
#pragma unroll
        for (int q = 0; q < kCpbPos; ++q) {
            const float s = wave_sum(acc[q]);
            if (lane == 0 && e0 + q < R) lut[(size_t)h * R + e0 + q] = 16.0f / (1.0f + expf(-s));
        }
    }
}

__global__ __launch_bounds__(256) void swin_cpb_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                       float* __restrict__ lut, int heads, int hidden, int wh, int ww, int pre) {
    swin_cpb_body(w1, b1, w2, lut, heads, hidden, wh, ww, pre);
}

__global__ __launch_bounds__(256) void swin_cpb_batch_kernel(const SwinCpbBatch b) {
    const int l = blockIdx.y;
    swin_cpb_body(b.w1[l], b.b1[l], b.w2[l], b.lut[l], b.heads[l], b.hidden, b.wh[l], b.ww[l], b.pre[l]);
}

// ---------------------------------------------------------------------------------------------------
// Q / K preparation. qkv fp32 [B*N, 3F] (biases already added by the GEMM epilogue) -> head-major window operands
//   Q[(p*H + h)*npad + i][32] = logit_scale[h] * q / max(|q|, 1e-12),  K likewise without the scale   (F.normalize eps)
// p = image*nW + window, i = token inside the window. One thread = 4 consecutive d of one (token, q|k, head);
// the 8 lanes of a head reduce |.|^2 with three xor-shuffles: group c with group c+4 first, then neighbours, then pairs - the order
// (and the unfused multiplies: fp contraction off) of epilogue_swin_qk in gemm.hip, which produces the same bits straight out of the
// QKV GEMM's accumulators when the 8-phase tile runs.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swin_qk_prep_kernel(const float* __restrict__ qkv, const int* __restrict__ rowmap,
                                                           const float* __restrict__ logit_scale, op_t* q_hi, op_t* q_lo, op_t* k_hi,
                                                           op_t* k_lo, int B, int N, int nw, int wa, int npad, int heads) {
#pragma clang fp contract(off)
    const int F = heads * 32;
    const size_t per_tok = (size_t)2 * heads * 8;
    const size_t total = (size_t)B * nw * wa * per_tok;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx & 7);
        const int h = (int)((idx >> 3) % heads);
        const int which = (int)((idx / ((size_t)heads * 8)) & 1);
        const size_t tokw = idx / per_tok;           // (p, i)
        const int i = (int)(tokw % wa);
        const size_t p = tokw / wa;
        const int w = (int)(p % nw);
        const size_t img = p / nw;
        const size_t src = (img * N + rowmap[(size_t)w * wa + i]) * (size_t)(3 * F) + (size_t)which * F + h * 32 + g * 4;
        f32x4 v = *(const f32x4*)(qkv + src);
        float ss = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        ss += __shfl_xor(ss, 4);
        ss += __shfl_xor(ss, 1);
        ss += __shfl_xor(ss, 2);
        float scale = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
        scale *= which == 0 ? logit_scale[h] : 1.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= scale;
        const size_t dst = ((p * heads + h) * npad + i) * 32 + g * 4;
        if (which == 0) split_store4(q_hi, q_lo, dst, v);
        else split_store4(k_hi, k_lo, dst, v);
    }
}

// V preparation: Vt[(p*H + h)*32 + d][npadv], token-contiguous, pad columns [wa, npadv) written as zero.
// One workgroup per (window p, head h): the window's [wa tokens][32 d] fp32 block is read with 128-byte rows (8 lanes x 16 B per token),
// transposed through an LDS tile [32][npadv + 8] bf16 (hi, and lo in bf16x3 mode) and written out as whole 16-byte runs of the
// token-contiguous rows. (Round 2 wrote every element with its own 2-byte global store: 74 us per block for Q, K and V at SwinV2-L
// stage 2 where the bytes moved take ~30 us.)
__global__ __launch_bounds__(256) void swin_v_prep_kernel(const float* __restrict__ qkv, const int* __restrict__ rowmap, op_t* vt_hi,
                                                          op_t* vt_lo, int B, int N, int nw, int wa, int npadv, int heads) {
    extern __shared__ __attribute__((aligned(16))) op_t vtile[];  // [planes][32][pitch]
    const int F = heads * 32, pitch = npadv + 8;
    const size_t ph = blockIdx.x;  // p * heads + h
    const int h = (int)(ph % heads);
    const size_t p = ph / heads;
    const int w = (int)(p % nw);
    const size_t img = p / nw;
    const int g = threadIdx.x & 7;
    op_t* t_hi = vtile;
    op_t* t_lo = vtile + 32 * pitch;
    for (int i = threadIdx.x >> 3; i < npadv; i += 32) {
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (i < wa) v = *(const f32x4*)(qkv + (img * N + rowmap[(size_t)w * wa + i]) * (size_t)(3 * F) + 2 * F + h * 32 + g * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const op_t hi = to_op(v[e]);
            t_hi[(g * 4 + e) * pitch + i] = __builtin_bit_cast(op_t, hi);
            if (vt_lo) {
                const op_t lo = to_op(v[e] - (float)hi);
                t_lo[(g * 4 + e) * pitch + i] = __builtin_bit_cast(op_t, lo);
            }
        }
    }
    __syncthreads();
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    const int chunks = npadv >> 3;  // 16-byte runs per row (npadv % 64 == 0)
    for (int idx = threadIdx.x; idx < 32 * chunks; idx += 256) {
        const int d = idx / chunks, c = idx - d * chunks;
        const size_t o = (ph * 32 + d) * (size_t)npadv + c * 8;
        *(u32x4_t*)(vt_hi + o) = *(const u32x4_t*)(t_hi + d * pitch + c * 8);
        if (vt_lo) *(u32x4_t*)(vt_lo + o) = *(const u32x4_t*)(t_lo + d * pitch + c * 8);
    }
}

// ---------------------------------------------------------------------------------------------------
// Patch merge gather: tokens fp32 [B, gh, gw, C] -> bf16 rows [B*(gh/2)*(gw/2), 4C] = cat(TL, BL, TR, BR) (patch_merge.py:79-91)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swin_merge_gather_kernel(const float* __restrict__ tok, op_t* out_hi, op_t* out_lo, int B, int gh,
                                                                int gw, int C) {
    const int oh = gh / 2, ow = gw / 2, cq = C / 4;
    const size_t total = (size_t)B * oh * ow * 4 * cq;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % cq) * 4;
        const int qd = (int)((idx / cq) & 3);
        const size_t orow = idx / ((size_t)4 * cq);
        const int ox = (int)(orow % ow), oy = (int)((orow / ow) % oh);
        const size_t b = orow / ((size_t)ow * oh);
        const int y = 2 * oy + (qd & 1), x = 2 * ox + (qd >> 1);
        const f32x4 v = *(const f32x4*)(tok + ((b * gh + y) * gw + x) * (size_t)C + c4);
        split_store4(out_hi, out_lo, orow * (size_t)(4 * C) + (size_t)qd * C + c4, v);
    }
}

// fp32 -> bf16 hi (+lo) planes, flat
__global__ __launch_bounds__(256) void f32_to_planes_kernel(const float* __restrict__ in, op_t* out_hi, op_t* out_lo, size_t n4, int F4,
                                                            int ld) {
    // rows of F4 float4 each -> planes with row stride ld (elements); ld == 4 * F4: plain contiguous conversion
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n4; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t row = idx / F4, c4 = idx - row * F4;
        split_store4(out_hi, out_lo, row * ld + c4 * 4, *(const f32x4*)(in + idx * 4));
    }
}

}  // namespace

#define LAUNCH_RET() return (int)hipGetLastError()

int MDPT_FN(mdpt_launch_ln_res)(const float* x, const float* add, const float* gamma, const float* beta, float eps, float* out_f32, op_t* out_hi,
                       op_t* out_lo, int rows, int F, hipStream_t stream, int ld_planes, const float* part) {
    if (ld_planes <= 0) ld_planes = F;
    if ((F & 3) || F > 2048 || ld_planes < F || (ld_planes & 3)) return (int)hipErrorInvalidValue;
    if (rows <= 0) return 0;
    MdptProfScope prof("ln_res_kernel", 0.0, stream);
    const dim3 grid((rows + 3) / 4), block(256);
#define LN_CASE(NV) hipLaunchKernelGGL(ln_res_kernel<NV>, grid, block, 0, stream, x, add, gamma, beta, eps, out_f32, out_hi, out_lo, rows, F, ld_planes, part)
    if (F <= 256) LN_CASE(1);
    else if (F <= 512) LN_CASE(2);
    else if (F <= 1024) LN_CASE(4);
    else if (F <= 1536) LN_CASE(6);
    else LN_CASE(8);
#undef LN_CASE
    LAUNCH_RET();
}

int MDPT_FN(mdpt_launch_swin_window_map)(int* rowmap, int* region, int* tq, int* tk, int gh, int gw, int wh, int ww, int sh, int sw, int region_ld,
                                int ntok_pad, hipStream_t stream, int* tokmap, int tok_stride, int* vtokmap, int vtok_stride) {
    if (wh <= 0 || ww <= 0 || gh % wh || gw % ww || region_ld < wh * ww) return (int)hipErrorInvalidValue;
    const size_t work = (size_t)gh * gw > (size_t)ntok_pad ? (size_t)gh * gw : (size_t)ntok_pad;
    hipLaunchKernelGGL(swin_window_map_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, stream, rowmap, region, tq, tk, gh, gw, wh,
                       ww, sh, sw, region_ld, ntok_pad, tokmap, tok_stride, vtokmap, vtok_stride);
    LAUNCH_RET();
}

int MDPT_FN(mdpt_launch_swin_cpb)(const float* w1, const float* b1, const float* w2, float* lut, int heads, int hidden, int wh, int ww, int pretrained,
                         hipStream_t stream) {
    const int R = (2 * wh - 1) * (2 * ww - 1);
    MdptProfScope prof("swin_cpb_kernel", 0.0, stream);
    hipLaunchKernelGGL(swin_cpb_kernel, dim3((R + kCpbPos - 1) / kCpbPos), dim3(256), (size_t)kCpbPos * hidden * 4, stream, w1, b1, w2, lut, heads,
                       hidden, wh, ww, pretrained);
    LAUNCH_RET();
}

int MDPT_FN(mdpt_launch_swin_cpb_batch)(const SwinCpbBatch& b, hipStream_t stream) {
    if (b.n < 1 || b.n > 32) return (int)hipErrorInvalidValue;
    int rmax = 0;
    for (int l = 0; l < b.n; ++l) rmax = std::max(rmax, (2 * b.wh[l] - 1) * (2 * b.ww[l] - 1));
    MdptProfScope prof("swin_cpb_batch_kernel", 0.0, stream);
    hipLaunchKernelGGL(swin_cpb_batch_kernel, dim3((rmax + kCpbPos - 1) / kCpbPos, b.n), dim3(256), (size_t)kCpbPos * b.hidden * 4, stream, b);
    LAUNCH_RET();
}

int MDPT_FN(mdpt_launch_swin_qkv_prep)(const float* qkv, const int* rowmap, const float* logit_scale, op_t* q_hi, op_t* q_lo, op_t* k_hi,
                              op_t* k_lo, op_t* vt_hi, op_t* vt_lo, int B, int N, int nw, int wa, int npad, int npadv, int heads,
                              hipStream_t stream, bool qk) {
    if (qk) {
        MdptProfScope prof_qk("swin_qk_prep", 0.0, stream);
        hipLaunchKernelGGL(swin_qk_prep_kernel, dim3(grid_for((size_t)B * nw * wa * 2 * heads * 8)), dim3(256), 0, stream, qkv, rowmap, logit_scale,
                           q_hi, q_lo, k_hi, k_lo, B, N, nw, wa, npad, heads);
    }
    MdptProfScope prof("swin_v_prep", 0.0, stream);
    const size_t v_lds = (size_t)(vt_lo ? 2 : 1) * 32 * (npadv + 8) * 2;
    if (v_lds > 160 * 1024 || (npadv & 63)) return (int)hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)swin_v_prep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    hipLaunchKernelGGL(swin_v_prep_kernel, dim3((unsigned)((size_t)B * nw * heads)), dim3(256), v_lds, stream, qkv, rowmap, vt_hi, vt_lo, B, N, nw, wa,
                       npadv, heads);
    LAUNCH_RET();
}

int MDPT_FN(mdpt_launch_swin_merge_gather)(const float* tok, op_t* out_hi, op_t* out_lo, int B, int gh, int gw, int C, hipStream_t stream) {
    if ((gh & 1) || (gw & 1) || (C & 3)) return (int)hipErrorInvalidValue;
    MdptProfScope prof("swin_merge_gather_kernel", 0.0, stream);
    hipLaunchKernelGGL(swin_merge_gather_kernel, dim3(grid_for((size_t)B * gh * gw * (C / 4))), dim3(256), 0, stream, tok, out_hi, out_lo, B, gh,
                       gw, C);
    LAUNCH_RET();
}

int MDPT_FN(mdpt_launch_f32_to_planes)(const float* in, op_t* out_hi, op_t* out_lo, size_t rows, int F, int ld, hipStream_t stream) {
    if ((F & 3) || ld < F || (ld & 3)) return (int)hipErrorInvalidValue;
    const size_t n4 = rows * (size_t)(F / 4);
    hipLaunchKernelGGL(f32_to_planes_kernel, dim3(grid_for(n4)), dim3(256), 0, stream, in, out_hi, out_lo, n4, F / 4, ld);
    LAUNCH_RET();
}

// Fused tail of the monocular depth head for gfx950 (MI355X):
//
//     x (P/8) bilinear upsample (align_corners=True)  ->  3x3 conv (C/2 -> 32) + ReLU  ->  1x1 conv (32 -> 1) + ReLU | sigmoid
//
// Reference: MonocularDepthHead.forward, v2_depthanything/head_model.py:74-85 (SpatialUpsampleLayer after the first 3x3 conv, then
// proj_1ch) and :100-106; SpatialUpsampleLayer = F.interpolate(scale_factor, "bilinear", align_corners=True), components/misc_helpers.py:39-42.
//
// Unfused, the upsampled C/2-channel map at full image resolution is the largest tensor of the whole forward pass (ViT-L, batch 32:
// 2.08 GB written by the upsample kernel and read back 9x from L2 by the im2col-free conv: profiles/r01_hbm_traffic.md). Here it only
// ever exists as a 16x16-pixel tile (+1 halo) in LDS:
//
//   persistent workgroup (8 waves), weights of the 3x3 conv resident in LDS for its whole life ([K/8][32][8] bf16 image: a fragment
//   read is 32 lanes x 16 consecutive bytes = conflict-free), per output tile:
//     1. halo tile: the 18x18 upsampled pixels the tile's 3x3 taps touch are interpolated from the low-resolution map (four 16-byte
//        L1/L2-resident loads + 3 lerps per 8 channels, same arithmetic as upsample_kernel) and written to LDS as bf16; pixels
//        outside the image are the conv's zero padding. 16-byte channel chunks are XOR-swizzled with the pixel's column so that the
//        32 pixels of an MFMA fragment hit distinct banks for every tap.
//     2. implicit GEMM straight out of LDS: wave w owns tile rows 2w, 2w+1 (32 pixels); for each of the 9 taps the B fragment is the
//        halo read shifted by (ky, kx) - no im2col, no re-fetch from L2. 32x32x16 bf16 MFMA with the WEIGHTS as the A operand, so
//        the accumulator holds C[n][pixel]: a lane owns one pixel and 16 of the 32 conv outputs.
//     3. epilogue in registers: + bias, ReLU, dot with the 1x1 conv weights (16 per lane + one exchange with lane^32), + bias,
//        ReLU | sigmoid, store in the caller's dtype.
// LDS: 9*CIN*64 B of weights + 324*CIN*2 B of halo = 153 KiB at CIN = 128 (one workgroup per CU), 76.5 KiB at CIN = 64 (two).
// The MFMA phase is LDS-bandwidth bound by construction (N = 32: every fragment byte read feeds one MFMA; 2 x 1 KiB reads per
// 32-cycle MFMA x 8 waves = 256 B/clk, the LDS peak), the staging phase VALU bound; both are far cheaper than the 4.2 GB of
// HBM/L2 traffic they replace.

#include "mdpt_kernels.h"
#include "mdpt_prof.h"
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

namespace {

constexpr int TS = 16, HS = TS + 2;  // output tile side, halo side

template <int CIN>
__global__ __launch_bounds__(512, 1) void head_tail_kernel(const HeadTailParams p) {
#pragma clang fp contract(off)  // same rounding as upsample_kernel (and independent of how the compiler would fuse per instantiation)
    constexpr int NCH = CIN / 8;               // 16-byte channel chunks per pixel
    constexpr int PIXB = CIN * 2;              // bytes per halo pixel
    constexpr int W_BYTES = 9 * CIN * 32 * 2;  // [9*NCH][32][8] bf16
    constexpr int KSTEPS = CIN / 16;           // MFMA k-steps per tap
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sW = smem;
    char* const sH = smem + W_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    // ---- weights -> LDS once per workgroup (linear copy of the pre-arranged image)
    for (int i = tid; i < W_BYTES / 16; i += 512) *(u32x4*)(sW + (size_t)i * 16) = *(const u32x4*)((const char*)p.w_kc + (size_t)i * 16);

    // ---- per-lane epilogue constants: conv output n(r) = (r&3) + 8*(r>>2) + 4*half
    float bias_r[16], hw_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = (r & 3) + 8 * (r >> 2) + 4 * half;
        bias_r[r] = p.bias[n];
        hw_r[r] = p.head_w[n];
    }
    const float head_b = p.head_b[0];

    const float sy = p.Ho > 1 ? (float)(p.Hi - 1) / (float)(p.Ho - 1) : 0.0f;
    const float sx = p.Wo > 1 ? (float)(p.Wi - 1) / (float)(p.Wo - 1) : 0.0f;
    const int tiles_x = (p.Wo + TS - 1) / TS, tiles_y = (p.Ho + TS - 1) / TS;
    const int ntiles = p.B * tiles_y * tiles_x;

    // fragment geometry of this lane: pixel (py, px) of the tile, channel chunk parity = half
    const int py = 2 * wave + (l31 >> 4), px = l31 & 15;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const int oy0 = ty * TS, ox0 = tx * TS;
        __syncthreads();  // previous tile's MFMA phase is done with the halo (first pass: the weight copy is published below)

        // ---- 1. halo tile
        const bf16_t* src = p.src + (size_t)b * p.Hi * p.Wi * CIN;
        for (int item = tid; item < HS * HS * NCH; item += 512) {
            const int c = item % NCH, hp = item / NCH;
            const int hy = hp / HS, hx = hp - hy * HS;
            const int oy = oy0 + hy - 1, ox = ox0 + hx - 1;
            bf16x8 out;
            if ((unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo) {
                const float fy = sy * (float)oy, fx = sx * (float)ox;
                const int y0 = (int)fy, x0 = (int)fx;
                const int y1 = y0 + (y0 < p.Hi - 1), x1 = x0 + (x0 < p.Wi - 1);
                const float ly = fy - (float)y0, lx = fx - (float)x0;
                const bf16x8 v00 = *(const bf16x8*)(src + ((size_t)y0 * p.Wi + x0) * CIN + c * 8);
                const bf16x8 v01 = *(const bf16x8*)(src + ((size_t)y0 * p.Wi + x1) * CIN + c * 8);
                const bf16x8 v10 = *(const bf16x8*)(src + ((size_t)y1 * p.Wi + x0) * CIN + c * 8);
                const bf16x8 v11 = *(const bf16x8*)(src + ((size_t)y1 * p.Wi + x1) * CIN + c * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    out[e] = (__bf16)((1.0f - ly) * ((1.0f - lx) * (float)v00[e] + lx * (float)v01[e]) +
                                      ly * ((1.0f - lx) * (float)v10[e] + lx * (float)v11[e]));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) out[e] = (__bf16)0.0f;
            }
            const int key = (NCH == 16 ? hx : (hx >> 1)) & (NCH - 1);
            *(bf16x8*)(sH + (size_t)hp * PIXB + ((c ^ key) << 4)) = out;
        }
        __syncthreads();

        // ---- 2. implicit GEMM out of LDS: C[n][pixel] += W[n][tap, ci] * halo[pixel + tap][ci]
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int hx = px + kx;
            const int key = (NCH == 16 ? hx : (hx >> 1)) & (NCH - 1);
            const char* hrow = sH + (size_t)((py + ky) * HS + hx) * PIXB;
            const char* wrow = sW + ((size_t)tap * NCH * 32 + l31) * 16;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const int c = ks * 2 + half;
                const bf16x8 wf = *(const bf16x8*)(wrow + (size_t)c * 32 * 16);
                const bf16x8 xf = *(const bf16x8*)(hrow + ((c ^ key) << 4));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc, 0, 0, 0);
            }
        }

        // ---- 3. relu(conv + bias) . w + b -> relu | sigmoid   (head_model.py:80-85)
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += fmaxf(acc[r] + bias_r[r], 0.0f) * hw_r[r];
        s += __shfl_xor(s, 32);
        s += head_b;
        const float dv = p.sigmoid ? 1.0f / (1.0f + __expf(-s)) : fmaxf(s, 0.0f);
        const int oy = oy0 + py, ox = ox0 + px;
        if (half == 0 && oy < p.Ho && ox < p.Wo) {
            const size_t o = ((size_t)b * p.Ho + oy) * p.Wo + ox;
            if (p.out_dtype == MDPT_DT_BF16) ((__bf16*)p.out)[o] = (__bf16)dv;
            else if (p.out_dtype == MDPT_DT_F16) ((_Float16*)p.out)[o] = (_Float16)dv;
            else ((float*)p.out)[o] = dv;
        }
    }
}

template <int CIN>
int launch_cin(const HeadTailParams& p, hipStream_t stream) {
    constexpr unsigned LDS = 9 * CIN * 32 * 2 + HS * HS * CIN * 2;
    auto kern = head_tail_kernel<CIN>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int ntiles = p.B * ((p.Ho + TS - 1) / TS) * ((p.Wo + TS - 1) / TS);
    const int per_cu = LDS <= 80 * 1024 ? 2 : 1;
    const int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;  // persistent: the weights are staged once per workgroup
    static char prof_name[48] = "";
    if (!prof_name[0]) snprintf(prof_name, sizeof(prof_name), "head_tail_kernel<%d>", CIN);
    MdptProfScope prof(prof_name, 2.0 * p.B * p.Ho * p.Wo * 32.0 * 9.0 * CIN, stream);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, stream, p);
    return (int)hipGetLastError();
}

}  // namespace

bool mdpt_head_tail_supported(int cin) { return cin == 64 || cin == 128; }

int mdpt_launch_head_tail(const HeadTailParams& p, int cin, hipStream_t stream) {
    if (p.B <= 0 || p.Hi <= 0 || p.Wi <= 0 || p.Ho <= 0 || p.Wo <= 0) return (int)hipErrorInvalidValue;
    if (cin == 128) return launch_cin<128>(p, stream);
    if (cin == 64) return launch_cin<64>(p, stream);
    return (int)hipErrorInvalidValue;
}

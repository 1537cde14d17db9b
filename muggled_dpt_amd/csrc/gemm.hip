// MFMA GEMM / implicit-GEMM 3x3 convolution family for gfx950 (MI355X).
//
//   C[M,N] = A[M,K] * W[N,K]^T      A, W: bf16, K-contiguous ("B^T input");  fp32 accumulate
//
// One kernel template serves every dense contraction on the DPT path (reference call sites:
// qkv/proj Linear  v2_depthanything/components/transformer_block.py:160,168; MLP misc_helpers.py:111-115;
// patch conv patch_embed.py:92; 1x1 convs reassembly_model.py:238,261,301 + fusion_model.py:178-182;
// ConvTranspose2d k==s reassembly_model.py:262-269; 3x3 convs reassembly_model.py:135,302-309,
// fusion_model.py:210-220, head_model.py:74-85). It is parameterised by
//   * the A-row address generator (dense rows | token rows without cls | 3x3 taps over NHWC, im2col-free)
//   * the epilogue (bias/act/layer-scale/residual/upsample-add | QKV head-major scatter | patch+pos |
//     depth-to-space | fused 32->1 depth head)
//   * the tile shape.
//
// Structure (per workgroup): BMxBNx64 tiles, 64-lane waves each owning a (BM/WM)x(BN/WN) sub-tile of
// 32x32x16 bf16 MFMAs. Operand tiles go HBM -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4, 1 KiB
// per wave-instruction, no VGPR round trip) into a 2-deep ring; one barrier per K-step (the DMA for
// step t+1 is in flight while step t computes). The LDS image is row-major [row][64 k] (128-B rows) with
// the 16-B chunk index XOR-swizzled by ((row>>1)&7): because LDS-DMA writes lane-linear, the swizzle is
// applied to the per-lane *source* address and again on the ds_read_b128 side (conflict-free for the
// 32x32x16 fragment pattern: 16 rows x 16 B land on 16 distinct 16-B slots of the 256-B bank row).
// bf16x3 mode (npass == 3) runs the K loop three times (A_lo*W_hi, A_hi*W_lo, A_hi*W_hi) into the same
// fp32 accumulators: fp32-class accuracy from bf16 MFMAs.
// The epilogue stages each wave's accumulators through its private LDS strip so that global stores are
// row-major 16-byte (fp32) / 8-byte (bf16) vectors.
//
// Workgroup -> tile mapping is XCD-aware: the dispatcher places block b on XCD b%8 (8 private L2s), so
// each XCD is given a contiguous run of tiles (same A rows, all N tiles) to keep operand panels L2-resident.

#include "mdpt_kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    // 64 lanes x 16 B -> lds_wave_base + lane*16 (destination is wave-uniform base + lane*16)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

__device__ __forceinline__ void split_store4(bf16_t* hi, bf16_t* lo, size_t off, f32x4 v) {
    bf16x4 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (__bf16)v[e];
    *(bf16x4*)(hi + off) = h;
    if (lo) {
        bf16x4 l;
#pragma unroll
        for (int e = 0; e < 4; ++e) l[e] = (__bf16)(v[e] - (float)h[e]);
        *(bf16x4*)(lo + off) = l;
    }
}

template <int BM, int BN, int WM, int WN, int AMODE, int EKIND>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int CA = BM / 8 / NW, CB = BN / 8 / NW;  // 1-KiB LDS-DMA chunks (8 rows x 128 B) per wave
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "chunk split");
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
    static_assert(32 * WTN * 4 * NW <= 2 * STAGE, "epilogue strip must fit in the ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- XCD-aware tile mapping (bijective for any grid size)
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tile_n = swz % tiles_n, tile_m = swz / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-lane staging addresses. Lane feeds LDS row (chunk*8 + lane>>3), slot (lane&7) of that row,
    //      which must hold global 16-B chunk (slot ^ swizzle(row)).
    const int lrow = lane >> 3, slot = lane & 7;
    const int sw_stage = ((wave & 1) * 4 + (lrow >> 1)) & 7;  // ((row>>1)&7) for row = (wave + NW*i)*8 + lrow
    const int koff = (slot ^ sw_stage) * 8;                   // element offset inside the 64-wide K slab

    size_t a_off[CA];
    int a_pix[CA], a_y[CA], a_x[CA];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        int m = m0 + (wave + NW * i) * 8 + lrow;
        m = m < p.M ? m : p.M - 1;  // clamp: rows past M are computed and discarded
        if (AMODE == MDPT_A_DENSE) {
            a_off[i] = (size_t)m * p.lda;
        } else if (AMODE == MDPT_A_TOKENS) {
            const int b = m / p.tok_np, t = m - b * p.tok_np;
            a_off[i] = ((size_t)b * p.tok_stride + 1 + t) * p.lda;
        } else {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, rem = m - b * hw;
            const int y = rem / p.Wo, x = rem - y * p.Wo;
            a_pix[i] = b * p.Hi * p.Wi;
            a_y[i] = y * p.cstride - 1;
            a_x[i] = x * p.cstride - 1;
            a_off[i] = 0;
        }
    }
    size_t b_off[CB];
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        int n = n0 + (wave + NW * i) * 8 + lrow;
        n = n < p.N ? n : p.N - 1;
        b_off[i] = (size_t)n * p.K;
    }

    int st_pass = 0, st_k0 = 0, st_tap = 0, st_ci = 0;
    auto issue_stage = [&](int buf) {
        const bf16_t* Ap = (p.npass == 3 && st_pass == 0) ? p.A_lo : p.A_hi;
        const bf16_t* Wp = (p.npass == 3 && st_pass == 1) ? p.W_lo : p.W_hi;
        char* sA = smem + buf * STAGE;
        char* sB = sA + A_BYTES;
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const bf16_t* src;
            if (AMODE == MDPT_A_CONV3) {
                const int ky = (st_tap * 11) >> 5, kx = st_tap - 3 * ky;
                const int iy = a_y[i] + ky, ix = a_x[i] + kx;
                const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                src = ok ? Ap + ((size_t)(a_pix[i] + iy * p.Wi + ix) * p.Cin + st_ci + koff) : p.zero_page + koff;
            } else {
                src = Ap + a_off[i] + st_k0 + koff;
            }
            glds16(src, sA + (wave + NW * i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) glds16(Wp + b_off[i] + st_k0 + koff, sB + (wave + NW * i) * 1024);
        st_k0 += 64;
        if (AMODE == MDPT_A_CONV3) {
            st_ci += 64;
            if (st_ci == p.Cin) { st_ci = 0; ++st_tap; }
        }
        if (st_k0 == p.K) { st_k0 = 0; st_tap = 0; st_ci = 0; ++st_pass; }
    };

    // ---- fragment read offsets: row = 32*blk + (lane&31), chunk = 2*kk + (lane>>5), swizzled
    const int l31 = lane & 31, half = lane >> 5;
    const int sw_frag = (l31 >> 1) & 7;
    int frag_off[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = l31 * 128 + (((kk * 2 + half) ^ sw_frag) << 4);
    const int wm = wave / WN, wn = wave % WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int total = (p.K >> 6) * p.npass;
    issue_stage(0);
    for (int t = 0; t < total; ++t) {
        __syncthreads();  // drains this wave's LDS-DMA (vmcnt(0)) and publishes every wave's tile t
        if (t + 1 < total) issue_stage((t + 1) & 1);
        const char* sA = smem + (t & 1) * STAGE + wm * WTM * 128;
        const char* sB = smem + (t & 1) * STAGE + A_BYTES + wn * WTN * 128;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8*)(sA + i * 4096 + frag_off[kk]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(const bf16x8*)(sB + j * 4096 + frag_off[kk]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();  // every wave is done reading the ring: reuse it as epilogue staging

    // ---- epilogue: per 32-row block, accumulators -> wave-private LDS strip [32][WTN] fp32 -> row-major vectors
    float* strip = (float*)smem + wave * (32 * WTN);
    constexpr int LPR = WTN / 4;     // lanes per row (each lane owns 4 consecutive columns)
    constexpr int RPP = 64 / LPR;    // rows per pass
    const int erow = lane / LPR, ecol = (lane % LPR) * 4;
    const int nbase = n0 + wn * WTN;

#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                strip[((r & 3) + 8 * (r >> 2) + 4 * half) * WTN + j * 32 + l31] = acc[i][j][r];
        const int mbase = m0 + wm * WTM + i * 32;

        if (EKIND == MDPT_E_QKV && nbase >= 2 * p.F) {
            // V columns: write transposed, Vt[(b,h,d), t..t+3] (4 consecutive tokens per lane, 8-byte stores)
#pragma unroll 2
            for (int pr = 0; pr < 8 * (WTN / 64 > 0 ? WTN / 64 : 1); ++pr) {
                const int col = (pr / 8) * 64 + lane, rg = pr % 8;
                if (col >= WTN) continue;
                const int m = mbase + rg * 4, n = nbase + col;
                if (m >= p.M || n >= p.N) continue;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = strip[(rg * 4 + e) * WTN + col] + p.bias[n];
                const int f = n - 2 * p.F, h = f >> 6, d = f & 63;
                const int b = m / p.npad, tk = m - b * p.npad;
                split_store4(p.vt_hi, p.vt_lo, ((size_t)(b * p.heads + h) * 64 + d) * p.npadv + tk, v);
            }
            continue;
        }

        for (int pr = 0; pr < 32 / RPP; ++pr) {
            const int row = pr * RPP + erow;
            const int m = mbase + row, n = nbase + ecol;
            if (m >= p.M || n >= p.N) continue;
            f32x4 v = *(const f32x4*)(strip + row * WTN + ecol);

            if (EKIND == MDPT_E_GENERIC) {
                if (p.bias) v += *(const f32x4*)(p.bias + n);
                if (p.act == MDPT_ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                } else if (p.act == MDPT_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                }
                if (p.gamma) v *= *(const f32x4*)(p.gamma + n);
                if (p.resid) v += *(const f32x4*)(p.resid + (size_t)m * p.ldr + n);
                if (p.up_src) {
                    // + bilinear x2 (align_corners=True) of the previous fusion level (fusion_model.py:151,178)
                    const int hw = p.Ho * p.Wo;
                    const int b = m / hw, rem = m - b * hw;
                    const int y = rem / p.Wo, x = rem - y * p.Wo;
                    const float sy = (float)(p.Hu - 1) / (float)(p.Ho - 1) * (float)y;
                    const float sx = (float)(p.Wu - 1) / (float)(p.Wo - 1) * (float)x;
                    const int y0 = (int)sy, x0 = (int)sx;
                    const int y1 = y0 + (y0 < p.Hu - 1), x1 = x0 + (x0 < p.Wu - 1);
                    const float ly = sy - (float)y0, lx = sx - (float)x0;
                    const float* base = p.up_src + (size_t)b * p.Hu * p.Wu * p.N + n;
                    const f32x4 v00 = *(const f32x4*)(base + ((size_t)y0 * p.Wu + x0) * p.N);
                    const f32x4 v01 = *(const f32x4*)(base + ((size_t)y0 * p.Wu + x1) * p.N);
                    const f32x4 v10 = *(const f32x4*)(base + ((size_t)y1 * p.Wu + x0) * p.N);
                    const f32x4 v11 = *(const f32x4*)(base + ((size_t)y1 * p.Wu + x1) * p.N);
                    v += (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
                }
                const size_t o = (size_t)m * p.ldc + n;
                if (p.out_f32) *(f32x4*)(p.out_f32 + o) = v;
                if (p.out_hi) {
                    if (p.relu_bf16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.0f);
                    }
                    split_store4(p.out_hi, p.out_lo, o, v);
                }
            } else if (EKIND == MDPT_E_QKV) {
                // Q (pre-scaled by 1/sqrt(d), exact power of two) and K, head-major [B,H,npad,64]
                v += *(const f32x4*)(p.bias + n);
                const int which = n >= p.F;
                const int f = n - which * p.F, h = f >> 6, d = f & 63;
                const int b = m / p.npad, tk = m - b * p.npad;
                const size_t o = ((size_t)(b * p.heads + h) * p.npad + tk) * 64 + d;
                if (!which) {
                    v *= p.qscale;
                    split_store4(p.q_hi, p.q_lo, o, v);
                } else {
                    split_store4(p.k_hi, p.k_lo, o, v);
                }
            } else if (EKIND == MDPT_E_PATCH) {
                const int b = m / p.tok_np, t = m - b * p.tok_np;
                v += *(const f32x4*)(p.bias + n);
                v += *(const f32x4*)(p.pos + (size_t)t * p.N + n);
                *(f32x4*)(p.out_f32 + ((size_t)b * p.npad + 1 + t) * p.ldc + n) = v;
            } else if (EKIND == MDPT_E_D2S) {
                const int kk2 = p.d2s_k * p.d2s_k;
                const int kidx = n / p.d2s_cout, co = n - kidx * p.d2s_cout;
                const int ky = kidx / p.d2s_k, kx = kidx - ky * p.d2s_k;
                const int hw = p.Ho * p.Wo;
                const int b = m / hw, rem = m - b * hw;
                const int y = rem / p.Wo, x = rem - y * p.Wo;
                (void)kk2;
                v += *(const f32x4*)(p.bias + co);
                const size_t o =
                    (((size_t)b * p.Ho * p.d2s_k + (y * p.d2s_k + ky)) * (p.Wo * p.d2s_k) + (x * p.d2s_k + kx)) * p.d2s_cout + co;
                split_store4(p.out_hi, p.out_lo, o, v);
            } else if (EKIND == MDPT_E_HEAD) {
                // relu(conv3x3 -> 32) . w[32] + b -> relu | sigmoid   (head_model.py:80-85)
                v += *(const f32x4*)(p.bias + n);
                const f32x4 w4 = *(const f32x4*)(p.head_w + n);
                float s = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) s += fmaxf(v[e], 0.0f) * w4[e];
#pragma unroll
                for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o);
                if ((lane % LPR) == 0) {
                    s += p.head_b[0];
                    p.head_out[m] = p.head_sigmoid ? 1.0f / (1.0f + __expf(-s)) : fmaxf(s, 0.0f);
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int AMODE, int EKIND>
int launch_cfg(const GemmParams& p, hipStream_t stream) {
    constexpr int LDS = 2 * (BM + BN) * 128;
    static bool attr_done = false;
    auto kern = gemm_kernel<BM, BN, WM, WN, AMODE, EKIND>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(64 * WM * WN), LDS, stream, p);
    return (int)hipGetLastError();
}

template <int AMODE, int EKIND>
int launch_tile(const GemmParams& p, hipStream_t stream) {
    int tile = p.tile;
    if (tile == MDPT_TILE_AUTO) {
        // 256x256 pays once there are enough big tiles to fill 256 CUs; otherwise 128x128
        const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
        tile = (p.N % 256 == 0 && tiles256 >= 512) ? MDPT_TILE_256x256 : MDPT_TILE_128x128;
    }
    if (tile == MDPT_TILE_256x256) return launch_cfg<256, 256, 2, 4, AMODE, EKIND>(p, stream);
    return launch_cfg<128, 128, 2, 2, AMODE, EKIND>(p, stream);
}

}  // namespace

int mdpt_launch_gemm(const GemmParams& p, hipStream_t stream) {
    if (p.M <= 0 || p.N <= 0) return 0;
    if (p.K <= 0 || (p.K & 63) || (p.N & 3)) return (int)hipErrorInvalidValue;
    if (p.npass != 1 && p.npass != 3) return (int)hipErrorInvalidValue;
    switch (p.ekind) {
        case MDPT_E_GENERIC:
            if (p.amode == MDPT_A_DENSE) return launch_tile<MDPT_A_DENSE, MDPT_E_GENERIC>(p, stream);
            if (p.amode == MDPT_A_TOKENS) return launch_tile<MDPT_A_TOKENS, MDPT_E_GENERIC>(p, stream);
            if (p.amode == MDPT_A_CONV3) return launch_tile<MDPT_A_CONV3, MDPT_E_GENERIC>(p, stream);
            break;
        case MDPT_E_QKV:
            if (p.amode == MDPT_A_DENSE && (p.F & 63) == 0 && (p.npad & 3) == 0)
                return launch_tile<MDPT_A_DENSE, MDPT_E_QKV>(p, stream);
            break;
        case MDPT_E_PATCH:
            if (p.amode == MDPT_A_DENSE) return launch_tile<MDPT_A_DENSE, MDPT_E_PATCH>(p, stream);
            break;
        case MDPT_E_D2S:
            if (p.amode == MDPT_A_DENSE && (p.d2s_cout & 3) == 0) return launch_tile<MDPT_A_DENSE, MDPT_E_D2S>(p, stream);
            break;
        case MDPT_E_HEAD:
            if (p.amode == MDPT_A_CONV3 && p.N == 32) return launch_cfg<128, 32, 4, 1, MDPT_A_CONV3, MDPT_E_HEAD>(p, stream);
            break;
    }
    return (int)hipErrorInvalidValue;
}

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
// Two main-loop families (picked per problem by launch_tile):
//   * gemm8_kernel: 256x256x64 tiles, 8 waves in two groups staggered by one barrier, 16x16x32 MFMA quadrants, half-tile DMA
//     prefetch with a counted vmcnt, direct register->global epilogues for the hot encoder shapes - the big-problem kernel
//     (see its own header further down);
//   * gemm_kernel ("lockstep"): BMxBNx64 tiles for small / narrow problems and odd K-tile counts, described next.
//
// Lockstep structure (per workgroup): BMxBNx64 tiles, 64-lane waves each owning a (BM/WM)x(BN/WN) sub-tile of
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
// K split (GemmParams::ksplit): K in equal ranges over grid.y. Three forms - (i) the consumer adds the partial planes (the LayerNorm behind a
// residual GEMM: 64x64 tile, and the DM_F32 form of the 8-phase kernel for SwinV2's fc2), (ii) the kernel reduces itself, last workgroup to
// arrive, fixed order (GemmParams::ks_ctr: 64x64 tile, any generic epilogue; latency mode), (iii) none. A split is fixed per shape.
//
// Workgroup -> tile mapping is XCD-aware: the dispatcher places block b on XCD b%8 (8 private L2s), so
// each XCD is given a contiguous run of tiles (same A rows, all N tiles) to keep operand panels L2-resident.

#include "mdpt_kernels.h"
#include "mdpt_prof.h"
#include <stdio.h>
#include <stdlib.h>

// Results must not depend on which tile instantiation a launch picks (the tile is chosen from the batch size, and
// data-parallel sharding must reproduce the single-GPU result bit for bit): with the default fp-contract=fast the
// compiler fuses the epilogue's mul/add chains differently per instantiation (seen: 2e-6 differences in the
// bilinear-add epilogue between the 128x128 and 256x256 kernels). Epilogue arithmetic is a negligible cost.
#pragma clang fp contract(off)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
    // 64 lanes x 16 B -> lds_wave_base + lane*16 (destination is wave-uniform base + lane*16)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ unsigned long long memtime_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// Exact-GELU 0.5 v (1 + erf(v/sqrt2)) (reference: nn.GELU() default, components/misc_helpers.py:113) with erf from
// Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute) on the hardware rcp/exp units: libm's erff costs ~180 us per
// fc1 launch (measured 618 us vs 438 us without it), this form a handful of FMAs.
__device__ __forceinline__ float gelu_erf(float v) {
    // exact (erf) GELU, erf by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7). The file is compiled with fp-contraction off (so that no
    // template instantiation fuses differently from another); the multiply-adds of this function are EXPLICIT fmas instead - the same
    // bits in every instantiation, 7 VALU operations fewer per value in the VALU-bound fc1 epilogue.
    const float x = fabsf(v) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, x, 1.0f));
    float poly = 1.061405429f;
    poly = __builtin_fmaf(poly, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-x * x * 1.4426950408889634f);
    const float erf_abs = __builtin_fmaf(-(poly * t), e, 1.0f);
    const float h = 0.5f * v;
    return __builtin_fmaf(h, copysignf(erf_abs, v), h);
}

// The same function on PAIRS of values: every multiply / fma becomes one packed instruction (v_pk_mul_f32 / v_pk_fma_f32: two lanes of
// IEEE fp32, the same bits as the scalar form above), only rcp / exp2 stay scalar. Halves the VALU issue slots of the fc1 epilogue.
typedef __attribute__((ext_vector_type(2))) float gelu_f32x2;
__device__ __forceinline__ gelu_f32x2 gelu_erf2(gelu_f32x2 v) {
    const gelu_f32x2 x = gelu_f32x2{fabsf(v[0]), fabsf(v[1])} * 0.70710678118654752f;
    const gelu_f32x2 d = __builtin_elementwise_fma(gelu_f32x2{0.3275911f, 0.3275911f}, x, gelu_f32x2{1.0f, 1.0f});
    const gelu_f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    gelu_f32x2 poly = {1.061405429f, 1.061405429f};
    poly = __builtin_elementwise_fma(poly, t, gelu_f32x2{-1.453152027f, -1.453152027f});
    poly = __builtin_elementwise_fma(poly, t, gelu_f32x2{1.421413741f, 1.421413741f});
    poly = __builtin_elementwise_fma(poly, t, gelu_f32x2{-0.284496736f, -0.284496736f});
    poly = __builtin_elementwise_fma(poly, t, gelu_f32x2{0.254829592f, 0.254829592f});
    const gelu_f32x2 a = -x * x * 1.4426950408889634f;
    const gelu_f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
    const gelu_f32x2 erf_abs = __builtin_elementwise_fma(-(poly * t), e, gelu_f32x2{1.0f, 1.0f});
    const gelu_f32x2 h = v * 0.5f;
    const gelu_f32x2 sgn = {copysignf(erf_abs[0], v[0]), copysignf(erf_abs[1], v[1])};
    return __builtin_elementwise_fma(h, sgn, h);
}

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

__device__ __forceinline__ void split_store8(op_t* hi, op_t* lo, size_t off, f32x4 v0, f32x4 v1) {
    opx8 h;
#pragma unroll
    for (int e = 0; e < 4; ++e) { h[e] = to_op(v0[e]); h[e + 4] = to_op(v1[e]); }
    *(opx8*)(hi + off) = h;
    if (lo) {
        opx8 l;
#pragma unroll
        for (int e = 0; e < 4; ++e) { l[e] = to_op(v0[e] - (float)h[e]); l[e + 4] = to_op(v1[e] - (float)h[e + 4]); }
        *(opx8*)(lo + off) = l;
    }
}

// ------------------------------------------------------------------------------------------------------------
// Operand stager: per-lane source addresses of this wave's LDS-DMA chunks for one (BM+BN) x BK slab, and the
// state machine that walks K (dense/token rows: a running pointer; 3x3 conv: tap / channel counters) and the
// bf16x3 operand planes.
// A DMA chunk (one global_load_lds_dwordx4 wave instruction, 1 KiB) is RPC rows x ROWB bytes; the lane feeds LDS
// row (chunk*RPC + lane/CPR), 16-B slot (lane%CPR), which must hold global chunk (slot ^ key(row)) with
// key(row) = (row / RPB) & (CPR-1): 16 consecutive rows then cover the 16 slots of a 256-B LDS bank row, so the
// ds_read_b128 fragment pattern (16 rows x 16 B per service group) is conflict-free.
// ------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int NW, int BK, int AMODE>
struct Stager {
    static constexpr int ROWB = BK * 2, CPR = BK / 8, RPC = 64 / CPR, RPB = 256 / ROWB;
    static constexpr int CA = BM / RPC / NW, CB = BN / RPC / NW, NLOAD = CA + CB;
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, SLAB = A_BYTES + B_BYTES;
    static_assert(BM % (RPC * NW) == 0 && BN % (RPC * NW) == 0, "chunk split");

    const op_t* a_ptr[CA];
    int a_pix[CA], a_y[CA], a_x[CA], a_ko[CA];
    const op_t* b_ptr[CB];
    ptrdiff_t a_hi_minus_lo, w_lo_minus_hi;
    const op_t* conv_plane;
    int st_pass, st_k0, st_tap, st_ci, st_sub;  // conv K order: 64-channel block (st_ci) outer, tap, then the BK-wide part of the block (st_sub)
    int kspan;                                  // K extent this workgroup walks per pass (p.K, or p.K / ksplit from kofs on: GemmParams::ksplit)
    int st_tap0, st_ci0;                        // conv state at the start of the range

    __device__ __forceinline__ void init(const GemmParams& p, int m0, int n0, int wave, int lane, int kofs = 0, int kspan_ = 0) {
        kspan = kspan_ > 0 ? kspan_ : p.K;
        const int lrow = lane / CPR, slot = lane % CPR;
        const op_t* A0 = p.npass == 3 ? p.A_lo : p.A_hi;  // operand plane of pass 0
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const int r = (wave + NW * i) * RPC + lrow;
            const int koff = (slot ^ ((r / RPB) & (CPR - 1))) * 8;
            int m = m0 + r;
            m = m < p.M ? m : p.M - 1;  // clamp: rows past M are computed and discarded
            a_pix[i] = a_y[i] = a_x[i] = 0;
            a_ko[i] = koff;
            a_ptr[i] = nullptr;
            if (AMODE == MDPT_A_DENSE) {
                a_ptr[i] = A0 + (size_t)m * p.lda + koff + kofs;
            } else if (AMODE == MDPT_A_TOKENS) {
                const int b = m / p.tok_np, t = m - b * p.tok_np;
                a_ptr[i] = A0 + ((size_t)b * p.tok_stride + 1 + t) * p.lda + koff + kofs;
            } else {
                const int hw = p.Ho * p.Wo;
                const int b = m / hw, rem = m - b * hw;
                const int y = rem / p.Wo, x = rem - y * p.Wo;
                a_pix[i] = b * p.Hi * p.Wi;
                a_y[i] = y * p.cstride - 1;
                a_x[i] = x * p.cstride - 1;
            }
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int r = (wave + NW * i) * RPC + lrow;
            const int koff = (slot ^ ((r / RPB) & (CPR - 1))) * 8;
            int n = n0 + r;
            n = n < p.N ? n : p.N - 1;
            b_ptr[i] = p.W_hi + (size_t)n * p.ldw + koff + kofs;
        }
        // plane switches at pass roll-over (bf16x3 passes: A_lo*W_hi, A_hi*W_lo, A_hi*W_hi); all wave-uniform
        a_hi_minus_lo = p.npass == 3 ? p.A_hi - p.A_lo : 0;
        w_lo_minus_hi = p.npass == 3 ? p.W_lo - p.W_hi : 0;
        conv_plane = A0;
        st_pass = st_k0 = st_sub = 0;
        // a K range that starts at kofs (a multiple of 64): conv K order k = (cb * 9 + tap) * 64 + c
        st_tap0 = AMODE == MDPT_A_CONV3 ? (kofs >> 6) % 9 : 0;
        st_ci0 = AMODE == MDPT_A_CONV3 ? ((kofs >> 6) / 9) * 64 : 0;
        st_tap = st_tap0; st_ci = st_ci0;
    }

    // issue this wave's NLOAD LDS-DMA instructions for the next slab into the ring slot at `slab_base`, then advance
    __device__ __forceinline__ void issue(const GemmParams& p, char* slab_base, int wave) {
        char* sA = slab_base;
        char* sB = slab_base + A_BYTES;
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const op_t* src;
            if (AMODE == MDPT_A_CONV3) {
                const int ky = (st_tap * 11) >> 5, kx = st_tap - 3 * ky;
                const int iy = a_y[i] + ky, ix = a_x[i] + kx;
                const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                src = ok ? conv_plane + ((size_t)(a_pix[i] + iy * p.Wi + ix) * p.Cin + st_ci + st_sub + a_ko[i]) : p.zero_page + a_ko[i];
            } else {
                src = a_ptr[i];
                a_ptr[i] += BK;
            }
            glds16(src, sA + (wave + NW * i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            glds16(b_ptr[i], sB + (wave + NW * i) * 1024);
            b_ptr[i] += BK;
        }
        st_k0 += BK;
        if (AMODE == MDPT_A_CONV3) {  // k = (cb * 9 + tap) * 64 + c: the nine taps of a 64-channel block are consecutive K tiles
            st_sub += BK;
            if (st_sub == 64) {
                st_sub = 0;
                if (++st_tap == 9) { st_tap = 0; st_ci += 64; }
            }
        }
        if (st_k0 == kspan) {  // next pass: rewind K and switch operand planes
            st_k0 = 0; st_tap = st_tap0; st_ci = st_ci0; st_sub = 0;
            const ptrdiff_t da = (st_pass == 0 ? a_hi_minus_lo : 0) - kspan;
            const ptrdiff_t dw = (st_pass == 0 ? w_lo_minus_hi : -w_lo_minus_hi) - kspan;
            if (st_pass == 0) conv_plane = p.A_hi;
#pragma unroll
            for (int i = 0; i < CA; ++i)
                if (AMODE != MDPT_A_CONV3) a_ptr[i] += da;
#pragma unroll
            for (int i = 0; i < CB; ++i) b_ptr[i] += dw;
            ++st_pass;
        }
    }
};

// XCD-aware tile mapping (bijective for any grid size): the dispatcher places block b on XCD b%8; give every XCD a
// contiguous run of tiles (same A rows, all N tiles) so operand panels stay in that XCD's L2.
__device__ __forceinline__ void tile_coords(int tiles_n, int BM, int BN, int& m0, int& n0) {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    if (tiles_n >= 8 && (tiles_n & 3) == 0) {
        // wide outputs (fc1: 16 column tiles, QKV: 12): row-major order makes the 32 concurrent tiles of an XCD cover 2 row
        // blocks x ALL columns, i.e. the whole weight matrix (8 MB for fc1, twice the 4 MB L2) is re-streamed for every pair
        // of row blocks (measured 969 MB fetched per fc1 launch vs 93 MB algorithmic). Walk 4-column panels instead: the
        // panel's weights (2 MB at K = 1024) stay in L2 while the row blocks stream past once per panel.
        const int tiles_m = nwg / tiles_n, per_panel = tiles_m * 4;
        const int panel = swz / per_panel, r = swz - panel * per_panel;
        m0 = (r >> 2) * BM;
        n0 = (panel * 4 + (r & 3)) * BN;
        return;
    }
    m0 = (swz / tiles_n) * BM;
    n0 = (swz % tiles_n) * BN;
}

// ------------------------------------------------------------------------------------------------------------
// Epilogue shared by both main-loop variants: per 32-row block, accumulators -> wave-private LDS strip [32][WTN] fp32
// -> row-major vectors. The vector-memory path moves 64 B/clk per CU and a store instruction costs about the same address
// processing whatever its width, so every store is 16 bytes per lane: each lane owns 8 consecutive columns (one opx8 store,
// two fp32x4 stores). Stores here sit behind per-lane bounds checks (divergent branches), which makes hipcc wait for every
// store's acknowledgement before the next one (see epilogue_direct for the branch-free form used on the big GEMMs).
// ------------------------------------------------------------------------------------------------------------
// One [32 rows][WTN cols] fp32 block, already transposed into the wave-private LDS `strip`, -> global memory.
template <int WTN, int EKIND>
__device__ __forceinline__ void epilogue_block(const GemmParams& p, const float* strip, int lane, int mbase, int nbase);

template <int WTN, int TM, int TN, int EKIND>
__device__ __forceinline__ void run_epilogue(const GemmParams& p, f32x16 (&acc)[TM][TN], char* smem, int wave, int lane, int mwave0,
                                             int nbase) {
    const int l31 = lane & 31, half = lane >> 5;
    float* strip = (float*)smem + wave * (32 * WTN);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                strip[((r & 3) + 8 * (r >> 2) + 4 * half) * WTN + j * 32 + l31] = acc[i][j][r];
        epilogue_block<WTN, EKIND>(p, strip, lane, mwave0 + i * 32, nbase);
    }
}

template <int WTN, int EKIND>
__device__ __forceinline__ void epilogue_block(const GemmParams& p, const float* strip, int lane, int mbase, int nbase) {
    constexpr int LPR = WTN / 8;     // lanes per row
    constexpr int RPP = 64 / LPR;    // rows per pass
    const int erow = lane / LPR, ecol = (lane % LPR) * 8;
    {

        if (EKIND == MDPT_E_QKV && nbase >= 2 * p.F) {
            // V columns: write transposed, Vt[(b,h,d), t..t+7] (8 consecutive tokens per lane, 16-byte stores)
            for (int pr = 0; pr < WTN * 4 / 64; ++pr) {  // WTN columns x 4 groups of 8 rows, 64 items per pass
                const int item = pr * 64 + lane;
                const int col = item % WTN, rg = item / WTN;
                const int m = mbase + rg * 8, n = nbase + col;
                if (m >= p.M || n >= p.N) continue;
                const float bz = p.bias_img_stride ? p.bias[(size_t)(m / p.bias_img_rows) * p.bias_img_stride + n] : p.bias[n];  // 8 rows, one image
                f32x4 v0, v1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = strip[(rg * 8 + e) * WTN + col] + bz;
                    v1[e] = strip[(rg * 8 + 4 + e) * WTN + col] + bz;
                }
                const int f = n - 2 * p.F, h = f >> 6, d = f & 63;
                const int b = m / p.npad, tk = m - b * p.npad;
                split_store8(p.vt_hi, p.vt_lo, ((size_t)(b * p.heads + h) * 64 + d) * p.npadv + tk, v0, v1);
            }
            return;
        }

        if (EKIND == MDPT_E_GENERIC) {
            // Every global load of the block is issued before the first use (bias / layer scale once per column group, the residual
            // of all passes up front, addresses clamped for out-of-range lanes): one memory round trip per block instead of one per
            // feature and pass - these epilogues dominate the short-K tiles of the small-batch path. Arithmetic order unchanged.
            constexpr int NP = 32 / RPP;
            const int n = nbase + ecol;
            const bool nvalid = n < p.N;
            const int nc = nvalid ? n : 0;
            const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
            f32x4 bia[2] = {zero4, zero4}, gam[2] = {zero4, zero4};
            if (p.bias && !p.bias_img_stride) { bia[0] = *(const f32x4*)(p.bias + nc); bia[1] = *(const f32x4*)(p.bias + nc + 4); }
            if (p.gamma) { gam[0] = *(const f32x4*)(p.gamma + nc); gam[1] = *(const f32x4*)(p.gamma + nc + 4); }
            f32x4 res[NP][2], bim[NP][2];
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                const int mq = mbase + pr * RPP + erow;
                const int mc = mq < p.M ? mq : p.M - 1;
                res[pr][0] = res[pr][1] = bim[pr][0] = bim[pr][1] = zero4;
                if (p.resid && !p.acc_init) {
                    const float* rp = p.resid + (size_t)mc * p.ldr + nc;
                    res[pr][0] = *(const f32x4*)rp;
                    res[pr][1] = *(const f32x4*)(rp + 4);
                }
                if (p.bias && p.bias_img_stride) {
                    const float* bp = p.bias + nc + (size_t)(mc / p.bias_img_rows) * p.bias_img_stride;
                    bim[pr][0] = *(const f32x4*)bp;
                    bim[pr][1] = *(const f32x4*)(bp + 4);
                }
            }
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
                const int row = pr * RPP + erow;
                const int m = mbase + row;
                if (m >= p.M || !nvalid) continue;
                f32x4 v[2];
                v[0] = *(const f32x4*)(strip + row * WTN + ecol);
                v[1] = *(const f32x4*)(strip + row * WTN + ecol + 4);
                if (p.bias) {
                    v[0] += p.bias_img_stride ? bim[pr][0] : bia[0];
                    v[1] += p.bias_img_stride ? bim[pr][1] : bia[1];
                }
                if (p.act == MDPT_ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[0][e] = gelu_erf(v[0][e]); v[1][e] = gelu_erf(v[1][e]); }
                } else if (p.act == MDPT_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[0][e] = fmaxf(v[0][e], 0.0f); v[1][e] = fmaxf(v[1][e], 0.0f); }
                }
                if (p.gamma) { v[0] *= gam[0]; v[1] *= gam[1]; }
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
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x4 v00 = *(const f32x4*)(base + ((size_t)y0 * p.Wu + x0) * p.N + 4 * q);
                        const f32x4 v01 = *(const f32x4*)(base + ((size_t)y0 * p.Wu + x1) * p.N + 4 * q);
                        const f32x4 v10 = *(const f32x4*)(base + ((size_t)y1 * p.Wu + x0) * p.N + 4 * q);
                        const f32x4 v11 = *(const f32x4*)(base + ((size_t)y1 * p.Wu + x1) * p.N + 4 * q);
                        v[q] += (1.0f - ly) * ((1.0f - lx) * v00 + lx * v01) + ly * ((1.0f - lx) * v10 + lx * v11);
                    }
                }
                // order: ((acc + bias) [* gamma] [+ up]) + resid - the halo-staged conv kernel (conv3h.hip) applies the same one
                if (p.resid && !p.acc_init) { v[0] += res[pr][0]; v[1] += res[pr][1]; }
                const size_t o = (size_t)m * p.ldc + n;
                if (p.out_f32) { *(f32x4*)(p.out_f32 + o) = v[0]; *(f32x4*)(p.out_f32 + o + 4) = v[1]; }
                if (p.out_hi) {
                    if (p.relu_bf16) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v[0][e] = fmaxf(v[0][e], 0.0f); v[1][e] = fmaxf(v[1][e], 0.0f); }
                    }
                    split_store8(p.out_hi, p.out_lo, o, v[0], v[1]);
                }
            }
            return;
        }

        for (int pr = 0; pr < 32 / RPP; ++pr) {
            const int row = pr * RPP + erow;
            const int m = mbase + row, n = nbase + ecol;
            if (m >= p.M || n >= p.N) continue;
            f32x4 v[2];
            v[0] = *(const f32x4*)(strip + row * WTN + ecol);
            v[1] = *(const f32x4*)(strip + row * WTN + ecol + 4);

            if (EKIND == MDPT_E_QKV) {
                // Q (pre-scaled by 1/sqrt(d), exact power of two) and K, head-major [B,H,npad,64]
                const float* bq = p.bias_img_stride ? p.bias + (size_t)(m / p.bias_img_rows) * p.bias_img_stride : p.bias;
                v[0] += *(const f32x4*)(bq + n);
                v[1] += *(const f32x4*)(bq + n + 4);
                const int which = n >= p.F;
                const int f = n - which * p.F, h = f >> 6, d = f & 63;
                const int b = m / p.npad, tk = m - b * p.npad;
                const size_t o = ((size_t)(b * p.heads + h) * p.npad + tk) * 64 + d;
                if (!which) {
                    v[0] *= p.qscale; v[1] *= p.qscale;
                    split_store8(p.q_hi, p.q_lo, o, v[0], v[1]);
                } else {
                    split_store8(p.k_hi, p.k_lo, o, v[0], v[1]);
                }
            } else if (EKIND == MDPT_E_PATCH) {
                const int b = m / p.tok_np, t = m - b * p.tok_np;
                v[0] += *(const f32x4*)(p.bias + n);
                v[1] += *(const f32x4*)(p.bias + n + 4);
                if (p.pos) {  // BEiT has no absolute position embedding
                    const float* pp = p.pos + (size_t)t * p.N + n;
                    v[0] += *(const f32x4*)pp;
                    v[1] += *(const f32x4*)(pp + 4);
                }
                float* op = p.out_f32 + ((size_t)b * p.npad + 1 + t) * p.ldc + n;
                *(f32x4*)op = v[0];
                *(f32x4*)(op + 4) = v[1];
            } else if (EKIND == MDPT_E_D2S) {
                const int kidx = n / p.d2s_cout, co = n - kidx * p.d2s_cout;
                const int ky = kidx / p.d2s_k, kx = kidx - ky * p.d2s_k;
                const int hw = p.Ho * p.Wo;
                const int b = m / hw, rem = m - b * hw;
                const int y = rem / p.Wo, x = rem - y * p.Wo;
                v[0] += *(const f32x4*)(p.bias + co);
                v[1] += *(const f32x4*)(p.bias + co + 4);
                const size_t o =
                    (((size_t)b * p.Ho * p.d2s_k + (y * p.d2s_k + ky)) * (p.Wo * p.d2s_k) + (x * p.d2s_k + kx)) * p.d2s_cout + co;
                split_store8(p.out_hi, p.out_lo, o, v[0], v[1]);
            } else if (EKIND == MDPT_E_HEAD) {
                // relu(conv3x3 -> 32) . w[32] + b -> relu | sigmoid   (head_model.py:80-85)
                v[0] += *(const f32x4*)(p.bias + n);
                v[1] += *(const f32x4*)(p.bias + n + 4);
                const f32x4 w0 = *(const f32x4*)(p.head_w + n), w1 = *(const f32x4*)(p.head_w + n + 4);
                float sacc = 0.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) sacc += fmaxf(v[0][e], 0.0f) * w0[e];
#pragma unroll
                for (int e = 0; e < 4; ++e) sacc += fmaxf(v[1][e], 0.0f) * w1[e];
#pragma unroll
                for (int o = 1; o < LPR; o <<= 1) sacc += __shfl_xor(sacc, o);
                if ((lane % LPR) == 0) {
                    sacc += p.head_b[0];
                    const float dv = p.head_sigmoid ? 1.0f / (1.0f + __expf(-sacc)) : fmaxf(sacc, 0.0f);
                    // depth leaves in the caller's dtype (the reference returns the model dtype, dpt_model.py:105-107)
                    if (p.head_out_dtype == MDPT_DT_BF16) ((__bf16*)p.head_out)[m] = (__bf16)dv;
                    else if (p.head_out_dtype == MDPT_DT_F16) ((_Float16*)p.head_out)[m] = (_Float16)dv;
                    else ((float*)p.head_out)[m] = dv;
                }
            }
        }
    }
}

// K-split partial sums (GemmParams::ksplit, ranges z >= 1): the bare accumulators as fp32 rows of `out` (row stride ldc), through the same
// wave-private strip; no bias, no activation - the consumer adds the partials to range 0's output in a fixed order.
template <int WTN, int TM, int TN>
__device__ __forceinline__ void run_epilogue_partial(const GemmParams& p, f32x16 (&acc)[TM][TN], char* smem, int wave, int lane, int mwave0,
                                                     int nbase, float* out) {
    constexpr int LPR = WTN / 8, RPP = 64 / LPR;
    const int l31 = lane & 31, half = lane >> 5;
    const int erow = lane / LPR, ecol = (lane % LPR) * 8;
    float* strip = (float*)smem + wave * (32 * WTN);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                strip[((r & 3) + 8 * (r >> 2) + 4 * half) * WTN + j * 32 + l31] = acc[i][j][r];
#pragma unroll
        for (int pr = 0; pr < 32 / RPP; ++pr) {
            const int row = pr * RPP + erow;
            const int m = mwave0 + i * 32 + row, n = nbase + ecol;
            if (m >= p.M || n >= p.N) continue;
            float* o = out + (size_t)m * p.ldc + n;
            *(f32x4*)o = *(const f32x4*)(strip + row * WTN + ecol);
            *(f32x4*)(o + 4) = *(const f32x4*)(strip + row * WTN + ecol + 4);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Variant A ("lockstep"): all waves run  wait-DMA -> barrier -> issue next DMA -> LDS reads -> MFMAs  together.
// Used for small tiles (128x128, 2 workgroups per CU) and the 32-wide head tile.
// ------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int BK, int NST, int MINW, int AMODE, int EKIND>
__global__ __launch_bounds__(64 * WM * WN, MINW) void gemm_kernel(const GemmParams p) {
    constexpr int NW = WM * WN;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int TM = WTM / 32, TN = WTN / 32;
    using St = Stager<BM, BN, NW, BK, AMODE>;
    constexpr int ROWB = St::ROWB, CPR = St::CPR, RPB = St::RPB, NLOAD = St::NLOAD, A_BYTES = St::A_BYTES, STAGE = St::SLAB;
    constexpr int KSTEPS = BK / 16;     // 32x32x16 MFMA k-steps per slab
    static_assert(BK == 32 || BK == 64, "BK");
    static_assert(NST == 2 || NST == 3, "ring depth");
    static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
    static_assert(32 * WTN * 4 * NW <= NST * STAGE, "epilogue strip must fit in the ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long t_start = 0, t_first = 0, t_loop = 0;
    if (p.dbg_times) t_start = memtime_now();

    int m0, n0;
    tile_coords((p.N + BN - 1) / BN, BM, BN, m0, n0);
    // K split (GemmParams::ksplit; the launcher sets grid.y for the 64x64 dense / generic instantiations only): range z of the K axis
    constexpr bool KSPLIT = EKIND == MDPT_E_GENERIC && BM == 64 && BN == 64 && BK == 64;
    const int kz = KSPLIT ? (int)blockIdx.y : 0;
    const int kspan = KSPLIT && p.ksplit > 1 ? p.K / p.ksplit : p.K;
    St st;
    st.init(p, m0, n0, wave, lane, kz * kspan, kspan);

    // ---- fragment read offsets: row = 32*blk + (lane&31), chunk = 2*kk + (lane>>5), swizzled with key(row)
    const int l31 = lane & 31, half = lane >> 5;
    const int sw_frag = (l31 / RPB) & (CPR - 1);
    int frag_off[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) frag_off[kk] = l31 * ROWB + (((kk * 2 + half) ^ sw_frag) << 4);
    const int wm = wave / WN, wn = wave % WN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    if (EKIND == MDPT_E_GENERIC && p.acc_init && kz == 0) {
        // residual GEMMs (out = resid + A W^T + bias, in place): the accumulators START at the residual, the epilogue adds the bias and
        // stores - the same order of operations in every tile variant (see gemm8_body's RI form, where this hides the residual read
        // under the main loop). acc[i][j][r] = C[32 i + (r&3) + 8 (r>>2) + 4 half][32 j + (lane&31)]; out-of-range elements read 0.
        // descriptor over THIS tile's rows (base = row m0, byte offsets inside the tile: < BM * ldr * 4, so no 32-bit wrap however large M is)
        const int mw = m0 + (wave / WN) * WTM, nw = n0 + (wave % WN) * WTN;
        const size_t tile_bytes = (size_t)(p.M - m0 < BM ? p.M - m0 : BM) * p.ldr * 4;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid + (size_t)m0 * p.ldr), 0, (int)(unsigned)(tile_bytes < 0xFFFFFFF0ull ? tile_bytes : 0xFFFFFFF0ull), 0x00020000);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = nw + j * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const unsigned off = (n < p.N && m < p.M) ? ((unsigned)(m - m0) * (unsigned)p.ldr + (unsigned)n) * 4u : 0xFFFFFFF0u;
                    acc[i][j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
                }
            }
    }

    // ---- main loop: NST-deep LDS ring. Iteration t: wait for THIS wave's DMA of slab t (counted vmcnt: with a
    //      3-deep ring the DMA of slab t+1 stays in flight across the barrier), barrier (everybody's part of slab t
    //      has landed AND everybody finished reading slab t-1), refill the slot of slab t-1 with slab t+NST-1, compute.
    //      LDS-DMA completion is only tracked by vmcnt: the waits are explicit (hipcc does not reliably insert them).
    const int total = (kspan / BK) * p.npass;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < total) st.issue(p, smem + s * STAGE, wave);
    int rd = 0, wr = NST - 1;
    for (int t = 0; t < total; ++t) {
        if (NST == 3 && t + 1 < total) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (p.dbg_times && t == 0) t_first = memtime_now();
        if (t + NST - 1 < total) st.issue(p, smem + wr * STAGE, wave);
        const char* sA = smem + rd * STAGE + wm * WTM * ROWB;
        const char* sB = smem + rd * STAGE + A_BYTES + wn * WTN * ROWB;
        opx8 a0[TM], b0[TN], a1[TM], b1[TN];
#define LOAD_FRAGS(A_, B_, KK_)                                                                           \
    do {                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) A_[i] = *(const opx8*)(sA + i * 32 * ROWB + frag_off[KK_]); \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) B_[j] = *(const opx8*)(sB + j * 32 * ROWB + frag_off[KK_]); \
    } while (0)
#define MFMA_RANGE(A_, B_, LO_, HI_)                                                                      \
    do {                                                                                                  \
        _Pragma("unroll") for (int ij = LO_; ij < HI_; ++ij)                                              \
            acc[ij / TN][ij % TN] = MDPT_MFMA_32x32x16(A_[ij / TN], B_[ij % TN], acc[ij / TN][ij % TN], 0, 0, 0); \
    } while (0)
#define PIN() __builtin_amdgcn_sched_barrier(0)
        // Fragment reads are double-buffered in registers and the issue order is pinned (hipcc would otherwise sink
        // every ds_read next to its MFMA: read; wait; mfma). hipcc's waits are always lgkmcnt(0), so the order is
        // chosen such that each wait sits a full MFMA block after the newest outstanding read:
        //   L0 L1 | M0 | L2 | M1 | M2[first] | L3 | M2[rest] | M3
        LOAD_FRAGS(a0, b0, 0);
        LOAD_FRAGS(a1, b1, 1);
        PIN();
        MFMA_RANGE(a0, b0, 0, TM * TN);
        PIN();
        if (KSTEPS == 4) {
            LOAD_FRAGS(a0, b0, 2);
            PIN();
        }
        MFMA_RANGE(a1, b1, 0, TM * TN);
        PIN();
        if (KSTEPS == 4) {
            MFMA_RANGE(a0, b0, 0, 1);
            PIN();
            LOAD_FRAGS(a1, b1, 3);
            PIN();
            MFMA_RANGE(a0, b0, 1, TM * TN);
            PIN();
            MFMA_RANGE(a1, b1, 0, TM * TN);
            PIN();
        }
#undef LOAD_FRAGS
#undef MFMA_RANGE
        rd = rd + 1 == NST ? 0 : rd + 1;
        wr = wr + 1 == NST ? 0 : wr + 1;
    }
    if (p.dbg_times) t_loop = memtime_now();
    __syncthreads();  // every wave is done reading the ring: reuse it as epilogue staging
    if constexpr (KSPLIT) {
        if (p.ksplit > 1 && p.ks_ctr) {
            // In-kernel reduction of the K ranges (GemmParams::ks_ctr): raw accumulators -> partial plane of (range, tile), device-scope
            // release, one ticket per workgroup; the last one to arrive (any of them) adds the planes in the order z = 0, 1, ... - its own
            // included, read back like the others - so the sum does not depend on who that is. No workgroup waits for another.
            static_assert(TM == 1 && TN == 1, "one accumulator block per wave");
            const int ks = p.ksplit;
            float* const plane0 = p.ks_part + (size_t)blockIdx.x * (BM * BN);
            const size_t zstride = (size_t)gridDim.x * (BM * BN);
            float* mine = plane0 + (size_t)kz * zstride;
            // Partial planes and tickets are DEVICE-SCOPE atomics (sc1: coherent across the per-XCD L2s by themselves). A device-scope
            // release / acquire FENCE would write back / invalidate the whole L2 of the XCD (measured: the split then costs more than it
            // saves); here the stores only have to be complete (vmcnt(0), workgroup-scope fence) before the ticket is taken.
#pragma unroll
            for (int r = 0; r < 16; ++r) __hip_atomic_store(mine + r * 256 + tid, acc[0][0][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            unsigned* const flag = (unsigned*)smem;
            if (tid == 0) {
                const unsigned ticket = __hip_atomic_fetch_add(p.ks_ctr + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ticket == (unsigned)(ks - 1)) __hip_atomic_store(p.ks_ctr + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
                *flag = ticket;
            }
            __syncthreads();
            const unsigned ticket = *flag;
            __syncthreads();  // (the strip epilogue reuses smem)
            if (ticket != (unsigned)(ks - 1)) return;
            // ranges in the order z = 0, 1, ...; four planes (64 loads per lane) in flight at a time
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] = __hip_atomic_load(plane0 + r * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int z0 = 1; z0 < ks; z0 += 4) {
                float t[4][16];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int z = z0 + u < ks ? z0 + u : 0;  // (past the end: re-read plane 0, not added)
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[u][r] = __hip_atomic_load(plane0 + (size_t)z * zstride + r * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (z0 + u < ks) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[0][0][r] += t[u][r];
                    }
            }
            run_epilogue<WTN, TM, TN, EKIND>(p, acc, smem, wave, lane, m0 + wm * WTM, n0 + wn * WTN);
            return;
        }
    }
    if (KSPLIT && kz > 0) run_epilogue_partial<WTN, TM, TN>(p, acc, smem, wave, lane, m0 + wm * WTM, n0 + wn * WTN, p.ks_part + (size_t)(kz - 1) * p.M * p.ldc);
    else run_epilogue<WTN, TM, TN, EKIND>(p, acc, smem, wave, lane, m0 + wm * WTM, n0 + wn * WTN);
    if (p.dbg_times && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = p.dbg_times + (size_t)blockIdx.x * 6;
        d[0] = t_start; d[1] = t_first; d[2] = t_loop; d[3] = memtime_now();
        d[4] = __builtin_amdgcn_s_getreg(20 << 0 | 0 << 6 | 3 << 11);  // HW_REG_XCC_ID bits [3:0]
        d[5] = __builtin_amdgcn_s_getreg(4 << 0 | 0 << 6 | 31 << 11);  // HW_REG_HW_ID
    }
}

// ------------------------------------------------------------------------------------------------------------
// Variant B ("8-phase"), 256x256 tile, 8 waves, 64-deep K tiles double-buffered in LDS (128 KiB), 16x16x32 MFMAs.
//
// Two K tiles per loop iteration, four phases per K tile; a phase is
//     { ds_read this phase's register sub-tile | issue the LDS-DMA of ONE half-tile | barrier | 16 MFMAs | barrier }.
// The two wave groups (waves 0-3 / 4-7: one wave of each group per SIMD) run the same program ONE BARRIER apart, so while
// one group issues its MFMA block the other one reads LDS and issues DMA - the matrix pipe sees back-to-back MFMA blocks.
// A wave owns four 64x32 quadrants of the tile, (qm, qn) = rows qm*128 + grp*64, cols qn*128 + wc*32: every 128-row A
// half-tile and every 128-column B half-tile of a K tile is therefore read in exactly one phase,
//     P1: B0, A0 -> MFMA(0,0)    P2: B1 -> MFMA(0,1)    P3: A1 -> MFMA(1,1)    P4: (none) -> MFMA(1,0)
// and can be re-staged early. DMA schedule (e/o = even/odd LDS buffer, t = this iteration's first K tile):
//     P1: -          P2: B0e(t+2)  P3: A0e(t+2)  P4: B1e(t+2), A1e(t+2) + vmcnt(8)
//     P5: -          P6: B0o(t+3)  P7: A0o(t+3)  P8: B1o(t+3), A1o(t+3) + vmcnt(8)
// (round 3: A1 moved out of the 12-read phases P1 / P5 into P4 / P8, which read nothing - QKV 243.9 -> 240.3 us, fc1 372.0 -> 368.0 us on
// one box, profiles/r03_gemm8_schedule_ab.txt.) vmcnt(8) (the 4 half-tiles x 2 DMA instructions issued since the last wait stay in
// flight) at P4 retires the odd buffer, which is read in P5-P7, and at P8 the even buffer, read in P1-P3 of the next iteration: the
// wait sits one phase (>= one workgroup barrier that both groups have passed) before the first read. WAR: a half-tile is re-staged two
// phases after the phase that read it; B0 and A1 are re-staged ONE phase later, which is safe because the reading phase retires those
// reads BEFORE its first barrier (P1 / P5: lgkmcnt(8) for the 4 B reads, issued first, order pinned; P3 / P7: lgkmcnt(0)).
// ------------------------------------------------------------------------------------------------------------
// Direct (register -> global) epilogues of the 8-phase kernel for tiles computed with SWAPPED MFMA operands
// (acc = mfma(B frag, A frag)): a lane then owns, for output row m = 16-row block + (lane & 15), four CONSECUTIVE columns
// n = 16-col block + 4*(lane >> 4) + r - fp32 values go out as 16-byte vectors straight from the accumulators, and a bf16
// result gets its 8-column / 16-byte vectors by one v_permlane16_swap per register pair between lanes l and l^16
// (even 16-lane rows collect block j = 0, odd rows block j = 1). No LDS round trip. Everything is unrolled over the 128
// accumulator registers, so only the three hot epilogues of the encoder get this form (small code, no per-row feature
// branches); every other combination runs the plain operand order + the LDS-strip epilogue:
//   DM_BF16  : out_hi(/lo) = act(acc + bias)                    (fc1 + GELU; plain bf16 outputs)
//   DM_RESID : out_f32 = resid + gamma * (acc + bias), fp32      (attention proj, fc2: in-place residual update)
//   DM_QK    : Q (pre-scaled) / K head-major bf16(/lo)            (QKV tiles without V columns)
//   DM_F32   : out_f32 = acc + bias, fp32                          (1x1 projections, SwinV2 QKV / proj / fc2, SwiGLU inner linear)
//   DM_RINIT : the DM_F32 epilogue behind a main loop whose accumulators were INITIALISED with the residual tile (attention proj,
//              fc2 with the layer scale folded into the packed weights and bias): out = (resid + A W'^T) + bias'. The 256 KB residual
//              read of a tile streams in under the first K tiles instead of sitting exposed in the epilogue.
//   DM_VT    : V transposed, token-contiguous (epilogue_direct_vt: PLAIN operand order, a lane owns 4 consecutive tokens)
//   DM_SWQK  : SwinV2 cosine-attention Q / K (heads of 32): L2-normalised, Q times the head's logit scale, scattered to window order
//              (QKV tiles without V columns when GemmParams::swin_tokmap is set; the V tiles of that GEMM use DM_F32)
//   DM_SWVT  : SwinV2 V columns written as the transposed window operand (4-token runs, plain operand order like DM_VT)
// Per-image bias tables in the direct epilogues exist in the fp16 build only (the bf16 build's kernels stay exactly what round 3 tuned);
// launches that carry one in the bf16 build run the strip-epilogue kernels.
constexpr bool HAVE_IMGB = MDPT_OP_IS_F16 != 0;
enum { DM_NONE = 0, DM_BF16 = 1, DM_RESID = 2, DM_QK = 3, DM_VT = 4, DM_F32 = 5, DM_RINIT = 6, DM_SWQK = 7, DM_SWVT = 8 };

// Everything that selects code is a template parameter (MODE, X3 = hi+lo output planes, ACT) and every memory access is a raw
// buffer op whose out-of-range lanes (tail rows: offset beyond num_records; tail columns: offset forced to ~0) are dropped by
// the hardware bounds check. No divergent branch means hipcc can count vmcnt: with `if (row < M) store` every store sat in its
// own basic block behind an `s_waitcnt vmcnt(0)`, i.e. each store waited for the previous one's write acknowledgement
// (measured: 520 cycles per store, 33-42k cycles for the in-place residual form whose loads waited the same way).
// Row addresses advance by uniform strides; DM_RESID reads back exactly the 16-byte groups it writes, group g+1's loads are
// issued ahead of group g's stores.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(unsigned)(bytes < 0xFFFFFFF0ull ? bytes : 0xFFFFFFF0ull), 0x00020000);
}

// IMGB: per-image bias table (GemmParams::bias_img_stride, rows of bias_img_rows >= 256 per image - the token-mean compensation of the
// weight rounding, mdpt_stages.cpp wrc_bias): a 256-row tile lies in at most two images, rows from tile-local index `bnd` on take the
// next image's bias vector. One select per value, then the SAME single add as every other form of the epilogue ((acc + bias) ...).
template <int MODE, bool X3, int ACT, bool IMGB = false>
__device__ __forceinline__ void epilogue_direct(const GemmParams& p, f32x4 (&acc)[2][2][4][2], int m0, int n0, int grp, int wc, int lane) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const int l15 = lane & 15, lh = lane >> 4;
    const int rows_here = p.M - m0 < 256 ? p.M - m0 : 256;  // valid rows of this tile (tail tile: fewer)
    int bnd = 1 << 30;  // tile-local row where the next image starts (IMGB)
    const float* bias0 = p.bias;
    const float* bias1 = p.bias;
    if (IMGB) {
        const int img0 = m0 / p.bias_img_rows;
        bnd = (img0 + 1) * p.bias_img_rows - m0;
        bias0 = p.bias + (size_t)img0 * p.bias_img_stride;
        bias1 = bnd < rows_here ? bias0 + p.bias_img_stride : bias0;
    }
    auto pick = [&](const f32x4& a, const f32x4& b, bool next) {  // (IMGB only) per-lane select, no arithmetic
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = next ? b[e] : a[e];
        return r;
    };

    if (MODE == DM_F32) {
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(p.out_f32 + (size_t)m0 * p.ldc, (size_t)rows_here * p.ldc * 4);
        const unsigned row_b = (unsigned)p.ldc * 4u;
#pragma unroll
        for (int qn = 0; qn < 2; ++qn) {
            f32x4 bias[2], biasn[2];
            unsigned col_off[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nc = n0 + qn * 128 + wc * 32 + j * 16 + 4 * lh;
                bias[j] = p.bias ? *(const f32x4*)(bias0 + (nc < p.N ? nc : p.N - 4)) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                if (IMGB) biasn[j] = *(const f32x4*)(bias1 + (nc < p.N ? nc : p.N - 4));
                col_off[j] = nc < p.N ? (unsigned)nc * 4u : OOB;
            }
#pragma unroll
            for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned row = (unsigned)(qm * 128 + grp * 64 + i * 16 + l15) * row_b;
                    const bool next = IMGB && qm * 128 + grp * 64 + i * 16 + l15 >= bnd;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 v = acc[qm][qn][i][j] + (IMGB ? pick(bias[j], biasn[j], next) : bias[j]);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, col_off[j] == OOB ? OOB : row + col_off[j], 0, 0);
                    }
                }
        }
        return;
    }
    if (MODE == DM_RESID) {
        // in-place residual update; descriptor over this tile's rows (base = row m0), byte offsets inside it
        const __amdgpu_buffer_rsrc_t rs = tile_rsrc(p.out_f32 + (size_t)m0 * p.ldc, (size_t)rows_here * p.ldc * 4);
        const unsigned row_b = (unsigned)p.ldc * 4u;
        f32x4 bias[2][2], gam[2][2];
        unsigned col_off[2][2];
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nc = n0 + qn * 128 + wc * 32 + j * 16 + 4 * lh;
                const int ncl = nc < p.N ? nc : p.N - 4;
                bias[qn][j] = *(const f32x4*)(p.bias + ncl);
                gam[qn][j] = *(const f32x4*)(p.gamma + ncl);
                col_off[qn][j] = nc < p.N ? (unsigned)nc * 4u : OOB;
            }
        auto row_off = [&](int g, int i) { return (unsigned)((g >> 1) * 128 + grp * 64 + i * 16 + l15) * row_b; };  // group g = 2 qm + qn
        // vmcnt retires in order, so a wait for loads that were issued after a store also waits for that store's write
        // acknowledgement (~2 us). Order: loads of groups 0,1 -> results 0,1 computed in registers -> loads of groups 2,3
        // (into the registers the finished accumulators free) -> stores 0,1 -> wait for loads 2,3 only -> stores 2,3.
        u32x4 old[4][4][2];
        f32x4 res[2][4][2];
        auto load_group = [&](int g) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned c = col_off[g & 1][j];
                    old[g][i][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, c == OOB ? OOB : row_off(g, i) + c, 0, 0);
                }
        };
        auto value = [&](int g, int i, int j) {
            return (acc[g >> 1][g & 1][i][j] + bias[g & 1][j]) * gam[g & 1][j] + __builtin_bit_cast(f32x4, old[g][i][j]);
        };
        auto store = [&](int g, int i, int j, f32x4 v) {
            const unsigned c = col_off[g & 1][j];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, c == OOB ? OOB : row_off(g, i) + c, 0, 0);
        };
        // test hook (dbg_times): wave 0 stamps [epilogue start, loads 0,1 issued, loads 0,1 landed, loads 2,3 + stores 0,1 issued,
        // loads 2,3 landed, all stores issued] behind the 6 per-workgroup slots of every workgroup
        unsigned long long* stamp = p.dbg_times && threadIdx.x == 0 ? p.dbg_times + (size_t)gridDim.x * 6 + (size_t)blockIdx.x * 16 : nullptr;
        if (stamp) stamp[0] = memtime_now();
        load_group(0);
        load_group(1);
        if (p.dbg_times) {
            if (stamp) stamp[1] = memtime_now();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (stamp) stamp[2] = memtime_now();
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) res[g][i][j] = value(g, i, j);
        load_group(2);
        load_group(3);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) store(g, i, j, res[g][i][j]);
        if (p.dbg_times) {
            if (stamp) stamp[3] = memtime_now();
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            if (stamp) stamp[4] = memtime_now();
        }
#pragma unroll
        for (int g = 2; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) store(g, i, j, value(g, i, j));
        if (stamp) stamp[5] = memtime_now();
        return;
    }

    // ---- bf16 outputs
    __amdgpu_buffer_rsrc_t rs_hi, rs_lo;
    if (MODE == DM_BF16) {
        rs_hi = tile_rsrc(p.out_hi + (size_t)m0 * p.ldc, (size_t)rows_here * p.ldc * 2);
        rs_lo = tile_rsrc(X3 ? p.out_lo + (size_t)m0 * p.ldc : p.out_hi, X3 ? (size_t)rows_here * p.ldc * 2 : 0);
    }
    // per-column constants of this lane's 4-column groups, both column halves loaded before the first store (a wait placed
    // after a store would also wait for that store's acknowledgement)
    f32x4 bias_q[2][2], biasn_q[2][2], gam_q[2][2];
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ncol = n0 + qn * 128 + wc * 32 + j * 16 + 4 * lh;
            const int nc = ncol < p.N ? ncol : p.N - 4;  // clamped: out-of-range columns are never stored
            bias_q[qn][j] = p.bias ? *(const f32x4*)(bias0 + nc) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (IMGB) biasn_q[qn][j] = *(const f32x4*)(bias1 + nc);
            if (MODE == DM_QK && ncol < p.F) gam_q[qn][j] = f32x4{p.qscale, p.qscale, p.qscale, p.qscale};
            else if (MODE == DM_QK) gam_q[qn][j] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
        }
#pragma unroll
    for (int qn = 0; qn < 2; ++qn) {
        const int nq = n0 + qn * 128 + wc * 32;
        const int n8 = nq + (lh & 1) * 16 + (lh >> 1) * 8;  // first of the 8 columns this lane stores as bf16
        const f32x4(&bias)[2] = bias_q[qn];
        const f32x4(&biasn)[2] = biasn_q[qn];
        const f32x4(&gam)[2] = gam_q[qn];
        // QKV: this wave's 32 columns lie in one of the Q / K planes (F is a multiple of 64): wave-uniform descriptor
        int qk_h = 0, qk_d = 0;
        if (MODE == DM_QK) {
            const int which = __builtin_amdgcn_readfirstlane(nq >= p.F);
            const int fcol = n8 - which * p.F;
            qk_h = fcol >> 6; qk_d = fcol & 63;
            const size_t plane = (size_t)p.M * p.F * 2;  // [B, heads, npad, 64] bf16 (M = B * npad); < 4 GiB checked by the caller
            rs_hi = tile_rsrc(which ? p.k_hi : p.q_hi, plane);
            rs_lo = tile_rsrc(X3 ? (which ? p.k_lo : p.q_lo) : p.q_hi, X3 ? plane : 0);
        }
        const bool nok = n8 < p.N;
#pragma unroll
        for (int qm = 0; qm < 2; ++qm) {
            const int rfirst = qm * 128 + grp * 64 + l15;  // tile-local row of block i is rfirst + 16 i
            int qb = 0, qt = 0;
            if (MODE == DM_QK) {
                qb = (m0 + rfirst) / p.npad;
                qt = m0 + rfirst - qb * p.npad;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v[2];
                const bool next = IMGB && rfirst + 16 * i >= bnd;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    v[j] = acc[qm][qn][i][j] + (IMGB ? pick(bias[j], biasn[j], next) : bias[j]);
                    if (MODE == DM_QK) {
                        v[j] *= gam[j];
                    } else if (ACT == MDPT_ACT_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {
                            const gelu_f32x2 g2 = gelu_erf2(gelu_f32x2{v[j][e], v[j][e + 1]});
                            v[j][e] = g2[0];
                            v[j][e + 1] = g2[1];
                        }
                    } else if (ACT == MDPT_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[j][e] = fmaxf(v[j][e], 0.0f);
                    }
                }
                // pack pairs, exchange halves with lane ^ 16, one 16-byte store (two in x3 mode)
                unsigned hw_[2][2], lw_[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2) {
                        const f32x2 pp = {v[j][2 * w2], v[j][2 * w2 + 1]};
                        const opx2 hh = to_op2(pp);
                        hw_[j][w2] = __builtin_bit_cast(unsigned, hh);
                        if (X3) {
                            const f32x2 rr = pp - __builtin_convertvector(hh, f32x2);
                            lw_[j][w2] = __builtin_bit_cast(unsigned, to_op2(rr));
                        }
                    }
                unsigned ph[4], pl[4];
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    auto r = __builtin_amdgcn_permlane16_swap(hw_[0][w2], hw_[1][w2], false, false);
                    ph[w2] = r[0];
                    ph[w2 + 2] = r[1];
                    if (X3) {
                        auto rl = __builtin_amdgcn_permlane16_swap(lw_[0][w2], lw_[1][w2], false, false);
                        pl[w2] = rl[0];
                        pl[w2 + 2] = rl[1];
                    }
                }
                unsigned off;
                if (MODE == DM_BF16) {
                    off = ((unsigned)(rfirst + 16 * i) * (unsigned)p.ldc + (unsigned)n8) * 2u;  // rows >= M lie beyond num_records
                    if (!nok) off = OOB;
                } else {
                    off = (unsigned)((((size_t)(qb * p.heads + qk_h) * p.npad + qt) * 64 + qk_d) * 2);
                    if (!nok || m0 + rfirst + 16 * i >= p.M) off = OOB;
                }
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{ph[0], ph[1], ph[2], ph[3]}, rs_hi, off, 0, 0);
                if (X3) __builtin_amdgcn_raw_buffer_store_b128(u32x4{pl[0], pl[1], pl[2], pl[3]}, rs_lo, off, 0, 0);
                if (MODE == DM_QK) {  // next row block: token + 16, rolling over into the next image
                    qt += 16;
                    while (qt >= p.npad) { qt -= p.npad; ++qb; }
                }
            }
        }
    }
}

// SwinV2 window-attention operands straight out of a Q / K tile of the QKV GEMM (windowed_attention.py:100-123: F.normalize(q), F.normalize(k),
// q * exp(clamped logit scale)). Heads are 32 wide = the 32 columns one wave owns in a quadrant: a lane holds two 4-column groups of a row (groups
// lh and 4 + lh of the head), the four lanes lane, lane^16, lane^32, lane^48 hold the row's whole head. |.|^2 is summed in the order
// swin_qk_prep_kernel (swin.hip) uses on the fp32 QKV rows - (a^2 + b^2) + (c^2 + d^2) per group, group c with group c+4, then the neighbour
// group pair, then the other half - with unfused multiplies, so the fused and the unfused form of a block give the same bits whichever tile
// rule picks which. Rows scatter through the token map: image token t -> swin_tokmap[t] = w*heads*npad + i, + img*swin_img_rows + h*npad.
template <bool X3, bool IMGB = false>
__device__ __forceinline__ void epilogue_swin_qk(const GemmParams& p, f32x4 (&acc)[2][2][4][2], int m0, int n0, int grp, int wc, int lane) {
#pragma clang fp contract(off)
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const int l15 = lane & 15, lh = lane >> 4;
    const size_t plane = (size_t)(p.M / p.swin_N) * p.swin_img_rows * 64;  // bytes of a Q / K plane (< 4 GiB: checked by the caller)
    // IMGB: per-image bias table (token-mean compensation, see epilogue_direct): >= 256 token rows per image, two bias vectors per tile
    int bnd = 1 << 30;
    const float* bias0 = p.bias;
    const float* bias1 = p.bias;
    if (IMGB) {
        const int rows_here = p.M - m0 < 256 ? p.M - m0 : 256;
        const int img0 = m0 / p.bias_img_rows;
        bnd = (img0 + 1) * p.bias_img_rows - m0;
        bias0 = p.bias + (size_t)img0 * p.bias_img_stride;
        bias1 = bnd < rows_here ? bias0 + p.bias_img_stride : bias0;
    }
    // everything that is loaded comes before the first store (a wait after a store also waits for the store)
    f32x4 bias_q[2][2], biasn_q[2][2];
    float scale_q[2];
#pragma unroll
    for (int qn = 0; qn < 2; ++qn) {
        const int nq = n0 + qn * 128 + wc * 32;
        const int which = __builtin_amdgcn_readfirstlane(nq >= p.F);
        const float ls = p.swin_logit_scale[(nq - which * p.F) >> 5];
        scale_q[qn] = which ? 1.0f : ls;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            bias_q[qn][j] = p.bias ? *(const f32x4*)(bias0 + nq + j * 16 + 4 * lh) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (IMGB) biasn_q[qn][j] = *(const f32x4*)(bias1 + nq + j * 16 + 4 * lh);
        }
    }
    int dst[2][4];
#pragma unroll
    for (int qm = 0; qm < 2; ++qm) {
        const int r0 = m0 + qm * 128 + grp * 64 + l15;
        int img = r0 / p.swin_N, t = r0 - img * p.swin_N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[qm][i] = r0 + 16 * i < p.M ? img * p.swin_img_rows + p.swin_tokmap[t] : -1;
            t += 16;
            while (t >= p.swin_N) { t -= p.swin_N; ++img; }
        }
    }
    const int d8 = (lh & 1) * 16 + (lh >> 1) * 8;  // first of the 8 head columns this lane stores (after the lane^16 exchange)
#pragma unroll
    for (int qn = 0; qn < 2; ++qn) {
        // (Q | K plane and head recomputed here, wave-uniform scalars: kept in arrays across the loops above they became a pointer table in scratch)
        const int nq = n0 + qn * 128 + wc * 32;
        const int which = __builtin_amdgcn_readfirstlane(nq >= p.F);
        const __amdgpu_buffer_rsrc_t rs_hi = tile_rsrc(which ? p.k_hi : p.q_hi, plane);
        const __amdgpu_buffer_rsrc_t rs_lo = tile_rsrc(X3 ? (which ? p.k_lo : p.q_lo) : p.q_hi, X3 ? plane : 0);
        const int hrow = ((nq - which * p.F) >> 5) * p.npad;
#pragma unroll
        for (int qm = 0; qm < 2; ++qm)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 v[2];
                float sg[2];
                const bool next = IMGB && qm * 128 + grp * 64 + 16 * i + l15 >= bnd;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 bj = bias_q[qn][j];
                    if (IMGB) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) bj[e] = next ? biasn_q[qn][j][e] : bj[e];
                    }
                    v[j] = acc[qm][qn][i][j] + bj;
                    sg[j] = (v[j][0] * v[j][0] + v[j][1] * v[j][1]) + (v[j][2] * v[j][2] + v[j][3] * v[j][3]);
                }
                float ss = sg[0] + sg[1];
                // cross-lane sums with the swap instructions: swap(x, x) leaves [R0 R0 R2 R2] / [R1 R1 R3 R3] (rows of 16 lanes) resp.
                // [lo lo] / [hi hi] in the two registers. The results go through scalars: __builtin_bit_cast applied to an element of the
                // returned vector reads element 0 both times (DESIGN.md bug 4)
                {
                    const unsigned u = __builtin_bit_cast(unsigned, ss);
                    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
                    const unsigned r0 = r[0], r1 = r[1];
                    ss = __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
                }
                {
                    const unsigned u = __builtin_bit_cast(unsigned, ss);
                    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                    const unsigned r0 = r[0], r1 = r[1];
                    ss = __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
                }
                float scale = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
                scale *= scale_q[qn];
                unsigned hw_[2][2], lw_[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2) {
                        const f32x2 pp = {v[j][2 * w2] * scale, v[j][2 * w2 + 1] * scale};
                        const opx2 hh = to_op2(pp);
                        hw_[j][w2] = __builtin_bit_cast(unsigned, hh);
                        if (X3) {
                            const f32x2 rr = pp - __builtin_convertvector(hh, f32x2);
                            lw_[j][w2] = __builtin_bit_cast(unsigned, to_op2(rr));
                        }
                    }
                unsigned ph[4], pl[4];
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    auto r = __builtin_amdgcn_permlane16_swap(hw_[0][w2], hw_[1][w2], false, false);
                    ph[w2] = r[0];
                    ph[w2 + 2] = r[1];
                    if (X3) {
                        auto rl = __builtin_amdgcn_permlane16_swap(lw_[0][w2], lw_[1][w2], false, false);
                        pl[w2] = rl[0];
                        pl[w2 + 2] = rl[1];
                    }
                }
                const unsigned off = dst[qm][i] < 0 ? OOB : (unsigned)(((size_t)(dst[qm][i] + hrow) * 32 + d8) * 2);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{ph[0], ph[1], ph[2], ph[3]}, rs_hi, off, 0, 0);
                if (X3) __builtin_amdgcn_raw_buffer_store_b128(u32x4{pl[0], pl[1], pl[2], pl[3]}, rs_lo, off, 0, 0);
            }
    }
}

// SwinV2 V columns as the window attention's transposed operand Vt[(img*nw + w)*heads + h][d][npadv] (what swin_v_prep_kernel builds from
// fp32 rows: bf16(acc + bias), lo = bf16(v - hi) in bf16x3 mode - the same two conversions). Plain operand order: a lane owns 4 consecutive
// rows = image tokens t .. t+3 (t % 4 == 0) of one column; the caller guarantees grid width, window width and shift are multiples of 4,
// so the four tokens are consecutive positions of one window: one 8-byte store per lane, row block and column. Pad positions [wa, npadv)
// are zeroed by the caller once per stage.
template <bool X3, bool IMGB = false>
__device__ __forceinline__ void epilogue_swin_vt(const GemmParams& p, f32x4 (&acc)[2][2][4][2], int m0, int n0, int grp, int wc, int lane) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const int l15 = lane & 15, lh = lane >> 4;
    const size_t plane = (size_t)(p.M / p.swin_N) * p.swin_img_velems * 2;  // bytes; < 4 GiB checked by the caller
    const __amdgpu_buffer_rsrc_t rs_hi = tile_rsrc(p.vt_hi, plane);
    const __amdgpu_buffer_rsrc_t rs_lo = tile_rsrc(X3 ? p.vt_lo : p.vt_hi, X3 ? plane : 0);
    int bnd = 1 << 30;  // IMGB: per-image bias table, see epilogue_direct_vt (a lane's 4 rows never straddle two images: token counts are multiples of 4)
    const float* bias0 = p.bias;
    const float* bias1 = p.bias;
    if (IMGB) {
        const int rows_here = p.M - m0 < 256 ? p.M - m0 : 256;
        const int img0 = m0 / p.bias_img_rows;
        bnd = (img0 + 1) * p.bias_img_rows - m0;
        bias0 = p.bias + (size_t)img0 * p.bias_img_stride;
        bias1 = bnd < rows_here ? bias0 + p.bias_img_stride : bias0;
    }
    float bias_q[2][2], biasn_q[2][2];
    int col_q[2][2];
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + qn * 128 + wc * 32 + j * 16 + l15;
            bias_q[qn][j] = p.bias ? bias0[n < p.N ? n : p.N - 1] : 0.0f;
            if (IMGB) biasn_q[qn][j] = bias1[n < p.N ? n : p.N - 1];
            col_q[qn][j] = n < p.N ? (n - 2 * p.F) * p.npadv : -1;  // (h*32 + d) * npadv
        }
    int dst[2][4];
#pragma unroll
    for (int qm = 0; qm < 2; ++qm) {
        const int r0 = m0 + qm * 128 + grp * 64 + 4 * lh;
        int img = r0 / p.swin_N, t = r0 - img * p.swin_N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dst[qm][i] = r0 + 16 * i < p.M ? img * p.swin_img_velems + p.swin_vtokmap[t] : -1;
            t += 16;
            while (t >= p.swin_N) { t -= p.swin_N; ++img; }
        }
    }
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int qm = 0; qm < 2; ++qm)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bool next = IMGB && qm * 128 + grp * 64 + 16 * i + 4 * lh >= bnd;
                    const f32x4 v = acc[qm][qn][i][j] + (next ? biasn_q[qn][j] : bias_q[qn][j]);
                    unsigned hw_[2], lw_[2];
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2) {
                        const f32x2 pp = {v[2 * w2], v[2 * w2 + 1]};
                        const opx2 hb = to_op2(pp);
                        hw_[w2] = __builtin_bit_cast(unsigned, hb);
                        if (X3) {
                            const f32x2 rr = pp - __builtin_convertvector(hb, f32x2);
                            lw_[w2] = __builtin_bit_cast(unsigned, to_op2(rr));
                        }
                    }
                    const unsigned off = dst[qm][i] < 0 || col_q[qn][j] < 0 ? OOB : (unsigned)(dst[qm][i] + col_q[qn][j]) * 2u;
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{hw_[0], hw_[1]}, rs_hi, off, 0, 0);
                    if (X3) __builtin_amdgcn_raw_buffer_store_b64(u32x2{lw_[0], lw_[1]}, rs_lo, off, 0, 0);
                }
}

// V columns of the QKV GEMM, written transposed: Vt[(b, h, d)][token]. With the PLAIN MFMA operand order a lane owns 4 consecutive
// rows (tokens 4*(lane>>4) .. +3 of a 16-row block) of one column, i.e. 8 bytes of a Vt row; v_permlane16_swap between the two
// 16-column blocks of the quadrant gives lanes (lane>>4) = 0,1 the 8 tokens 0-7 of a column of block 0 / block 1 and lanes 2,3
// the tokens 8-15: one 16-byte store per lane and row block, no LDS. Same arithmetic as the strip path (bias add, hi/lo split).
template <bool X3, bool IMGB = false>
__device__ __forceinline__ void epilogue_direct_vt(const GemmParams& p, f32x4 (&acc)[2][2][4][2], int m0, int n0, int grp, int wc, int lane) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    constexpr unsigned OOB = 0xFFFFFFF0u;
    const int l15 = lane & 15, lh = lane >> 4;
    const size_t plane = (size_t)(p.M / p.npad) * p.F * p.npadv * 2;  // [B, heads, 64, npadv] bf16; < 4 GiB checked by the caller
    const __amdgpu_buffer_rsrc_t rs_hi = tile_rsrc(p.vt_hi, plane);
    const __amdgpu_buffer_rsrc_t rs_lo = tile_rsrc(X3 ? p.vt_lo : p.vt_hi, X3 ? plane : 0);
    // IMGB: per-image bias table, see epilogue_direct. A lane's 4 rows (tokens) start at a multiple of 4 and images at multiples of 8
    // rows: the four values of a register quad always belong to one image
    int bnd = 1 << 30;
    const float* bias0 = p.bias;
    const float* bias1 = p.bias;
    if (IMGB) {
        const int rows_here = p.M - m0 < 256 ? p.M - m0 : 256;
        const int img0 = m0 / p.bias_img_rows;
        bnd = (img0 + 1) * p.bias_img_rows - m0;
        bias0 = p.bias + (size_t)img0 * p.bias_img_stride;
        bias1 = bnd < rows_here ? bias0 + p.bias_img_stride : bias0;
    }
    float bias_q[2][2], biasn_q[2][2];
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + qn * 128 + wc * 32 + j * 16 + l15;
            bias_q[qn][j] = bias0[n < p.N ? n : p.N - 1];
            if (IMGB) biasn_q[qn][j] = bias1[n < p.N ? n : p.N - 1];
        }
#pragma unroll
    for (int qn = 0; qn < 2; ++qn) {
        const int ncol = n0 + qn * 128 + wc * 32 + (lh & 1) * 16 + l15;  // the column this lane stores after the swap
        const int fcol = ncol - 2 * p.F, hh = fcol >> 6, dd = fcol & 63;
        const bool nok = ncol < p.N;
#pragma unroll
        for (int qm = 0; qm < 2; ++qm) {
            const int mfirst = m0 + qm * 128 + grp * 64 + (lh >> 1) * 8;  // first of this lane's 8 tokens in row block 0
            int qb = mfirst / p.npad, qt = mfirst - qb * p.npad;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned hw_[2][2], lw_[2][2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // plain operand order: this lane's rows of block i are qm*128 + grp*64 + 16 i + 4 lh .. + 3
                    const bool next = IMGB && qm * 128 + grp * 64 + 16 * i + 4 * lh >= bnd;
                    const f32x4 v = acc[qm][qn][i][j] + (next ? biasn_q[qn][j] : bias_q[qn][j]);
#pragma unroll
                    for (int w2 = 0; w2 < 2; ++w2) {
                        const f32x2 pp = {v[2 * w2], v[2 * w2 + 1]};
                        const opx2 hb = to_op2(pp);
                        hw_[j][w2] = __builtin_bit_cast(unsigned, hb);
                        if (X3) {
                            const f32x2 rr = pp - __builtin_convertvector(hb, f32x2);
                            lw_[j][w2] = __builtin_bit_cast(unsigned, to_op2(rr));
                        }
                    }
                }
                unsigned ph[4], pl[4];
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    auto r = __builtin_amdgcn_permlane16_swap(hw_[0][w2], hw_[1][w2], false, false);
                    ph[w2] = r[0];
                    ph[w2 + 2] = r[1];
                    if (X3) {
                        auto rl = __builtin_amdgcn_permlane16_swap(lw_[0][w2], lw_[1][w2], false, false);
                        pl[w2] = rl[0];
                        pl[w2 + 2] = rl[1];
                    }
                }
                unsigned off = (unsigned)((((size_t)(qb * p.heads + hh) * 64 + dd) * p.npadv + qt) * 2);
                if (!nok || mfirst + 16 * i >= p.M) off = OOB;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{ph[0], ph[1], ph[2], ph[3]}, rs_hi, off, 0, 0);
                if (X3) __builtin_amdgcn_raw_buffer_store_b128(u32x4{pl[0], pl[1], pl[2], pl[3]}, rs_lo, off, 0, 0);
                qt += 16;
                while (qt >= p.npad) { qt -= p.npad; ++qb; }
            }
        }
    }
}

template <int AMODE>
struct HalfStager {  // BM = BN = 256, 8 waves, BK = 64: chunk c = wave + 8 i covers rows 8c..8c+7; half h = chunks i in {2h, 2h+1}
    static constexpr int A_BYTES = 256 * 128;
    // Measured on one box, interleaved (profiles/r03_gemm8_staging_ab.txt): the buffered form is NOT faster for the dense GEMMs (QKV 240.7
    // vs 237.8 us, fc1 366.8 vs 364.0, residual GEMMs 231.5 vs 230.5) - their loop is bound by the 64 DMA instructions per K tile, not by
    // the 16 VALU operations of the pointer form - so the pointer form stays the default and -DMDPT_GEMM8_BUF_STAGING builds the other one.
#ifdef MDPT_GEMM8_BUF_STAGING
    static constexpr bool BUFFERED = AMODE != MDPT_A_CONV3;
#else
    static constexpr bool BUFFERED = false;
#endif
    // Dense / token rows (BUFFERED): LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... offen lds) whose base is THIS tile's
    // first row: the per-lane byte offsets are constants of the tile (row, swizzled k-chunk), the position along K and the 64-row step
    // between a wave's DMA instructions are SCALAR offsets - no 64-bit pointer arithmetic in the loop (16 VALU operations per K tile in
    // the pointer form: measured 2500 -> 2200 cycles per K tile on the conv kernel that was written this way first) - and rows past
    // M / N fail the descriptor's bounds check and are staged as zeros (tools/probes/buffer_lds_oob.hip) instead of being clamped.
    // 3x3 taps (im2col-free conv): per-lane source pointers, recomputed per tap (global_load_lds).
    __amdgpu_buffer_rsrc_t rs_a, rs_w;
    unsigned a_voff[4], b_voff;
    int a_soff, b_soff, a_row_step, b_row_step;
    const op_t* a_base_hi; const op_t* a_base_lo; const op_t* w_base_hi; const op_t* w_base_lo;
    size_t a_bytes, w_bytes;
    const op_t* a_ptr[4];
    int a_pix[4], a_y[4], a_x[4], a_ko[4];
    const op_t* b_ptr[4];
    ptrdiff_t a_hi_minus_lo, w_lo_minus_hi;
    const op_t* conv_plane;
    int a_pass, a_k0, a_tap, a_ci, b_pass, b_k0;

    static __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, size_t bytes) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(unsigned)(bytes < 0xFFFFFFF0ull ? bytes : 0xFFFFFFF0ull), 0x00020000);
    }

    __device__ __forceinline__ void init(const GemmParams& p, int m0, int n0, int wave, int lane) {
        const int lrow = lane >> 3, slot = lane & 7;
        const op_t* A0 = p.npass == 3 ? p.A_lo : p.A_hi;
        a_pass = a_k0 = a_tap = a_ci = b_pass = b_k0 = 0;
        if constexpr (BUFFERED) {
            constexpr unsigned OOB = 0xFFFFFFF0u;
            const int rows_here = p.M - m0 < 256 ? p.M - m0 : 256, cols_here = p.N - n0 < 256 ? p.N - n0 : 256;
            // source row of logical row m (token rows skip the cls row of every image)
            auto src_row = [&](int m) -> size_t {
                if (AMODE == MDPT_A_TOKENS) { const int b = m / p.tok_np, t = m - b * p.tok_np; return (size_t)b * p.tok_stride + 1 + t; }
                return (size_t)m;
            };
            const size_t row0 = src_row(m0), row_last = src_row(m0 + rows_here - 1);
            a_base_hi = p.A_hi + row0 * p.lda; a_base_lo = p.npass == 3 ? p.A_lo + row0 * p.lda : a_base_hi;
            a_bytes = (row_last - row0 + 1) * p.lda * 2;
            w_base_hi = p.W_hi + (size_t)n0 * p.ldw; w_base_lo = p.npass == 3 ? p.W_lo + (size_t)n0 * p.ldw : w_base_hi;
            w_bytes = ((size_t)(cols_here - 1) * p.ldw + p.K) * 2;
            rs_a = rsrc(p.npass == 3 ? a_base_lo : a_base_hi, a_bytes);  // pass 0 of bf16x3: A_lo * W_hi
            rs_w = rsrc(w_base_hi, w_bytes);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = (wave + 8 * i) * 8 + lrow;
                const int koff = (slot ^ ((r >> 1) & 7)) * 8;
                a_voff[i] = r < rows_here ? (unsigned)(((src_row(m0 + r) - row0) * p.lda + koff) * 2) : OOB;
            }
            {
                const int r = wave * 8 + lrow;  // rows of DMA instruction i: r + 64 i, same swizzle key
                b_voff = (unsigned)(((size_t)r * p.ldw + ((slot ^ ((r >> 1) & 7)) * 8)) * 2);
            }
            a_soff = b_soff = 0;
            b_row_step = 64 * p.ldw * 2;
            a_row_step = 0;
            a_hi_minus_lo = w_lo_minus_hi = 0;
            conv_plane = nullptr;
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (wave + 8 * i) * 8 + lrow;
            const int koff = (slot ^ ((r >> 1) & 7)) * 8;
            int m = m0 + r;
            m = m < p.M ? m : p.M - 1;
            a_ko[i] = koff;
            a_ptr[i] = nullptr;
            a_pix[i] = a_y[i] = a_x[i] = 0;
            if (AMODE == MDPT_A_DENSE) {
                a_ptr[i] = A0 + (size_t)m * p.lda + koff;
            } else if (AMODE == MDPT_A_TOKENS) {
                const int b = m / p.tok_np, t = m - b * p.tok_np;
                a_ptr[i] = A0 + ((size_t)b * p.tok_stride + 1 + t) * p.lda + koff;
            } else {
                const int hw = p.Ho * p.Wo;
                const int b = m / hw, rem = m - b * hw;
                const int y = rem / p.Wo, x = rem - y * p.Wo;
                a_pix[i] = b * p.Hi * p.Wi;
                a_y[i] = y * p.cstride - 1;
                a_x[i] = x * p.cstride - 1;
            }
            int n = n0 + r;
            n = n < p.N ? n : p.N - 1;
            b_ptr[i] = p.W_hi + (size_t)n * p.ldw + koff;
        }
        a_hi_minus_lo = p.npass == 3 ? p.A_hi - p.A_lo : 0;
        w_lo_minus_hi = p.npass == 3 ? p.W_lo - p.W_hi : 0;
        conv_plane = A0;
    }

    template <int H>
    __device__ __forceinline__ void issue_a(const GemmParams& p, char* buf, int wave) {
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        if constexpr (BUFFERED) {
#pragma unroll
            for (int i = 2 * H; i < 2 * H + 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)(buf + (wave + 8 * i) * 1024), 16, a_voff[i], a_soff, 0, 0);
            if (H == 1) {  // both halves of this K tile are on their way: advance K (and the bf16x3 operand planes at roll-over)
                a_soff += 128;
                if (a_soff == p.K * 2) {
                    a_soff = 0;
                    if (a_pass == 0) rs_a = rsrc(a_base_hi, a_bytes);  // passes 1, 2 read A_hi
                    ++a_pass;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 2 * H; i < 2 * H + 2; ++i) {
            const op_t* src;
            if (AMODE == MDPT_A_CONV3) {
                const int ky = (a_tap * 11) >> 5, kx = a_tap - 3 * ky;
                const int iy = a_y[i] + ky, ix = a_x[i] + kx;
                const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                src = ok ? conv_plane + ((size_t)(a_pix[i] + iy * p.Wi + ix) * p.Cin + a_ci + a_ko[i]) : p.zero_page + a_ko[i];
            } else {
                src = a_ptr[i];
                a_ptr[i] += 64;
            }
            glds16(src, buf + (wave + 8 * i) * 1024);
        }
        if (H == 1) {  // both halves of this K tile are on their way: advance K (and the bf16x3 operand planes at roll-over)
            a_k0 += 64;
            if (AMODE == MDPT_A_CONV3) {
                if (++a_tap == 9) { a_tap = 0; a_ci += 64; }  // tap-inner K order (see Stager)
            }
            if (a_k0 == p.K) {
                a_k0 = 0; a_tap = 0; a_ci = 0;
                const ptrdiff_t da = (a_pass == 0 ? a_hi_minus_lo : 0) - p.K;
                if (a_pass == 0) conv_plane = p.A_hi;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (AMODE != MDPT_A_CONV3) a_ptr[i] += da;
                ++a_pass;
            }
        }
    }

    template <int H>
    __device__ __forceinline__ void issue_b(const GemmParams& p, char* buf, int wave) {
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        if constexpr (BUFFERED) {
#pragma unroll
            for (int i = 2 * H; i < 2 * H + 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(buf + A_BYTES + (wave + 8 * i) * 1024), 16, b_voff, b_soff + i * b_row_step, 0, 0);
            if (H == 1) {
                b_soff += 128;
                if (b_soff == p.K * 2) {  // bf16x3 passes: W_hi, W_lo, W_hi
                    b_soff = 0;
                    rs_w = rsrc(b_pass == 0 ? w_base_lo : w_base_hi, w_bytes);
                    ++b_pass;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 2 * H; i < 2 * H + 2; ++i) {
            glds16(b_ptr[i], buf + A_BYTES + (wave + 8 * i) * 1024);
            b_ptr[i] += 64;
        }
        if (H == 1) {
            b_k0 += 64;
            if (b_k0 == p.K) {
                b_k0 = 0;
                const ptrdiff_t dw = (b_pass == 0 ? w_lo_minus_hi : -w_lo_minus_hi) - p.K;
#pragma unroll
                for (int i = 0; i < 4; ++i) b_ptr[i] += dw;
                ++b_pass;
            }
        }
    }
};

// FAST (dense rows, one pass - every encoder GEMM of the bf16 mode): operand staging without ANY per-K-tile state. A phase's MFMA block is
// 16 x 16 = 256 cycles and the other wave group's whole load phase has to fit under it, so every instruction of a load phase is on the
// critical path (profiles/r03_gemm8_loop_experiments.txt: ~10 scalar instructions + 4 branches per K tile cost 7 %). Here the operands go
// through buffer descriptors over this tile's rows (LDS-DMA, buffer_load ... lds): ONE per-lane byte offset per operand for the whole
// kernel, the K position and the 64-row step between a wave's DMA instructions are scalar offsets computed from the loop counter, rows past
// M / N fail the bounds check and are staged as zeros, and the last iteration (which issues nothing) is peeled off the loop instead of
// being tested for in every phase.
template <int AMODE, int EKIND, bool SW, bool RI = false, bool FAST = false>
__device__ __forceinline__ void gemm8_body(const GemmParams& p, char* smem, const int dmode, const int m0, const int n0,
                                           const unsigned long long t_start) {
    static_assert(!RI || SW, "residual-initialised accumulators use the swapped operand order (4 consecutive columns per lane)");
    static_assert(!FAST || AMODE == MDPT_A_DENSE, "the stateless staging is for dense rows");
    constexpr int A_BYTES = 256 * 128, BUF = 2 * A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    unsigned long long t_first = 0, t_loop = 0;
    constexpr bool swapped = SW;
    HalfStager<AMODE> st;
    if constexpr (!FAST) st.init(p, m0, n0, wave, lane);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    __amdgpu_buffer_rsrc_t rs_a, rs_w;
    unsigned fa_voff = 0, fb_voff = 0;
    int fa_step = 0, fb_step = 0;
    if constexpr (FAST) {
        const int rows_here = p.M - m0 < 256 ? p.M - m0 : 256, cols_here = p.N - n0 < 256 ? p.N - n0 : 256;
        rs_a = tile_rsrc(p.A_hi + (size_t)m0 * p.lda, (size_t)rows_here * p.lda * 2);
        rs_w = tile_rsrc(p.W_hi + (size_t)n0 * p.ldw, ((size_t)(cols_here - 1) * p.ldw + p.K) * 2);
        const int r = wave * 8 + (lane >> 3);  // DMA instruction i of this wave stages rows r + 64 i (same swizzle key for all four)
        const int koff = ((lane & 7) ^ ((r >> 1) & 7)) * 8;
        fa_voff = (unsigned)(r * p.lda + koff) * 2u;
        fb_voff = (unsigned)(r * p.ldw + koff) * 2u;
        fa_step = 64 * p.lda * 2;
        fb_step = 64 * p.ldw * 2;
    }
    // half-tile H of the K tile at byte offset KB_ (FAST) / of the stager's next K tile (state machine) -> ring buffer BUF_
#define ISSUE_A(H_, BUF_, KB_)                                                                                          \
    do {                                                                                                                \
        if constexpr (FAST) {                                                                                           \
            _Pragma("unroll") for (int i = 2 * (H_); i < 2 * (H_) + 2; ++i)                                             \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (lds_ptr_t)((BUF_) + (wave + 8 * i) * 1024), 16, fa_voff, (KB_) + i * fa_step, 0, 0); \
        } else {                                                                                                        \
            st.template issue_a<H_>(p, BUF_, wave);                                                                     \
        }                                                                                                               \
    } while (0)
#define ISSUE_B(H_, BUF_, KB_)                                                                                          \
    do {                                                                                                                \
        if constexpr (FAST) {                                                                                           \
            _Pragma("unroll") for (int i = 2 * (H_); i < 2 * (H_) + 2; ++i)                                             \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)((BUF_) + 256 * 128 + (wave + 8 * i) * 1024), 16, fb_voff, (KB_) + i * fb_step, 0, 0); \
        } else {                                                                                                        \
            st.template issue_b<H_>(p, BUF_, wave);                                                                     \
        }                                                                                                               \
    } while (0)

    // fragment read offsets (16x16x32 MFMA: lane = (row l&15, 16-byte k-chunk l>>4)); key(row) = (row >> 1) & 7 as staged
    const int l15 = lane & 15, lh = lane >> 4, key = (l15 >> 1) & 7;
    const int a_off0 = (grp * 64 + l15) * 128 + ((lh ^ key) << 4), a_off1 = (grp * 64 + l15) * 128 + (((4 + lh) ^ key) << 4);
    const int b_off0 = A_BYTES + (wc * 32 + l15) * 128 + ((lh ^ key) << 4), b_off1 = A_BYTES + (wc * 32 + l15) * 128 + (((4 + lh) ^ key) << 4);

    f32x4 acc[2][2][4][2];
    if constexpr (!RI) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][b][i][j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }

    opx8 fa[4][2], fb0[2][2], fb1[2][2];
#define PIN() __builtin_amdgcn_sched_barrier(0)
#define BAR() do { PIN(); __builtin_amdgcn_s_barrier(); PIN(); } while (0)
#define LOAD_A(QM_, BUF_)                                                                                             \
    do {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                               \
            fa[i][0] = *(const opx8*)(smem + (BUF_) * BUF + (QM_) * 16384 + i * 2048 + a_off0);                      \
            fa[i][1] = *(const opx8*)(smem + (BUF_) * BUF + (QM_) * 16384 + i * 2048 + a_off1);                      \
        }                                                                                                             \
    } while (0)
#define LOAD_B(DST_, QN_, BUF_)                                                                                       \
    do {                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                               \
            DST_[j][0] = *(const opx8*)(smem + (BUF_) * BUF + (QN_) * 16384 + j * 2048 + b_off0);                    \
            DST_[j][1] = *(const opx8*)(smem + (BUF_) * BUF + (QN_) * 16384 + j * 2048 + b_off1);                    \
        }                                                                                                             \
    } while (0)
#define MFMA_Q(QM_, QN_, FB_)                                                                                         \
    do {                                                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                                \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                              \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                         \
                    acc[QM_][QN_][i][j] = swapped ? MDPT_MFMA_16x16x32(FB_[j][kk], fa[i][kk], acc[QM_][QN_][i][j], 0, 0, 0) \
                                                  : MDPT_MFMA_16x16x32(fa[i][kk], FB_[j][kk], acc[QM_][QN_][i][j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                \
    } while (0)
#define WAIT_LGKM(N_) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" ::: "memory")
#define WAIT_VM(N_) asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory")
// RI form: quadrant (QM_, QN_)'s residual loads (inline asm, in-order return) have landed once at most N_ younger vector-memory operations
// are outstanding; naming the eight accumulators as read-write operands keeps every use of them below this statement
#define ACC_READY(QM_, QN_, N_)                                                                                                          \
    asm volatile("s_waitcnt vmcnt(" #N_ ")"                                                                                              \
                 : "+v"(acc[QM_][QN_][0][0]), "+v"(acc[QM_][QN_][0][1]), "+v"(acc[QM_][QN_][1][0]), "+v"(acc[QM_][QN_][1][1]),           \
                   "+v"(acc[QM_][QN_][2][0]), "+v"(acc[QM_][QN_][2][1]), "+v"(acc[QM_][QN_][3][0]), "+v"(acc[QM_][QN_][3][1])            \
                 :                                                                                                                       \
                 : "memory")

    // ---- prologue: K tiles 0 and 1 complete (even / odd buffer); T >= 2 and even (checked by the launcher)
    const int T = (p.K / 64) * p.npass;
    char* const bufE = smem;
    char* const bufO = smem + BUF;
    ISSUE_A(0, bufE, 0);
    ISSUE_B(0, bufE, 0);
    ISSUE_B(1, bufE, 0);
    ISSUE_A(1, bufE, 0);
    if constexpr (RI) {
        // Accumulators start at the residual tile (out = resid + A W^T + bias, in place): 32 x 16-byte loads per lane in the order the
        // quadrants are first used (P1 (0,0), P2 (0,1), P3 (1,1), P4 (1,0)), issued between the DMAs of K tile 0 and K tile 1. vmcnt retires
        // in order, so the first MFMA block waits for K tile 0 + quadrant (0,0) only; the other 24 loads land under the first phases (counted
        // waits in front of each quadrant's first MFMA in the first iteration, ACC_READY). Lanes outside the tile read 0.
        // The loads are inline asm: beside LDS-DMA hipcc waits vmcnt(0) for any register load it knows about (the whole prologue would
        // drain before the first MFMA); an asm load is invisible to its wait insertion and is waited for by hand (ACC_READY below,
        // which also names the destination registers so that nothing reads them earlier).
        constexpr unsigned OOB = 0xFFFFFFF0u;
        typedef __attribute__((ext_vector_type(4))) int i32x4;
        const int rows_here = p.M - m0 < 256 ? p.M - m0 : 256;
        const unsigned long long base = (unsigned long long)(p.resid + (size_t)m0 * p.ldr);
        i32x4 rs;  // raw buffer descriptor of this tile's rows: base, stride 0, num_records = bytes, dst_sel/format word as tile_rsrc()
        rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)base);
        rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32) & 0xFFFF);
        rs[2] = __builtin_amdgcn_readfirstlane((int)(unsigned)((size_t)rows_here * p.ldr * 4));
        rs[3] = 0x00020000;
        const unsigned row_b = (unsigned)p.ldr * 4u;
        auto load_quadrant = [&](int qm, int qn) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nc = n0 + qn * 128 + wc * 32 + j * 16 + 4 * lh;
                const unsigned col = nc < p.N ? (unsigned)nc * 4u : OOB;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned row = (unsigned)(qm * 128 + grp * 64 + i * 16 + l15) * row_b;
                    const unsigned off = col == OOB ? OOB : row + col;
                    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(acc[qm][qn][i][j]) : "v"(off), "s"(rs) : "memory");
                }
            }
        };
        load_quadrant(0, 0);
        ISSUE_B(0, bufO, 128);
        ISSUE_A(0, bufO, 128);
        ISSUE_B(1, bufO, 128);
        ISSUE_A(1, bufO, 128);
        load_quadrant(0, 1);
        load_quadrant(1, 1);
        load_quadrant(1, 0);
        WAIT_VM(40);  // K tile 0 has landed: 8 (quadrant (0,0)) + 8 (K tile 1) + 24 younger operations may still be in flight
    } else {
        ISSUE_B(0, bufO, 128);
        ISSUE_A(0, bufO, 128);
        ISSUE_B(1, bufO, 128);
        ISSUE_A(1, bufO, 128);
        WAIT_VM(8);
    }
    BAR();
    if (p.dbg_times) t_first = memtime_now();
    if (grp == 1) BAR();  // stagger: group 1 runs one barrier behind group 0


#define GEMM8_ITER(MORE_, FIRST_)                                                                                           \
    do {                                                                                                              \
        const int kE = t * 128 + 256, kO = kE + 128; /* byte offsets of K tiles t + 2 / t + 3 (FAST staging) */        \
        /* P1 */                                                                                                      \
        LOAD_B(fb0, 0, 0); PIN(); LOAD_A(0, 0); PIN();                                                                \
        WAIT_LGKM(8); BAR(); WAIT_LGKM(0); PIN();                                                                     \
        if (FIRST_) { ACC_READY(0, 0, 32); PIN(); } /* younger: 8 (K tile 1) + 24 */                                   \
        MFMA_Q(0, 0, fb0); BAR();                                                                                     \
        /* P2 */                                                                                                      \
        LOAD_B(fb1, 1, 0); PIN();                                                                                     \
        if (MORE_) ISSUE_B(0, bufE, kE);                                                             \
        BAR(); WAIT_LGKM(0); PIN();                                                                                   \
        if (FIRST_) { ACC_READY(0, 1, 18); PIN(); } /* younger: 16 + 2 */                                             \
        MFMA_Q(0, 1, fb1); BAR();                                                                                     \
        /* P3 */                                                                                                      \
        LOAD_A(1, 0); PIN();                                                                                          \
        if (MORE_) ISSUE_A(0, bufE, kE);                                                             \
        PIN(); WAIT_LGKM(0); BAR(); PIN();                                                                            \
        if (FIRST_) { ACC_READY(1, 1, 12); PIN(); } /* younger: 8 + 4 */                                              \
        MFMA_Q(1, 1, fb1); BAR();                                                                                     \
        /* P4 */                                                                                                      \
        if (MORE_) { ISSUE_B(1, bufE, kE); ISSUE_A(1, bufE, kE); PIN(); WAIT_VM(8); } else { WAIT_VM(0); }                      \
        BAR();                                                                                                        \
        if (FIRST_) { ACC_READY(1, 0, 8); PIN(); } /* the phase's own vmcnt(8) already covers the last 8 loads */     \
        MFMA_Q(1, 0, fb0); BAR();                                                                                     \
        /* P5 */                                                                                                      \
        LOAD_B(fb0, 0, 1); PIN(); LOAD_A(0, 1); PIN();                                                                \
        WAIT_LGKM(8); BAR(); WAIT_LGKM(0); PIN();                                                                     \
        MFMA_Q(0, 0, fb0); BAR();                                                                                     \
        /* P6 */                                                                                                      \
        LOAD_B(fb1, 1, 1); PIN();                                                                                     \
        if (MORE_) ISSUE_B(0, bufO, kO);                                                             \
        BAR(); WAIT_LGKM(0); PIN();                                                                                   \
        MFMA_Q(0, 1, fb1); BAR();                                                                                     \
        /* P7 */                                                                                                      \
        LOAD_A(1, 1); PIN();                                                                                          \
        if (MORE_) ISSUE_A(0, bufO, kO);                                                             \
        PIN(); WAIT_LGKM(0); BAR(); PIN();                                                                            \
        MFMA_Q(1, 1, fb1); BAR();                                                                                     \
        /* P8 */                                                                                                      \
        if (MORE_) { ISSUE_B(1, bufO, kO); ISSUE_A(1, bufO, kO); PIN(); WAIT_VM(8); }                                          \
        BAR();                                                                                                        \
        MFMA_Q(1, 0, fb0); BAR();                                                                                     \
    } while (0)
    // RI: the residual loads are waited for in the first iteration only (t == 0: T >= 4 is checked by the launcher, so that iteration's
    // DMA issues all happen and the younger-operation counts of ACC_READY are exact); the accumulators stay in the loop-carried registers
    if constexpr (FAST) {
        // compile-time MORE / FIRST: the first iteration (RI: counted waits for the residual loads) and the last one (issues nothing,
        // drains) are peeled; T >= 4 when RI (launcher)
        int t = 0;
        if constexpr (RI) {
            GEMM8_ITER(true, true);
            t = 2;
        }
        for (; t + 2 < T; t += 2) GEMM8_ITER(true, false);
        GEMM8_ITER(false, false);
    } else {
        for (int t = 0; t < T; t += 2) {
            const bool more = t + 2 < T;  // wave-uniform: the last iteration issues nothing after P1 and drains at P4
            GEMM8_ITER(more, (RI && t == 0));
        }
    }
#undef GEMM8_ITER
#undef ACC_READY
#undef ISSUE_A
#undef ISSUE_B
    if (grp == 0) BAR();  // re-join the two groups
#undef LOAD_A
#undef LOAD_B
#undef MFMA_Q
    if (p.dbg_times) t_loop = memtime_now();
    if constexpr (SW) {
        if (EKIND == MDPT_E_SWQKV) {
            if (dmode == DM_F32) {  // V columns as fp32 rows (swin_v_prep follows)
                if (HAVE_IMGB && p.bias_img_stride) epilogue_direct<DM_F32, false, MDPT_ACT_NONE, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
                else epilogue_direct<DM_F32, false, MDPT_ACT_NONE>(p, acc, m0, n0, grp, wc, lane);
            } else if (HAVE_IMGB && p.bias_img_stride) {
                if (p.q_lo) epilogue_swin_qk<true, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
                else epilogue_swin_qk<false, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
            } else if (p.q_lo) {
                epilogue_swin_qk<true>(p, acc, m0, n0, grp, wc, lane);
            } else {
                epilogue_swin_qk<false>(p, acc, m0, n0, grp, wc, lane);
            }
        } else if (EKIND == MDPT_E_QKV) {
            if (HAVE_IMGB && p.bias_img_stride) {
                if (p.q_lo) epilogue_direct<DM_QK, true, MDPT_ACT_NONE, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
                else epilogue_direct<DM_QK, false, MDPT_ACT_NONE, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
            } else if (p.q_lo) epilogue_direct<DM_QK, true, MDPT_ACT_NONE>(p, acc, m0, n0, grp, wc, lane);
            else epilogue_direct<DM_QK, false, MDPT_ACT_NONE>(p, acc, m0, n0, grp, wc, lane);
        } else if (HAVE_IMGB && dmode == DM_BF16 && p.bias_img_stride) {  // the encoder's fc1 with a per-image bias table (single plane or hi + lo)
            if (p.act == MDPT_ACT_GELU) {
                if (p.out_lo) epilogue_direct<DM_BF16, true, MDPT_ACT_GELU, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
                else epilogue_direct<DM_BF16, false, MDPT_ACT_GELU, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
            } else {
                if (p.out_lo) epilogue_direct<DM_BF16, true, MDPT_ACT_NONE, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
                else epilogue_direct<DM_BF16, false, MDPT_ACT_NONE, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
            }
        } else if (dmode == DM_BF16) {
            const int sel = (p.out_lo ? 3 : 0) + (p.act == MDPT_ACT_GELU ? 2 : (p.act == MDPT_ACT_RELU || p.relu_bf16) ? 1 : 0);
            switch (sel) {
                case 0: epilogue_direct<DM_BF16, false, MDPT_ACT_NONE>(p, acc, m0, n0, grp, wc, lane); break;
                case 1: epilogue_direct<DM_BF16, false, MDPT_ACT_RELU>(p, acc, m0, n0, grp, wc, lane); break;
                case 2: epilogue_direct<DM_BF16, false, MDPT_ACT_GELU>(p, acc, m0, n0, grp, wc, lane); break;
                case 3: epilogue_direct<DM_BF16, true, MDPT_ACT_NONE>(p, acc, m0, n0, grp, wc, lane); break;
                case 4: epilogue_direct<DM_BF16, true, MDPT_ACT_RELU>(p, acc, m0, n0, grp, wc, lane); break;
                default: epilogue_direct<DM_BF16, true, MDPT_ACT_GELU>(p, acc, m0, n0, grp, wc, lane); break;
            }
        } else if (dmode == DM_F32 || dmode == DM_RINIT) {  // DM_RINIT: the residual is already in the accumulators
            if (HAVE_IMGB && p.bias_img_stride) epilogue_direct<DM_F32, false, MDPT_ACT_NONE, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
            else epilogue_direct<DM_F32, false, MDPT_ACT_NONE>(p, acc, m0, n0, grp, wc, lane);
        } else {
            epilogue_direct<DM_RESID, false, MDPT_ACT_NONE>(p, acc, m0, n0, grp, wc, lane);
        }
        if (p.dbg_times && tid == 0) {
            const unsigned long long t_issued = memtime_now();  // all of this wave's stores issued, none waited for
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned long long* d = p.dbg_times + (size_t)blockIdx.x * 6;
            d[0] = t_start; d[1] = t_first; d[2] = t_loop; d[3] = memtime_now();
            d[4] = t_issued;
            d[5] = __builtin_amdgcn_s_getreg(4 << 0 | 0 << 6 | 31 << 11);
        }
        return;
    }
    if ((EKIND == MDPT_E_QKV && dmode == DM_VT) || (EKIND == MDPT_E_SWQKV && dmode == DM_SWVT)) {
        if (EKIND == MDPT_E_SWQKV) {
            if (HAVE_IMGB && p.bias_img_stride) {
                if (p.vt_lo) epilogue_swin_vt<true, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
                else epilogue_swin_vt<false, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
            } else if (p.vt_lo) epilogue_swin_vt<true>(p, acc, m0, n0, grp, wc, lane);
            else epilogue_swin_vt<false>(p, acc, m0, n0, grp, wc, lane);
        } else if (HAVE_IMGB && p.bias_img_stride) {
            if (p.vt_lo) epilogue_direct_vt<true, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
            else epilogue_direct_vt<false, HAVE_IMGB>(p, acc, m0, n0, grp, wc, lane);
        } else if (p.vt_lo) {
            epilogue_direct_vt<true>(p, acc, m0, n0, grp, wc, lane);
        } else {
            epilogue_direct_vt<false>(p, acc, m0, n0, grp, wc, lane);
        }
        if (p.dbg_times && tid == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned long long* d = p.dbg_times + (size_t)blockIdx.x * 6;
            d[0] = t_start; d[1] = t_first; d[2] = t_loop; d[3] = memtime_now(); d[4] = d[3]; d[5] = 0;
        }
        return;
    }
    __syncthreads();  // every wave is done reading the ring: reuse it as epilogue staging

    // ---- epilogue: per quadrant two [32][32] blocks through the wave-private strip (C layout of the 16x16 MFMA:
    //      acc[r] = C[4*(lane>>4) + r][lane&15])
    float* strip = (float*)smem + wave * (32 * 32);
#pragma unroll
    for (int qm = 0; qm < 2; ++qm)
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) strip[(ii * 16 + 4 * lh + r) * 32 + j * 16 + l15] = acc[qm][qn][rb * 2 + ii][j][r];
                epilogue_block<32, EKIND>(p, strip, lane, m0 + qm * 128 + grp * 64 + rb * 32, n0 + qn * 128 + wc * 32);
            }
    if (p.dbg_times && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = p.dbg_times + (size_t)blockIdx.x * 6;
        d[0] = t_start; d[1] = t_first; d[2] = t_loop; d[3] = memtime_now();
        d[4] = __builtin_amdgcn_s_getreg(20 << 0 | 0 << 6 | 3 << 11);
        d[5] = __builtin_amdgcn_s_getreg(4 << 0 | 0 << 6 | 31 << 11);
    }
}
#undef PIN
#undef BAR
#undef WAIT_LGKM
#undef WAIT_VM

// Operand order of the MFMAs: swapped -> accumulators hold 4 consecutive COLUMNS per lane and one of the direct epilogues
// applies; plain order (4 consecutive ROWS per lane) + LDS strip for every other epilogue. For the generic epilogue the choice is
// a property of the launch and is made on the host (DMODE template parameter: one main loop + one epilogue per kernel keeps the
// register allocation of the 256-VGPR loop predictable); QKV launches decide per tile (tiles that contain V columns are written
// transposed, token-contiguous, through the strip).
__host__ __device__ inline int generic_direct_mode(const GemmParams& p) {
    // a per-image bias table needs the IMGB epilogues (fp16 build, >= 256 rows per image, the forms the encoder uses)
    const bool imgb_ok = HAVE_IMGB && p.bias_img_rows >= 256 && p.bias && !p.gamma && !p.relu_bf16 && p.act != MDPT_ACT_RELU;
    const bool plain = !p.up_src && (!p.bias_img_stride || imgb_ok);
    const bool fits = (size_t)256 * p.ldc * 4 < 0xFFFFFFF0ull;  // tile-local byte offsets of the buffer ops are 32-bit
    // relu_bf16 without an fp32 copy is just a ReLU activation (first conv of every residual conv unit)
    if (plain && fits && p.out_hi && !p.out_f32 && !p.gamma && !p.resid && !(p.relu_bf16 && p.act != MDPT_ACT_NONE)) return DM_BF16;
    if (plain && fits && !p.bias_img_stride && p.out_f32 && !p.out_hi && p.gamma && p.resid && p.resid == p.out_f32 && p.bias && p.act == MDPT_ACT_NONE && p.ldr == p.ldc)
        return DM_RESID;
    if (plain && fits && p.acc_init && p.out_f32 && !p.out_hi && !p.gamma && p.resid == p.out_f32 && p.ldr == p.ldc && p.act == MDPT_ACT_NONE && !p.relu_bf16)
        return DM_RINIT;
    if (plain && fits && p.out_f32 && !p.out_hi && !p.gamma && !p.resid && p.act == MDPT_ACT_NONE && !p.relu_bf16) return DM_F32;
    return DM_NONE;
}

template <int AMODE, int EKIND, int DMODE>
__global__ __launch_bounds__(512, 1) void gemm8_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long t_start = 0;
    if (p.dbg_times) t_start = memtime_now();
    int m0, n0;
    tile_coords((p.N + 255) / 256, 256, 256, m0, n0);
    if constexpr (EKIND == MDPT_E_SWQKV) {
        // SwinV2 QKV projection (the host guarantees dense A, 2F % 256 == 0 and 32-bit plane offsets): Q / K tiles -> DM_SWQK; V tiles ->
        // transposed window operand (DM_SWVT, plain operand order) when token runs of 4 stay together, else fp32 rows for swin_v_prep
        const bool fast = p.npass == 1;
        if (n0 >= 2 * p.F && p.swin_vtokmap) {
            if (fast) gemm8_body<AMODE, EKIND, false, false, true>(p, smem, DM_SWVT, m0, n0, t_start);
            else gemm8_body<AMODE, EKIND, false>(p, smem, DM_SWVT, m0, n0, t_start);
        } else {
            const int dm = n0 >= 2 * p.F ? DM_F32 : DM_SWQK;
            if (fast) gemm8_body<AMODE, EKIND, true, false, true>(p, smem, dm, m0, n0, t_start);
            else gemm8_body<AMODE, EKIND, true>(p, smem, dm, m0, n0, t_start);
        }
    } else if (EKIND == MDPT_E_QKV) {
        // per tile: Q / K columns only -> swapped order + head-major direct epilogue; V columns only -> plain order + transposed
        // direct epilogue; a tile that straddles 2F (odd head counts) or planes beyond 32-bit offsets -> plain order + LDS strip
#ifdef MDPT_GEMM8_NO_FAST
        const bool fast = false;
#else
        const bool fast = AMODE == MDPT_A_DENSE && p.npass == 1;  // bf16 mode: stateless operand staging
#endif
        if (n0 + 256 <= 2 * p.F && (size_t)p.M * p.F * 2 < 0xFFFFFFF0ull) {
            if constexpr (AMODE == MDPT_A_DENSE) {
                if (fast) gemm8_body<AMODE, EKIND, true, false, true>(p, smem, DM_QK, m0, n0, t_start);
                else gemm8_body<AMODE, EKIND, true>(p, smem, DM_QK, m0, n0, t_start);
            } else {
                gemm8_body<AMODE, EKIND, true>(p, smem, DM_QK, m0, n0, t_start);
            }
        } else if (n0 >= 2 * p.F && n0 + 256 <= p.N && (size_t)(p.M / p.npad) * p.F * p.npadv * 2 < 0xFFFFFFF0ull) {
            if constexpr (AMODE == MDPT_A_DENSE) {
                if (fast) gemm8_body<AMODE, EKIND, false, false, true>(p, smem, DM_VT, m0, n0, t_start);
                else gemm8_body<AMODE, EKIND, false>(p, smem, DM_VT, m0, n0, t_start);
            } else {
                gemm8_body<AMODE, EKIND, false>(p, smem, DM_VT, m0, n0, t_start);
            }
        } else {
            gemm8_body<AMODE, EKIND, false>(p, smem, DM_NONE, m0, n0, t_start);
        }
    } else if constexpr (AMODE == MDPT_A_DENSE && DMODE == DM_F32) {
        // plain fp32 outputs; with GemmParams::ksplit > 1 (grid.y = range) this workgroup walks K range z of rows that are ldw wide: range 0 is the
        // normal kernel on a shorter K, ranges z >= 1 store bare partial sums (no bias) to the partial planes
        GemmParams q = p;
        if (p.ksplit > 1) {
            const int z = (int)blockIdx.y, ks = p.K / p.ksplit;
            q.K = ks;
            q.A_hi = p.A_hi + z * ks; q.W_hi = p.W_hi + z * ks;
            if (p.npass == 3) { q.A_lo = p.A_lo + z * ks; q.W_lo = p.W_lo + z * ks; }
            if (z > 0) { q.bias = nullptr; q.bias_img_stride = 0; q.out_f32 = p.ks_part + (size_t)(z - 1) * p.M * p.ldc; }
        }
        if (q.npass == 1) gemm8_body<AMODE, EKIND, true, false, true>(q, smem, DMODE, m0, n0, t_start);
        else gemm8_body<AMODE, EKIND, true, false, false>(q, smem, DMODE, m0, n0, t_start);
    } else if constexpr (AMODE == MDPT_A_DENSE && DMODE != DM_NONE) {
        // the encoder's hot forms: stateless staging when there is one pass (bf16 mode), the state-machine stager for bf16x3
#ifdef MDPT_GEMM8_NO_FAST  // A/B builds only
        if (false) {}
#else
        if (p.npass == 1) gemm8_body<AMODE, EKIND, true, DMODE == DM_RINIT, true>(p, smem, DMODE, m0, n0, t_start);
#endif
        else gemm8_body<AMODE, EKIND, true, DMODE == DM_RINIT, false>(p, smem, DMODE, m0, n0, t_start);
    } else {
        gemm8_body<AMODE, EKIND, DMODE != DM_NONE, DMODE == DM_RINIT>(p, smem, DMODE, m0, n0, t_start);
    }
}

template <int AMODE, int EKIND, int DMODE>
int launch_pp_mode(const GemmParams& p, hipStream_t stream) {
    constexpr unsigned LDS = 2 * 65536;
    auto kern = gemm8_kernel<AMODE, EKIND, DMODE>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    static char prof_name[64] = "";
    if (!prof_name[0]) snprintf(prof_name, sizeof(prof_name), "gemm8_kernel<%d, %d, %d>", AMODE, EKIND, DMODE);
    MdptProfScope prof(prof_name, 2.0 * (p.M_alg > 0 ? p.M_alg : p.M) * p.N * p.K, stream);  // algorithmic rows (token pad rows are not work)
    const int ks = (AMODE == MDPT_A_DENSE && EKIND == MDPT_E_GENERIC && DMODE == DM_F32 && p.ksplit > 1) ? p.ksplit : 1;
    hipLaunchKernelGGL(kern, dim3(tiles, ks), dim3(512), LDS, stream, p);
    return (int)hipGetLastError();
}

template <int AMODE, int EKIND>
int launch_pp(const GemmParams& p, hipStream_t stream) {
    if constexpr (EKIND == MDPT_E_GENERIC) {
        const int dmode = generic_direct_mode(p);
        if (dmode == DM_BF16) return launch_pp_mode<AMODE, EKIND, DM_BF16>(p, stream);
        if (dmode == DM_RESID) return launch_pp_mode<AMODE, EKIND, DM_RESID>(p, stream);
        if (dmode == DM_F32) return launch_pp_mode<AMODE, EKIND, DM_F32>(p, stream);
        if constexpr (AMODE == MDPT_A_DENSE) {
            if (dmode == DM_RINIT) return launch_pp_mode<AMODE, EKIND, DM_RINIT>(p, stream);
        }
        return launch_pp_mode<AMODE, EKIND, DM_NONE>(p, stream);
    } else if constexpr (EKIND == MDPT_E_QKV) {
        return launch_pp_mode<AMODE, EKIND, DM_QK>(p, stream);
    } else if constexpr (EKIND == MDPT_E_SWQKV) {
        return launch_pp_mode<AMODE, EKIND, DM_SWQK>(p, stream);
    } else {
        return launch_pp_mode<AMODE, EKIND, DM_NONE>(p, stream);
    }
}

template <int BM, int BN, int WM, int WN, int BK, int NST, int MINW, int AMODE, int EKIND>
int launch_cfg(const GemmParams& p, hipStream_t stream) {
    constexpr int LDS = NST * (BM + BN) * BK * 2;
    static bool attr_done = false;
    auto kern = gemm_kernel<BM, BN, WM, WN, BK, NST, MINW, AMODE, EKIND>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    static char prof_name[112] = "";
    if (!prof_name[0])
        snprintf(prof_name, sizeof(prof_name), "gemm_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %d>", BM, BN, WM, WN, BK, NST, MINW, AMODE, EKIND);
    MdptProfScope prof(prof_name, 2.0 * (p.M_alg > 0 ? p.M_alg : p.M) * p.N * p.K, stream);  // algorithmic flops (real rows, one pass whatever npass is)
    const int ks = (EKIND == MDPT_E_GENERIC && BM == 64 && BN == 64 && BK == 64 && p.ksplit > 1) ? p.ksplit : 1;
    hipLaunchKernelGGL(kern, dim3(tiles, ks), dim3(64 * WM * WN), LDS, stream, p);
    return (int)hipGetLastError();
}

// the tile mdpt_launch_gemm runs for p (MDPT_TILE_AUTO resolved); -1: the 128x64 form of narrow outputs
int resolve_tile(const GemmParams& p) {
    int tile = p.tile;
    if (p.ksplit > 1 && p.ks_ctr) return MDPT_TILE_64x64;  // in-kernel reduction: the small tile only
    if (p.ksplit > 1) {
        // the K split exists on the 64x64 tile and in the DM_F32 form of the 8-phase kernel (>= 4 K tiles per range, in pairs; same sums, same bits):
        // the big tile when all ranges together make enough workgroups (the rule of the unsplit launches, counted over the ranges)
        const long wgs = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * p.ksplit;
        const int kt = (p.K / 64 / p.ksplit) * p.npass;
        const bool pp = p.tile == MDPT_TILE_PP256 || (p.tile == MDPT_TILE_AUTO && wgs >= (p.throughput_mode ? 70 : 140));
        return (pp && p.amode == MDPT_A_DENSE && generic_direct_mode(p) == DM_F32 && kt >= 4 && !(kt & 1) && (p.N & 255) == 0) ? MDPT_TILE_PP256 : MDPT_TILE_64x64;
    }
    if (tile == MDPT_TILE_AUTO) {
        // measured on MI355X, kernel alone on the GPU (tests/gpu_gemm_tile_sweep.py): the 8-phase 256x256 tile wins from ~140
        // tiles (0.55 rounds; one tile takes ~25 us at K = 1024 whatever the count), 64x64 tiles win the latency race while
        // there are <= ~330 128x128 tiles, 128x128 (2 workgroups per CU) in between. When the other half batch runs on a second
        // stream (throughput_mode) idle CUs are not wasted, so the tile with the best CU-time per flop - the big one - is taken
        // earlier.
        const long tiles256 = (long)((p.M + 255) / 256) * ((p.N + 255) / 256);
        const long tiles128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
        // a partial last column tile is accepted while it wastes <= 1/4 of the padded columns (N = 384 -> 2 tiles: measured +1 % on ViT-S)
        const long ncol = (p.N + 255) / 256 * 256;
        const bool cols_ok = (ncol - p.N) * 4 <= ncol;
        // (round 3: 70, i.e. the same ~140 big tiles in flight as when alone, now counted over both streams; it was 24. A 54-tile launch
        // of each half - SwinV2-L stage 2 proj / fc2 at batch 16 - fills 42 % of the CUs with big tiles and 84 % with 128x128 ones:
        // SwinV2-L +1.1 ... 1.6 %, ViT-S batch 32 +0.3 %, ViT-L / BEiT-L (76 tiles) unchanged, profiles/r03_gemm_tile_threshold_ab.txt)
        const bool big = cols_ok && tiles256 >= (p.throughput_mode ? 70 : 140);
        // (round 4: 352, was 330 - fc1 of ViT-L at batch 1, M = 1304, N = 4096, is exactly 352 tiles: 22.5 us on the 64x64 tile against 24.4 us,
        // profiles/r04_b1_tile_sweep.txt; same bits on every tile)
        tile = big ? MDPT_TILE_PP256 : (tiles128 <= 352 ? MDPT_TILE_64x64 : MDPT_TILE_128x128);
        // narrow outputs with many rows (64-channel decoder convs of the small models): a 128-wide tile would spend half of its
        // MFMAs on padding columns. 128x64, four waves stacked in M: ViT-S B=32 +8.4 %
        if (!big && tiles128 > 352 && p.N <= 64) return -1;
    }
    // odd number of K tiles: the 8-phase loop handles pairs. Three K tiles or fewer (SwinV2's 192-wide first stage) are all prologue and
    // epilogue on a big tile: the 64x64 tile wins there (M = 147456, K = 192: N = 576 93.8 vs 116.4 us, N = 192 33.0 vs 47.3 us, N = 768 a tie;
    // tools/probes/gpu_swin_tile_sweep.py, profiles/r04_swin_tile_sweep.txt), the lockstep 256x256 tile from five K tiles on
    if (tile == MDPT_TILE_PP256 && (((p.K / 64) * p.npass) & 1)) tile = (p.K / 64) * p.npass <= 3 && p.tile == MDPT_TILE_AUTO ? MDPT_TILE_64x64 : MDPT_TILE_256x256;
    // per-image bias table: the direct epilogues of the 8-phase kernel take it in the fp16 build for images of >= 256 rows (two images per
    // tile at most); everything else goes through the strip epilogues of the lockstep kernels (same arithmetic, same bits)
    if (tile == MDPT_TILE_PP256 && p.bias_img_stride && p.ekind == MDPT_E_QKV && !(HAVE_IMGB && p.bias_img_rows >= 256)) tile = MDPT_TILE_256x256;
    return tile;
}
template <int AMODE, int EKIND>
int launch_tile(const GemmParams& p, hipStream_t stream) {
    int tile = resolve_tile(p);
    if (tile < 0) return launch_cfg<128, 64, 4, 1, 64, 2, 2, AMODE, EKIND>(p, stream);
    if constexpr (EKIND == MDPT_E_GENERIC) {
        // residual-initialised accumulators: the 8-phase kernel has them in its DM_RINIT form only (dense A, >= 4 K tiles, 32-bit tile
        // offsets); anything else runs the lockstep 256x256 tile, whose prologue loads the residual the same way
        if (tile == MDPT_TILE_PP256 && p.acc_init &&
            !(AMODE == MDPT_A_DENSE && generic_direct_mode(p) == DM_RINIT && (p.K / 64) * p.npass >= 4))
            tile = MDPT_TILE_256x256;
    }
    if (tile == MDPT_TILE_64x64) {
        // long K on the small tile (fc2 at batch 1: 64 K tiles per workgroup, a serial chain): a 3-deep ring hides more of the L2 latency
        // per step - same K order, same bits. Measured on the bare kernels (profiles/r04_b1_tile_sweep.txt, column t7): K = 4096 32.6 -> 28.5 us
        // (ViT-L, M = 1304), 30.7 -> 22.4 us (BEiT-L, M = 584); K <= 1536 is 2-8 % slower with the deeper ring and keeps the 2-deep one.
        if constexpr ((AMODE == MDPT_A_DENSE || AMODE == MDPT_A_CONV3) && EKIND == MDPT_E_GENERIC) {
            if ((p.K / 64 / (p.ksplit > 1 ? p.ksplit : 1)) * p.npass >= 32) return launch_cfg<64, 64, 2, 2, 64, 3, 1, AMODE, EKIND>(p, stream);
        }
        return launch_cfg<64, 64, 2, 2, 64, 2, 1, AMODE, EKIND>(p, stream);
    }
    if (tile == MDPT_TILE_PP256) return launch_pp<AMODE, EKIND>(p, stream);
    if (tile == MDPT_TILE_256x128) return launch_cfg<256, 128, 2, 2, 32, 3, 2, AMODE, EKIND>(p, stream);
    if (tile == MDPT_TILE_256x256) return launch_cfg<256, 256, 2, 4, 64, 2, 1, AMODE, EKIND>(p, stream);
    return launch_cfg<128, 128, 2, 2, 64, 2, 1, AMODE, EKIND>(p, stream);
}

}  // namespace

bool MDPT_FN(mdpt_gemm_resolves_to_pp256)(const GemmParams& p) { return p.M > 0 && p.N > 0 && p.K > 0 && !(p.K & 63) && resolve_tile(p) == MDPT_TILE_PP256; }

int MDPT_FN(mdpt_launch_gemm)(const GemmParams& p_in, hipStream_t stream) {
    GemmParams p = p_in;
    if (p.ldw <= 0) p.ldw = p.K;  // packed panels: rows are K wide
    if (p.M <= 0 || p.N <= 0) return 0;
    if (p.K <= 0 || (p.K & 63) || (p.N & 7)) return (int)hipErrorInvalidValue;
    if (p.npass != 1 && p.npass != 3) return (int)hipErrorInvalidValue;
    if (p.ksplit <= 1 && p.ks_auto && p.ks_ctr && p.ks_part && p.ekind == MDPT_E_GENERIC && p.tile == MDPT_TILE_AUTO && resolve_tile(p) == MDPT_TILE_64x64) {
        // latency mode: few workgroups, each walking a long K (the small decoder convs of a batch of one: 24 ... 96 workgroups x 36 ... 144 K
        // tiles). Smallest divisor of the K-tile count that brings the launch to >= 256 workgroups while a range keeps >= 6 K tiles.
        const long tiles = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
        const int kt = p.K / 64;
        constexpr int lim = 400, tgt = 256, minr = 6;  // (swept on the GPU: 192 / 192 ... 700 / 512 workgroups, 4 ... 9 K tiles per range - flat within 0.6 %)
        if (tiles < lim && kt >= 2 * minr) {
            int best = 1;
            for (int d = 2; d <= 8 && kt / d >= minr; ++d)
                if (kt % d == 0) { best = d; if (tiles * d >= tgt) break; }
            if (best > 1 && tiles <= p.ks_ctr_n && (size_t)best * tiles * 64 * 64 * 4 <= p.ks_cap) p.ksplit = best;
        }
        if (p.ksplit <= 1) p.ks_ctr = nullptr;
    }
    if (p.ksplit > 1 && p.ks_ctr) {
        const long tiles = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
        if (p.ekind != MDPT_E_GENERIC || !p.ks_part || (p.K / 64) % p.ksplit || p.ldw != p.K || tiles > p.ks_ctr_n ||
            (size_t)p.ksplit * tiles * 64 * 64 * 4 > p.ks_cap)
            return (int)hipErrorInvalidValue;
    } else if (p.ksplit > 1 && (p.ekind != MDPT_E_GENERIC || p.amode != MDPT_A_DENSE || !p.ks_part || !p.out_f32 || p.out_hi || p.up_src || p.gamma ||
                         p.act != MDPT_ACT_NONE || (p.K / 64) % p.ksplit || p.ldw != p.K))
        return (int)hipErrorInvalidValue;  // the split is for fp32 outputs whose consumer adds the partial sums
    switch (p.ekind) {
        case MDPT_E_GENERIC:
            if (p.amode == MDPT_A_DENSE) return launch_tile<MDPT_A_DENSE, MDPT_E_GENERIC>(p, stream);
            if (p.amode == MDPT_A_TOKENS) return launch_tile<MDPT_A_TOKENS, MDPT_E_GENERIC>(p, stream);
            if (p.amode == MDPT_A_CONV3) return launch_tile<MDPT_A_CONV3, MDPT_E_GENERIC>(p, stream);
            break;
        case MDPT_E_QKV:
            if (p.amode == MDPT_A_DENSE && (p.F & 63) == 0 && (p.npad & 7) == 0 && (p.npadv & 7) == 0)
                return launch_tile<MDPT_A_DENSE, MDPT_E_QKV>(p, stream);
            break;
        case MDPT_E_SWQKV:  // exists in the 8-phase kernel only: the caller asks mdpt_gemm_resolves_to_pp256() first
            if (p.amode == MDPT_A_DENSE && p.swin_tokmap && (2 * p.F) % 256 == 0 && p.N == 3 * p.F && (p.npadv & 3) == 0 &&
                resolve_tile(p) == MDPT_TILE_PP256)
                return launch_pp<MDPT_A_DENSE, MDPT_E_SWQKV>(p, stream);
            break;
        case MDPT_E_PATCH:
            if (p.amode == MDPT_A_DENSE) return launch_tile<MDPT_A_DENSE, MDPT_E_PATCH>(p, stream);
            break;
        case MDPT_E_D2S:
            if (p.amode == MDPT_A_DENSE && (p.d2s_cout & 7) == 0) return launch_tile<MDPT_A_DENSE, MDPT_E_D2S>(p, stream);
            break;
        case MDPT_E_HEAD:
            if (p.amode == MDPT_A_CONV3 && p.N == 32) return launch_cfg<128, 32, 4, 1, 64, 2, 1, MDPT_A_CONV3, MDPT_E_HEAD>(p, stream);
            break;
    }
    return (int)hipErrorInvalidValue;
}

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
// File layout (round 4: one 2100-line file before): this file = the launchers, the tile rule and the C entry point; the kernels are textually
// included parts - gemm_common.inc (helpers, lockstep operand stager, tile map), gemm_epilogue_strip.inc (LDS-strip epilogues of every
// epilogue kind), gemm_lockstep.inc (gemm_kernel), gemm8_epilogues.inc (direct epilogues), gemm8.inc (gemm8_kernel).
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
// fp32 accumulators: fp32-class accuracy from bf16 MFMAs. npass == 2 is the activation-split form (A_lo*W_hi, A_hi*W_hi): the
// activations keep their hi + lo planes, the weights one rounded plane - for the decoder classes whose error is the rounding of their
// activations (profiles/r05_precision_budget.md).
// The epilogue stages each wave's accumulators through its private LDS strip so that global stores are
// row-major 16-byte (fp32) / 8-byte (bf16) vectors.
//
// K split (GemmParams::ksplit): K in equal ranges over grid.y, the CONSUMER adds the partial planes (the LayerNorm behind a residual GEMM):
// the 64x64 tile (latency mode: proj / fc2 of a small batch) and the DM_F32 form of the 8-phase kernel (SwinV2's fc2, every batch size). A
// split is fixed per shape. With GemmParams::ks_all EVERY range stores a bare partial plane and a finishing kernel (elementwise.hip) adds them
// and applies the epilogue - the long-K decoder convs of a batch of one (latency mode). (A form that reduces inside the kernel - last
// workgroup to arrive, fixed order - was built and removed: between XCDs it needs a device-scope release that costs what the split saves,
// DESIGN.md section 3.)
//
// Workgroup -> tile mapping is XCD-aware: the dispatcher places block b on XCD b%8 (8 private L2s), so
// each XCD is given a contiguous run of tiles (same A rows, all N tiles) to keep operand panels L2-resident.

#include "mdpt_kernels.h"
#include "mdpt_prof.h"
#include "f8_cross.h"
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

#include "gemm_common.inc"            // helpers, lockstep operand stager, tile map
#include "gemm_epilogue_strip.inc"    // LDS-strip epilogues (all epilogue kinds)
#include "gemm_lockstep.inc"          // gemm_kernel: lockstep main loop, K-split forms of the 64x64 tile
#include "gemm8_epilogues.inc"        // direct epilogues of the 8-phase kernel
#include "gemm8.inc"                  // gemm8_kernel: 8-phase main loop

template <int AMODE, int EKIND, int DMODE, bool F8 = false>
int launch_pp_mode(const GemmParams& p, hipStream_t stream) {
    constexpr unsigned LDS = 2 * 65536;
    if constexpr (MDPT_OP_IS_F16 && !F8 && (EKIND == MDPT_E_GENERIC || EKIND == MDPT_E_D2S) && DMODE != DM_RINIT) {
        if (p.f8) return launch_pp_mode<AMODE, EKIND, DMODE, true>(p, stream);  // fp8 cross terms: their own kernels
    }
    auto kern = gemm8_kernel<AMODE, EKIND, DMODE, F8>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    static char prof_name[64] = "";
    if (!prof_name[0]) snprintf(prof_name, sizeof(prof_name), F8 ? "gemm8_kernel<%d, %d, %d, f8>" : "gemm8_kernel<%d, %d, %d>", AMODE, EKIND, DMODE);
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

template <int BM, int BN, int WM, int WN, int BK, int NST, int MINW, int AMODE, int EKIND, bool F8 = false>
int launch_cfg(const GemmParams& p, hipStream_t stream) {
    constexpr int LDS = NST * (BM + BN) * BK * 2;
    if constexpr (MDPT_OP_IS_F16 && !F8 && BK == 64 && (EKIND == MDPT_E_GENERIC || EKIND == MDPT_E_D2S) && !(BM == 256 && BN == 256)) {
        if (p.f8) return launch_cfg<BM, BN, WM, WN, BK, NST, MINW, AMODE, EKIND, true>(p, stream);  // fp8 cross terms: their own kernels
    }
    static bool attr_done = false;
    auto kern = gemm_kernel<BM, BN, WM, WN, BK, NST, MINW, AMODE, EKIND, F8>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_done = true;
    }
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    static char prof_name[112] = "";
    if (!prof_name[0])
        snprintf(prof_name, sizeof(prof_name), F8 ? "gemm_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %d, f8>" : "gemm_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %d>", BM, BN, WM, WN, BK, NST, MINW, AMODE, EKIND);
    MdptProfScope prof(prof_name, 2.0 * (p.M_alg > 0 ? p.M_alg : p.M) * p.N * p.K, stream);  // algorithmic flops (real rows, one pass whatever npass is)
    const int ks = ((AMODE == MDPT_A_DENSE || AMODE == MDPT_A_CONV3) && EKIND == MDPT_E_GENERIC && BM == 64 && BN == 64 && BK == 64 && p.ksplit > 1) ? p.ksplit : 1;
    hipLaunchKernelGGL(kern, dim3(tiles, ks), dim3(64 * WM * WN), LDS, stream, p);
    return (int)hipGetLastError();
}

// the tile mdpt_launch_gemm runs for p (MDPT_TILE_AUTO resolved); -1: the 128x64 form of narrow outputs
int resolve_tile(const GemmParams& p) {
    int tile = p.tile;
    if (p.ksplit > 1 && p.ks_all) return MDPT_TILE_64x64;  // all-partial form (a finishing kernel follows): the small tile only
    if (p.ksplit > 1) {
        // the K split exists on the 64x64 tile and in the DM_F32 form of the 8-phase kernel (>= 4 K tiles per range, in pairs; same sums, same bits):
        // the big tile when all ranges together make enough workgroups (the rule of the unsplit launches, counted over the ranges)
        const long wgs = (long)((p.M + 255) / 256) * ((p.N + 255) / 256) * p.ksplit;
        const int kt = (p.K / 64 / p.ksplit) * p.npass;  // (no fp8 cross terms with a K split: mdpt_launch_gemm)
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
    if (tile == MDPT_TILE_PP256 && (((p.K / 64) * p.npass) & 1) && !p.f8) tile = (p.K / 64) * p.npass <= 3 && p.tile == MDPT_TILE_AUTO ? MDPT_TILE_64x64 : MDPT_TILE_256x256;
    // fp8 cross terms: a pair of K tiles must not straddle a pass - every pass an even number of tiles (K % 256 == 0), else the lockstep tile
    if (tile == MDPT_TILE_PP256 && p.f8 && ((p.K & 255) || (p.N & 255))) tile = MDPT_TILE_128x128;  // (and whole 256-column tiles: HalfStager steps one weight pointer)
    if (tile == MDPT_TILE_256x256 && p.f8) tile = MDPT_TILE_128x128;  // (the 256x256 lockstep tile has no registers left for the second MFMA family)
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
            !(AMODE == MDPT_A_DENSE && generic_direct_mode(p) == DM_RINIT && (p.K / 64) * p.npass >= 4 && !p.f8))
            tile = MDPT_TILE_256x256;
    }
    if (tile == MDPT_TILE_64x64) {
        // long K on the small tile (fc2 at batch 1: 64 K tiles per workgroup, a serial chain): a 3-deep ring hides more of the L2 latency
        // per step - same K order, same bits. Measured on the bare kernels (profiles/r04_b1_tile_sweep.txt, column t7): K = 4096 32.6 -> 28.5 us
        // (ViT-L, M = 1304), 30.7 -> 22.4 us (BEiT-L, M = 584); K <= 1536 is 2-8 % slower with the deeper ring and keeps the 2-deep one.
        if constexpr ((AMODE == MDPT_A_DENSE || AMODE == MDPT_A_CONV3) && EKIND == MDPT_E_GENERIC) {
            if (ktiles_total(p, p.K / (p.ksplit > 1 ? p.ksplit : 1)) >= 32) return launch_cfg<64, 64, 2, 2, 64, 3, 1, AMODE, EKIND>(p, stream);
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
    if (p.npass < 1 || p.npass > 3 || (p.npass >= 2 && !p.A_lo) || (p.npass == 3 && !p.f8 && !p.W_lo)) return (int)hipErrorInvalidValue;
    if (p.f8) {  // fp8 cross terms (f8_cross.h): fp16 build, 128-element K tiles of byte planes, packed panels, the generic / depth-to-space epilogues, no K split
        if (!MDPT_OP_IS_F16 || p.npass < 2 || (p.K & 127) || !p.W8 || !p.S8 || (p.npass == 3 && (!p.W8_lo || !p.S8_lo || !p.a8_off)) || p.ksplit > 1 || p.ldw != p.K ||
            (p.ekind != MDPT_E_GENERIC && p.ekind != MDPT_E_D2S) || p.acc_init || p.tile == MDPT_TILE_256x128 || p.tile == MDPT_TILE_128x32)
            return (int)hipErrorInvalidValue;
    }
    if (p.out_f8 && (!MDPT_OP_IS_F16 || !p.out_lo)) return (int)hipErrorInvalidValue;
    if (p.ksplit > 1 && p.ks_all) {
        if (p.ekind != MDPT_E_GENERIC || (p.amode != MDPT_A_DENSE && p.amode != MDPT_A_CONV3) || !p.ks_part || (p.K / 64) % p.ksplit || p.ldw != p.K || p.acc_init)
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

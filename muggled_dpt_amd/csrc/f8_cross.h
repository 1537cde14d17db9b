// fp8 CROSS TERMS of the split products (gfx950, fp16 operand build only; round 6).
//
// A class at 2 or 3 "passes" computes  A W^T  as  A_hi W_hi^T + A_lo W_hi^T (+ A_hi W_lo^T)  with A = A_hi + A_lo, W = W_hi + W_lo split into
// fp16 planes. The cross terms are 2^-11 of the main term: they need 3-4 significant bits, not 11. In the F8 forms of a class
// (MDPT_PASSES_2F8 / _3F8) they run on the block-scaled MFMA  v_mfma_scale_f32_{16x16x128,32x32x64}_f8f6f4  at twice the fp16 rate, with
// STATIC scales - no per-block scale arithmetic anywhere:
//
//   activations   E5M2 (fp16's exponent range, 2 significand bits), produced by the epilogue that produces the fp16 plane:
//                   lo8 = e5m2((A - A_hi) * 2^16)   the residue is <= 2^-11 |A|: the shift keeps the residue of every normal fp16 value in e5m2's
//                                                    normal range (a plain e5m2 of the residue would be subnormal below |A| = 2^-3)
//                   a8  = e5m2(A_hi)                 (3F8 only: the operand of the A_hi W_lo^T term)
//                 both enter the MFMA with ONE constant E8M0 scale byte, 2^-16 (F8_A_SCALE): the a8 term's 2^16 is folded into its weight rows' scale.
//   weights       E4M3 with one power-of-two scale per OUTPUT ROW (pack time): w8 = e4m3(W_hi / 2^e_row), row maximum in [256, 448]; a float format
//                 keeps its 3 significand bits over 15 binades below the row maximum, so the MX block granularity (32 K elements) buys nothing
//                 here, and a per-row scale is a register constant of the kernel (4 bytes per lane: one per 16-row block the lane touches).
//                   wlo8 = e4m3((W - W_hi) / 2^e'_row), scale byte e'_row + 16 + 127.
//   Measured premise: tests/precision_budget/emulate_operand_rounding.py --study lowlo (format "sf8") - the mixed table with every cross term
//   in this form reads 6.8e-4 / 8.1e-4 on ViT-L images 0 / 31 against 7.3e-4 / 8.3e-4 with fp16 cross terms. Instruction semantics pinned by
//   tools/probes/mfma_scale_probe.hip (layout, scale rows) and tools/probes/f8_cross_probe.hip (formats, op_sel, conversions).
//
// K layout: an fp8 K tile is 128 elements = 128 bytes per row - the byte geometry of a 64-element fp16 K tile, so the LDS images, the DMA
// stagers and the fragment reads of every kernel are unchanged; a lane's two 16-byte fragment reads are the 32 operand bytes of ONE
// 16x16x128 MFMA (or of one 32x32x64 MFMA per pair of k-steps). Both operands permute K the same way, and with per-row / constant scales the
// hardware's scale-block structure does not care which 32 elements share a block.
#pragma once
#include "op_types.h"

#if MDPT_OP_IS_F16 && (defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__))
#define MDPT_HAVE_F8 1
constexpr int F8_LO_SHIFT = 16;                    // lo8 = e5m2(residue * 2^16)
constexpr int F8_A_SCALE = 127 - F8_LO_SHIFT;      // E8M0 byte of every activation operand
constexpr float F8_E5M2_MAX = 57344.0f;

typedef __attribute__((ext_vector_type(8))) int f8_i32x8;
typedef __attribute__((ext_vector_type(4))) float f8_f32x4;
typedef __attribute__((ext_vector_type(16))) float f8_f32x16;

// two fp32 -> two e5m2 bytes in the low (hi = false) or high half of a dword, round to nearest even, saturating at +-57344
// (the conversion itself produces inf beyond the format's range: clamp first; NaN stays NaN through v_med3? no - med3 returns a finite value, the
// same convention as op_sat() of this build)
__device__ __forceinline__ unsigned f8_pk_e5m2(float a, float b, unsigned old, bool hi) {
    a = __builtin_amdgcn_fmed3f(a, -F8_E5M2_MAX, F8_E5M2_MAX);
    b = __builtin_amdgcn_fmed3f(b, -F8_E5M2_MAX, F8_E5M2_MAX);
    return hi ? (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, (int)old, true) : (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, (int)old, false);
}
// the same without the clamp, for values known to be inside the range (a residue times 2^16: |.| <= 2^-11 * 65504 * 2^16 overflows only for
// |A| > 1792 - NOT bounded; the fp16 value itself: bounded by 65504 > 57344 - NOT bounded either. Kept for the halo builders whose values are bounded.)
__device__ __forceinline__ unsigned f8_pk_e5m2_bounded(float a, float b, unsigned old, bool hi) {
    return hi ? (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, (int)old, true) : (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(a, b, (int)old, false);
}

// four values v[0..3] and their fp16 roundings h[0..3] (as fp32) -> one dword of lo8 bytes / one dword of a8 bytes
__device__ __forceinline__ unsigned f8_lo8x4(float v0, float v1, float v2, float v3, float h0, float h1, float h2, float h3) {
    const float s = 65536.0f;
    unsigned d = f8_pk_e5m2((v0 - h0) * s, (v1 - h1) * s, 0u, false);
    return f8_pk_e5m2((v2 - h2) * s, (v3 - h3) * s, d, true);
}
__device__ __forceinline__ unsigned f8_a8x4(float h0, float h1, float h2, float h3) {
    unsigned d = f8_pk_e5m2(h0, h1, 0u, false);
    return f8_pk_e5m2(h2, h3, d, true);
}

// ---- the MFMAs, accumulating IN PLACE. "W first": the weight fragment is the first operand (the swapped operand order of the 8-phase kernels: a
// lane owns four consecutive output columns), "A first": the activation fragment is. cbsz = format of the first operand, blgp = of the second
// (0 = e4m3, 1 = e5m2). WSEL: which byte of the weight-scale register (one byte per 16- / 32-row block the lane owns): bit 0 -> op_sel, bit 1 ->
// op_sel_hi of that operand (tools/probes/f8_cross_probe.hip).
// Inline asm, not __builtin_amdgcn_mfma_scale_*: with the builtin hipcc (ROCm 7.2) does not tie the destination to the accumulator input - the
// results rotate through the registers of dead operands, ~30 VGPRs more live in a 256-register kernel (conv3h 128-channel form: 196 against 168
// with fp16 MFMAs in the same loop; the 256-channel forms spilled 200 ... 7000 bytes). "+v" ties them. The compiler does not know the statement is
// an MFMA: what it would have inserted by itself is ours to provide - f8_mfma_settle() between the last of these MFMAs and the first NON-MFMA
// reader of an accumulator (MFMA -> MFMA accumulation on the same registers needs nothing: the pipe interlocks, as in every K loop).
#define F8_MFMA_ASM(NAME_, FMT_, SEL_, SELHI_)                                                                              \
    asm volatile(NAME_ " %0, %1, %2, %0, %3, %4 op_sel:" SEL_ " op_sel_hi:" SELHI_ FMT_                                     \
                 : "+v"(acc) : "v"(first), "v"(second), "v"(s_first), "v"(s_second), "n"(WSEL & 1), "n"(WSEL >> 1))
template <int WSEL>
__device__ __forceinline__ void f8_mfma16_w_first(f8_f32x4& acc, f8_i32x8 first, f8_i32x8 second, int s_first, int s_second) {
    F8_MFMA_ASM("v_mfma_scale_f32_16x16x128_f8f6f4", " blgp:1", "[%5,0,0]", "[%6,0,0]");
}
template <int WSEL>
__device__ __forceinline__ void f8_mfma16_a_first(f8_f32x4& acc, f8_i32x8 first, f8_i32x8 second, int s_first, int s_second) {
    F8_MFMA_ASM("v_mfma_scale_f32_16x16x128_f8f6f4", " cbsz:1", "[0,%5,0]", "[0,%6,0]");
}
// The 32x32x64 form of the lockstep GEMM tiles is the BUILTIN: those kernels keep their accumulators in AGPRs between the fp16 MFMAs, an asm
// statement with "+v" makes hipcc copy them to VGPRs and back around it - and the copy back (v_accvgpr_write) reads the MFMA's result one cycle
// after an instruction the compiler does not know to be an MFMA: the last MFMA of every fp8 K tile was lost (found by tests/test_gpu_f8_cross.py
// as cross terms that were "half there"). The lockstep kernels have the registers the untied builtin wants.
template <int WSEL>
__device__ __forceinline__ void f8_mfma32_a_first(f8_f32x16& acc, f8_i32x8 first, f8_i32x8 second, int s_first, int s_second) {
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(first, second, acc, 1, 0, 0, s_first, WSEL, s_second);
}
#undef F8_MFMA_ASM
// (16 passes of 4 cycles at most before a result may be read by anything but the matrix pipe)
__device__ __forceinline__ void f8_mfma_settle() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory"); }
// two 16-byte fragment reads -> the 32 operand bytes of one MFMA
__device__ __forceinline__ f8_i32x8 f8_cat(opx8 lo, opx8 hi) {
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    const i32x4 a = __builtin_bit_cast(i32x4, lo), b = __builtin_bit_cast(i32x4, hi);
    return f8_i32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
#else
#define MDPT_HAVE_F8 0
#endif

// ---- producer side: the lo plane(s) of consecutive elements of a row, in whichever form the CONSUMING class reads (GemmParams::out_f8).
// `lo` is the plane pointer the 16-bit form would use, `off` the element offset; f8_elems != 0 selects bytes. Works in both builds (the bf16
// build has no fp8 form: f8_elems is never set there).
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
typedef __attribute__((ext_vector_type(4))) float f8s_f32x4;
__device__ __forceinline__ void lo_store4(op_t* lo, size_t off, size_t f8_elems, int a8, f8s_f32x4 v, opx4 h) {
#if MDPT_HAVE_F8
    if (f8_elems) {
        unsigned char* b = (unsigned char*)lo;
        *(unsigned*)(b + off) = f8_lo8x4(v[0], v[1], v[2], v[3], (float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        if (a8) *(unsigned*)(b + f8_elems + off) = f8_a8x4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        return;
    }
#endif
    opx4 l;
#pragma unroll
    for (int e = 0; e < 4; ++e) l[e] = to_op(v[e] - (float)h[e]);
    *(opx4*)(lo + off) = l;
}
__device__ __forceinline__ void lo_store8(op_t* lo, size_t off, size_t f8_elems, int a8, f8s_f32x4 v0, f8s_f32x4 v1, opx8 h) {
#if MDPT_HAVE_F8
    if (f8_elems) {
        typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
        unsigned char* b = (unsigned char*)lo;
        const u32x2 l8 = {f8_lo8x4(v0[0], v0[1], v0[2], v0[3], (float)h[0], (float)h[1], (float)h[2], (float)h[3]),
                          f8_lo8x4(v1[0], v1[1], v1[2], v1[3], (float)h[4], (float)h[5], (float)h[6], (float)h[7])};
        *(u32x2*)(b + off) = l8;
        if (a8) {
            const u32x2 a = {f8_a8x4((float)h[0], (float)h[1], (float)h[2], (float)h[3]), f8_a8x4((float)h[4], (float)h[5], (float)h[6], (float)h[7])};
            *(u32x2*)(b + f8_elems + off) = a;
        }
        return;
    }
#endif
    opx8 l;
#pragma unroll
    for (int e = 0; e < 4; ++e) { l[e] = to_op(v0[e] - (float)h[e]); l[e + 4] = to_op(v1[e] - (float)h[e + 4]); }
    *(opx8*)(lo + off) = l;
}
#endif

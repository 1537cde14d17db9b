// The 16-bit MFMA OPERAND format of a translation unit (gfx950). Every kernel file that touches operand planes is compiled twice:
//
//   (default)      op_t = __bf16     v_mfma_f32_*_bf16   launcher symbols  mdpt_launch_*_bf16
//   -DMDPT_OP_F16  op_t = _Float16   v_mfma_f32_*_f16    launcher symbols  mdpt_launch_*_f16
//
// Both MFMA families run at the same rate on CDNA4; fp16 carries 11 significand bits instead of 8 (8x smaller operand rounding) and
// a 5-bit exponent, so every fp32 -> fp16 conversion of an unbounded value saturates at +-65504 (one v_med3_f32) instead of
// producing inf. fp32 accumulation, the fp32 residual stream, LayerNorm and the softmax statistics are the same in both builds.
// This is the only place that knows which format a build uses; kernels use op_t / to_op* / op2_to_f32 / MDPT_MFMA_*.
#pragma once

#if defined(MDPT_OP_F16)
typedef _Float16 op_t;
#define MDPT_FN(name) name##_f16
#define MDPT_OP_IS_F16 1
#define MDPT_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define MDPT_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __bf16 op_t;
#define MDPT_FN(name) name##_bf16
#define MDPT_OP_IS_F16 0
#define MDPT_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MDPT_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
typedef __attribute__((ext_vector_type(8))) op_t opx8;
typedef __attribute__((ext_vector_type(4))) op_t opx4;
typedef __attribute__((ext_vector_type(2))) op_t opx2;
typedef __attribute__((ext_vector_type(2))) float op_f32x2;

// fp32 -> operand, round to nearest even; saturating in the fp16 build (bf16 has fp32's exponent range).
// NaN: v_med3_f32 returns a FINITE value for a NaN input (min/max semantics), so in the fp16 operand modes a NaN produced upstream does not
// propagate through an operand conversion - it becomes -65504 - where the bf16 build's conversions (and the reference's own float16 path) carry
// it on. (The ReLUs of both builds are v_max_f32, which also map NaN to 0 where torch.relu propagates it.) A NaN-preserving clamp costs a
// compare + select per converted value in VALU-bound epilogues; the behaviour is documented instead (include/mdpt.h MDPT_PREC_FP16,
// tests/test_gpu_precision_modes.py::test_fp16_operand_converts_swallow_nan_documented): validate inputs with torch.isfinite where it matters.
__device__ __forceinline__ float op_sat(float v) {
#if MDPT_OP_IS_F16
    return __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
#else
    return v;
#endif
}
__device__ __forceinline__ op_t to_op(float v) { return (op_t)op_sat(v); }
// a packed pair: one v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 (+ two v_med3_f32 in the fp16 build)
__device__ __forceinline__ opx2 to_op2(op_f32x2 v) {
    const op_f32x2 s = {op_sat(v[0]), op_sat(v[1])};
    return __builtin_convertvector(s, opx2);
}
// the same for values known to be bounded (softmax probabilities): no saturation
__device__ __forceinline__ opx2 to_op2_bounded(op_f32x2 v) { return __builtin_convertvector(v, opx2); }
// packed operand pair (one dword) -> two fp32
__device__ __forceinline__ op_f32x2 op2_to_f32(unsigned d) {
#if MDPT_OP_IS_F16
    return __builtin_convertvector(__builtin_bit_cast(opx2, d), op_f32x2);
#else
    return op_f32x2{__builtin_bit_cast(float, d << 16), __builtin_bit_cast(float, d & 0xFFFF0000u)};
#endif
}
#endif

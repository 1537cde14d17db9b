// Bilinear interpolation (align_corners=True: SpatialUpsampleLayer, v2_depthanything/components/misc_helpers.py:39-42) of EIGHT channels of a
// bf16 NHWC map from its four source pixels, result rounded to bf16. ONE definition of the arithmetic for every kernel that upsamples
// the bf16 output of the last fusion projection in front of the head's first conv (fusion_model.py:182 -> head_model.py:74-76): the
// halo-staged conv kernel that interpolates its input patch in LDS (conv3h.hip, big launches) and the stand-alone upsample kernel
// (elementwise.hip, small launches) must produce the same bits - one image's result may not depend on the batch it is part of.
//   out = (1 - ly) * ((1 - lx) * v00 + lx * v01) + ly * ((1 - lx) * v10 + lx * v11)      in fp32, no contraction
#pragma once

typedef __attribute__((ext_vector_type(4))) unsigned mdpt_u32x4;

__device__ __forceinline__ mdpt_u32x4 mdpt_up_bf16x8(mdpt_u32x4 v00, mdpt_u32x4 v01, mdpt_u32x4 v10, mdpt_u32x4 v11, float lx, float ly) {
#pragma clang fp contract(off)
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const float wx0 = 1.0f - lx, wy0 = 1.0f - ly;
    mdpt_u32x4 out;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned d00 = v00[w], d01 = v01[w], d10 = v10[w], d11 = v11[w];  // (scalar copies: bit_cast on a vector ELEMENT expression reads element 0)
        const f32x2_t a00 = op2_to_f32(d00);
        const f32x2_t a01 = op2_to_f32(d01);
        const f32x2_t a10 = op2_to_f32(d10);
        const f32x2_t a11 = op2_to_f32(d11);
        const f32x2_t r = wy0 * (wx0 * a00 + lx * a01) + ly * (wx0 * a10 + lx * a11);
        out[w] = __builtin_bit_cast(unsigned, to_op2(r));
    }
    return out;
}

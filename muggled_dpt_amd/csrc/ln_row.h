// LayerNorm of ONE row by ONE wave (eps 1e-6; reference LayerNormEPS6, components/misc_helpers.py:190-210): the row sits in registers
// (NV float4 per lane, F <= 256 NV), two-pass mean / centred variance in fp32, output as bf16 hi (+lo) planes and / or fp32.
// One routine for every kernel that normalises rows (today the standalone kernel in elementwise.hip; round 2 also tried it behind the
// residual GEMM - the last-arriving workgroup of a 256-row block normalising those rows out of L2 - which was bit-identical and 2.2x
// slower per GEMM launch). Whoever calls it must produce the SAME BITS, so every multiply-add is an explicit fma /
// separate operation and contraction is off inside this function, whatever the translation unit's setting is.
#pragma once
#include "mdpt_kernels.h"
#include "f8_cross.h"

typedef __attribute__((ext_vector_type(4))) float ln_f32x4;

__device__ __forceinline__ float ln_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// `load(c)` returns the four row values at columns c .. c+3 (a plain row read, or the row plus pending partial sums: layernorm_addp_kernel).
// ln_row_values: the normalised row y = (x - mean) * rstd * gamma + beta, in registers - the ONE definition of the row arithmetic; ln_row_from
// stores it (operand planes and / or fp32); a round-5 experiment also summed its rounded values column-wise inside the LayerNorm launch (not kept).
template <int NV, class Load>
__device__ __forceinline__ void ln_row_values(Load load, const float* __restrict__ gamma, const float* __restrict__ beta, int F, int lane,
                                              ln_f32x4 (&y)[NV]) {
#pragma clang fp contract(off)
    ln_f32x4 v[NV];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < F) {
            v[i] = load(c, i);  // (c = the column, i = the lane's vector index: a caller that holds the row in registers indexes by i)
            s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
    }
    const float mean = ln_wave_sum(s) / (float)F;
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < F) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                ss = __builtin_fmaf(d, d, ss);
            }
        }
    }
    const float rstd = rsqrtf(ln_wave_sum(ss) / (float)F + 1e-6f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < F) {
            const ln_f32x4 g = *(const ln_f32x4*)(gamma + c), bt = *(const ln_f32x4*)(beta + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[i][e] = __builtin_fmaf((v[i][e] - mean) * rstd, g[e], bt[e]);
        }
    }
}

template <int NV, class Load>
__device__ __forceinline__ void ln_row_from(Load load, const float* __restrict__ gamma, const float* __restrict__ beta,
                                            op_t* out_hi, op_t* out_lo, float* out_f32, size_t out_off, int F, int lane, size_t out_f8 = 0, int out_a8 = 0) {
#pragma clang fp contract(off)
    ln_f32x4 yy[NV];
    ln_row_values<NV>([&](int c, int) { return load(c); }, gamma, beta, F, lane, yy);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < F) {
            const ln_f32x4 y = yy[i];
            if (out_hi) {
                opx4 h;
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = to_op(y[e]);
                *(opx4*)(out_hi + out_off + c) = h;
                if (out_lo) lo_store4(out_lo, out_off + c, out_f8, out_a8, y, h);  // 16-bit residue plane, or the fp8 form of an F8 consumer (f8_cross.h)
            }
            if (out_f32) *(ln_f32x4*)(out_f32 + out_off + c) = y;
        }
    }
}

template <int NV>
__device__ __forceinline__ void ln_row(const float* __restrict__ xr, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       op_t* out_hi, op_t* out_lo, float* out_f32, size_t out_off, int F, int lane, size_t out_f8 = 0, int out_a8 = 0) {
    ln_row_from<NV>([&](int c) { return *(const ln_f32x4*)(xr + c); }, gamma, beta, out_hi, out_lo, out_f32, out_off, F, lane, out_f8, out_a8);
}

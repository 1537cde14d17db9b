// Do the two block-scaled MFMA shapes give the same BITS for the same dot products? The 8-phase kernels / conv3h use 16x16x128 (one
// instruction per 128-element fp8 K tile), the lockstep GEMM tiles 32x32x64 (two per K tile, accumulating): the tile rule depends on the
// batch, so batch-independent bits of the fp8 cross-term passes need  D16[r][c] == D32[r][c]  for the same operand rows, whatever the hardware's
// internal summation is. Random e5m2 x e4m3 operands with scales, accumulators preloaded with a large "main term" (so the cross term's
// rounding lands on fp32 bits the way it does in the kernels), many trials; also 16x16x128 with its K halves swapped (order sensitivity).
//   hipcc --offload-arch=gfx950 -O2 -o f8_shape_equiv_probe f8_shape_equiv_probe.hip && ./f8_shape_equiv_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// A: 32 rows x 128 bytes (e5m2), B: 32 cols x 128 bytes (e4m3), c0: 32x32 initial accumulators, sa: scale byte, sb[32]: per-column scale bytes
// out16: rows 0-15 x cols 0-15 through one 16x16x128; out32: 32x32 through two 32x32x64; out16s: 16x16x128 with 64-byte K halves swapped
__global__ void both(const uint8_t* A, const uint8_t* B, const float* c0, const int* sb, float* out16, float* out32, float* out16s) {
    const int l = threadIdx.x;
    {   // 16x16x128: lane l = row / col l & 15, lane group g = l >> 4 holds K bytes [32 g, 32 g + 32)
        i32x8 a, b, as, bs;
        memcpy(&a, A + (l & 15) * 128 + 32 * (l >> 4), 32);
        memcpy(&b, B + (l & 15) * 128 + 32 * (l >> 4), 32);
        memcpy(&as, A + (l & 15) * 128 + 32 * ((l >> 4) ^ 2), 32);
        memcpy(&bs, B + (l & 15) * 128 + 32 * ((l >> 4) ^ 2), 32);
        f32x4 acc, acc2;
        for (int r = 0; r < 4; ++r) acc[r] = acc2[r] = c0[(4 * (l >> 4) + r) * 32 + (l & 15)];
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 1, 0, 0, 111, 0, sb[l & 15]);
        acc2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(as, bs, acc2, 1, 0, 0, 111, 0, sb[l & 15]);
        for (int r = 0; r < 4; ++r) { out16[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r]; out16s[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc2[r]; }
    }
    {   // 32x32x64 twice: lane l = row / col l & 31, half h = l >> 5 holds K bytes [64 s + 32 h, +32) of step s
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = c0[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
        for (int s = 0; s < 2; ++s) {
            i32x8 a, b;
            memcpy(&a, A + (l & 31) * 128 + 64 * s + 32 * (l >> 5), 32);
            memcpy(&b, B + (l & 31) * 128 + 64 * s + 32 * (l >> 5), 32);
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 1, 0, 0, 111, 0, sb[l & 31]);
        }
        for (int r = 0; r < 16; ++r) out32[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
    }
}

int main() {
    uint8_t *dA, *dB; float *dc, *d16, *d32, *d16s; int* dsb;
    CK(hipMalloc(&dA, 32 * 128)); CK(hipMalloc(&dB, 32 * 128)); CK(hipMalloc(&dc, 32 * 32 * 4)); CK(hipMalloc(&dsb, 32 * 4));
    CK(hipMalloc(&d16, 16 * 16 * 4)); CK(hipMalloc(&d32, 32 * 32 * 4)); CK(hipMalloc(&d16s, 16 * 16 * 4));
    std::vector<uint8_t> A(32 * 128), B(32 * 128);
    std::vector<float> c(32 * 32), o16(256), o32(1024), o16s(256);
    std::vector<int> sb(32);
    srand(11);
    long diff = 0, diff_sw = 0, total = 0;
    for (int trial = 0; trial < 400; ++trial) {
        for (auto& v : A) { v = (uint8_t)(rand() & 0xFF); if (((v >> 2) & 31) == 31) v &= 0xEF; }          // no inf / NaN codes
        for (auto& v : B) { v = (uint8_t)(rand() & 0xFF); if ((v & 0x7F) == 0x7F) v &= 0xFE; }
        for (auto& v : c) v = (trial & 1 ? 1.0f : 1024.0f) * ((float)(rand() % 2001 - 1000) / 1000.0f);     // the "main term" the cross term lands on
        for (auto& v : sb) v = 100 + rand() % 20;
        CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 32 * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(both, dim3(1), dim3(64), 0, 0, dA, dB, dc, dsb, d16, d32, d16s);
        CK(hipMemcpy(o16.data(), d16, 256 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o32.data(), d32, 1024 * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(o16s.data(), d16s, 256 * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < 16; ++r)
            for (int cc = 0; cc < 16; ++cc) {
                ++total;
                if (memcmp(&o16[r * 16 + cc], &o32[r * 32 + cc], 4)) ++diff;
                if (memcmp(&o16[r * 16 + cc], &o16s[r * 16 + cc], 4)) ++diff_sw;
            }
    }
    printf("16x16x128 vs two accumulating 32x32x64 on the same rows: %ld of %ld results differ in bits\n", diff, total);
    printf("16x16x128 vs the same with its 64-element K halves swapped: %ld of %ld results differ in bits\n", diff_sw, total);
    return 0;
}

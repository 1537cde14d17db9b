// Probe for the fp8 cross-term pass (round 6, csrc/f8_cross.h): what the gfx950 conversion and block-scaled MFMA instructions do with the operand
// formats the kernels use - E5M2 activations (v_cvt_pk_bf8_f32), E4M3 weights, one E8M0 scale register per operand with a byte select.
//  (1) v_cvt_pk_bf8_f32 / v_cvt_pk_fp8_f32: rounding (nearest even?), overflow (saturate or inf / NaN?), NaN, subnormals;
//  (2) v_mfma_scale_f32_16x16x128_f8f6f4 with cbsz = 1 (first operand bf8), blgp = 0 (second operand fp8): which operand is which;
//  (3) op_sel: which byte of the scale register an op_sel value picks (four different bytes in the register);
//  (4) the 32x32x64 form with the same formats.
//   hipcc --offload-arch=gfx950 -O2 -o f8_cross_probe f8_cross_probe.hip && ./f8_cross_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void cvt(const float* in, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i < n) {
        out[2 * i] = (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(in[i], 0.0f, 0, false) & 0xFF;
        out[2 * i + 1] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(in[i], 0.0f, 0, false) & 0xFF;
    }
}

template <int OPA, int OPB>
__global__ void mfma16(const uint8_t* a, const uint8_t* b, const int* sa, const int* sb, float* d) {
    const int lane = threadIdx.x;
    i32x8 av, bv;
    memcpy(&av, a + lane * 32, 32);
    memcpy(&bv, b + lane * 32, 32);
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 1, 0, OPA, sa[lane], OPB, sb[lane]);  // cbsz = 1: first operand bf8; blgp = 0: second fp8
    for (int r = 0; r < 4; ++r) d[lane * 4 + r] = acc[r];
}

__global__ void mfma32(const uint8_t* a, const uint8_t* b, const int* sa, const int* sb, float* d) {
    const int lane = threadIdx.x;
    i32x8 av, bv;
    memcpy(&av, a + lane * 32, 32);
    memcpy(&bv, b + lane * 32, 32);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 1, 0, 0, sa[lane], 0, sb[lane]);
    for (int r = 0; r < 16; ++r) d[lane * 16 + r] = acc[r];
}

static float dec_e5m2(uint8_t v) {
    const int s = v >> 7, e = (v >> 2) & 31, m = v & 3;
    float r;
    if (e == 31) r = m ? NAN : INFINITY;
    else if (e == 0) r = ldexpf((float)m, -16);
    else r = ldexpf(1.0f + m / 4.0f, e - 15);
    return s ? -r : r;
}
static float dec_e4m3(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r;
    if (e == 15 && m == 7) r = NAN;
    else if (e == 0) r = ldexpf((float)m, -9);
    else r = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -r : r;
}

int main() {
    // ---- (1) conversions
    std::vector<float> in = {0.0f, 1.0f, 1.125f, 1.375f, 1.625f, 1.875f, 1.0625f, 1.1875f, 3.0e-5f, 1.5e-5f, 7.0e-6f, 57344.0f, 60000.0f, 61440.0f, 65504.0f, 1.0e6f, INFINITY, -INFINITY, NAN,
                             448.0f, 464.0f, 480.0f, 500.0f, 0.001953125f, 0.0009765625f, 0.0029296875f, -2.5f, 1.0e-9f};
    float* din; unsigned* dout;
    CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dout, in.size() * 8));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, din, dout, (int)in.size());
    std::vector<unsigned> out(in.size() * 2);
    CK(hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost));
    printf("value            bf8 (e5m2)            fp8 (e4m3)\n");
    for (size_t i = 0; i < in.size(); ++i)
        printf("%-14g   0x%02x = %-12g   0x%02x = %-12g\n", in[i], out[2 * i], dec_e5m2((uint8_t)out[2 * i]), out[2 * i + 1], dec_e4m3((uint8_t)out[2 * i + 1]));

    // ---- (2) operand formats + (3) op_sel
    srand(5);
    std::vector<uint8_t> a(64 * 32), b(64 * 32);
    std::vector<float> A(16 * 128), B(16 * 128);  // A[row][k] decoded as e5m2, B[col][k] decoded as e4m3; lane l byte j <-> k = 32 (l >> 4) + j
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            uint8_t va = (uint8_t)(rand() & 0xFF), vb = (uint8_t)(rand() & 0xFF);
            if (((va >> 2) & 31) >= 20) va &= 0xBF;               // keep e5m2 values small and finite
            if (((vb >> 3) & 15) >= 12) vb &= 0xBF;               // keep e4m3 values small, never the NaN code
            a[l * 32 + j] = va; b[l * 32 + j] = vb;
            A[(l & 15) * 128 + 32 * (l >> 4) + j] = dec_e5m2(va);
            B[(l & 15) * 128 + 32 * (l >> 4) + j] = dec_e4m3(vb);
        }
    uint8_t *da, *db; int *dsa, *dsb; float* dd;
    CK(hipMalloc(&da, a.size())); CK(hipMalloc(&db, b.size())); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 64 * 16 * 4));
    CK(hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice));
    std::vector<int> sa(64), sb(64);
    std::vector<float> d(64 * 16);
    auto ref = [&](int row, int col, const std::vector<float>& X, const std::vector<float>& Y, double scale) {
        double s = 0;
        for (int k = 0; k < 128; ++k) s += (double)X[row * 128 + k] * Y[col * 128 + k];
        return s * scale;
    };
    auto check16 = [&](const char* what, double scale, bool first_is_e5m2) {
        double worst = 0, mag = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (l >> 4) + r, col = l & 15;
                // decode hypothesis: first operand (rows) read as e5m2 / e4m3
                double s = 0;
                for (int k = 0; k < 128; ++k) {
                    const uint8_t ba = a[(row) * 0 + ((k >> 5) * 16 + row) * 32 + (k & 31)], bb = b[((k >> 5) * 16 + col) * 32 + (k & 31)];
                    s += (double)(first_is_e5m2 ? dec_e5m2(ba) : dec_e4m3(ba)) * (first_is_e5m2 ? dec_e4m3(bb) : dec_e5m2(bb));
                }
                worst = fmax(worst, fabs(s * scale - d[l * 4 + r])); mag = fmax(mag, fabs(s * scale));
            }
        printf("%s: max |D - ref| = %g (max |ref| %g)\n", what, worst, mag);
    };
    for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = 127; }
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((mfma16<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
    CK(hipMemcpy(d.data(), dd, 64 * 4 * 4, hipMemcpyDeviceToHost));
    check16("cbsz=1 blgp=0, hypothesis first operand e5m2 / second e4m3", 1.0, true);
    check16("cbsz=1 blgp=0, hypothesis first operand e4m3 / second e5m2", 1.0, false);
    // op_sel: scale register = bytes {127, 128, 129, 130} (x1, x2, x4, x8), the same in every lane; op_sel value v on the A side
    for (int l = 0; l < 64; ++l) sa[l] = 127 | (128 << 8) | (129 << 16) | (130 << 24);
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((mfma16<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd); CK(hipMemcpy(d.data(), dd, 64 * 4 * 4, hipMemcpyDeviceToHost)); check16("op_sel_a = 0 vs x1", 1.0, true);
    hipLaunchKernelGGL((mfma16<1, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd); CK(hipMemcpy(d.data(), dd, 64 * 4 * 4, hipMemcpyDeviceToHost)); check16("op_sel_a = 1 vs x2", 2.0, true); check16("op_sel_a = 1 vs x4", 4.0, true);
    hipLaunchKernelGGL((mfma16<2, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd); CK(hipMemcpy(d.data(), dd, 64 * 4 * 4, hipMemcpyDeviceToHost)); check16("op_sel_a = 2 vs x2", 2.0, true); check16("op_sel_a = 2 vs x4", 4.0, true);
    hipLaunchKernelGGL((mfma16<3, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd); CK(hipMemcpy(d.data(), dd, 64 * 4 * 4, hipMemcpyDeviceToHost)); check16("op_sel_a = 3 vs x8", 8.0, true);
    for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = 127 | (128 << 8) | (129 << 16) | (130 << 24); }
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((mfma16<0, 1>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd); CK(hipMemcpy(d.data(), dd, 64 * 4 * 4, hipMemcpyDeviceToHost)); check16("op_sel_b = 1 vs x2", 2.0, true); check16("op_sel_b = 1 vs x4", 4.0, true);
    hipLaunchKernelGGL((mfma16<0, 3>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd); CK(hipMemcpy(d.data(), dd, 64 * 4 * 4, hipMemcpyDeviceToHost)); check16("op_sel_b = 3 vs x8", 8.0, true);
    // a scale exponent far from 127 (the constant 2^-16 of the activation residue planes): byte 111
    for (int l = 0; l < 64; ++l) { sa[l] = 111; sb[l] = 127; }
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((mfma16<0, 0>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd); CK(hipMemcpy(d.data(), dd, 64 * 4 * 4, hipMemcpyDeviceToHost)); check16("scale_a = 111 vs x 2^-16", ldexp(1.0, -16), true);

    // ---- (4) 32x32x64, same formats: lane l byte j <-> row / column l & 31, k = 32 (l >> 5) + j
    for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = 127; }
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma32, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
    CK(hipMemcpy(d.data(), dd, 64 * 16 * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            double s = 0;
            for (int k = 0; k < 64; ++k) s += (double)dec_e5m2(a[((k >> 5) * 32 + row) * 32 + (k & 31)]) * dec_e4m3(b[((k >> 5) * 32 + col) * 32 + (k & 31)]);
            worst = fmax(worst, fabs(s - d[l * 16 + r]));
        }
    printf("32x32x64 cbsz=1 blgp=0 (first e5m2, second e4m3): max |D - ref| = %g\n", worst);
    return 0;
}

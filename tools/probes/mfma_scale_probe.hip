// Probe of gfx950's block-scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4) for the round-6 plan "cross terms of the split products on MX operands"
// (DESIGN.md section 9). What it established on an MI355X (profiles/r05_mfma_scale_probe.txt):
//  (1) operands: lane l holds 32 bytes (fp8) of row / column l & 31; byte j of lane l pairs with byte j of lane l on the other operand (any
//      K order that is the same for A and B contracts correctly); C/D layout as every 32x32 MFMA;
//  (2) the scale operand is NOT per lane: the E8M0 byte of lane r (< 32) scales bytes 0-15 of lanes r AND r + 32, the byte of lane r + 32
//      scales bytes 16-31 of both - i.e. the hardware's K order is k = 32 (j >> 4) + 16 (l >> 5) + (j & 15) and an MX block of 32 consecutive
//      k sits in TWO lanes, 16 bytes each;
//  (2b) the 16x16x128 form (the accumulator shape of the 8-phase kernels): lane l = row / column l & 15, lane group g = l >> 4 holds 32 of the 128 K
//      bytes, C/D as every 16x16 MFMA; scales: lane r scales bytes 0-15 of lane groups 0 and 1, lane r + 16 bytes 0-15 of groups 2 and 3,
//      lane r + 32 bytes 16-31 of groups 0 and 1, lane r + 48 bytes 16-31 of groups 2 and 3;
//  (3) issue rate against v_mfma_f32_32x32x16_f16: fp8 x fp8 2.2x, fp6 x fp6 and fp4 x fp4 4.1-4.2x (K = 64 per instruction against 16).
//   hipcc --offload-arch=gfx950 -O2 -o mfma_scale_probe mfma_scale_probe.hip && ./mfma_scale_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// one wave: a, b = 64 lanes x 32 bytes (fp8 e4m3), sa / sb = 64 E8M0 bytes; d = 64 lanes x 16 floats
__global__ void one_mfma(const uint8_t* a, const uint8_t* b, const uint8_t* sa, const uint8_t* sb, float* d) {
    const int lane = threadIdx.x;
    i32x8 av, bv;
    memcpy(&av, a + lane * 32, 32);
    memcpy(&bv, b + lane * 32, 32);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 0, 0, 0, (int)sa[lane], 0, (int)sb[lane]);
    for (int r = 0; r < 16; ++r) d[lane * 16 + r] = acc[r];
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
// one wave of the 16x16x128 form: a, b = 64 lanes x 32 bytes (fp8), d = 64 lanes x 4 floats
__global__ void one_mfma16(const uint8_t* a, const uint8_t* b, const uint8_t* sa, const uint8_t* sb, float* d) {
    const int lane = threadIdx.x;
    i32x8 av, bv;
    memcpy(&av, a + lane * 32, 32);
    memcpy(&bv, b + lane * 32, 32);
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, (int)sa[lane], 0, (int)sb[lane]);
    for (int r = 0; r < 4; ++r) d[lane * 4 + r] = acc[r];
}

template <int FMT>
__global__ void rate_scaled(float* out, int iters) {
    i32x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = 0x38383838 + threadIdx.x * 0x01010101 * (i & 1); bv[i] = 0x3c3c3c3c ^ (i * 0x00010001); }
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = acc3[r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc0, FMT, FMT, 0, 127, 0, 127);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc1, FMT, FMT, 0, 127, 0, 127);
        acc2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc2, FMT, FMT, 0, 127, 0, 127);
        acc3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc3, FMT, FMT, 0, 127, 0, 127);
    }
    float s = 0.0f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void rate_f16(float* out, int iters) {
    f16x8 av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = (_Float16)(0.5f + 0.001f * threadIdx.x); bv[i] = (_Float16)(0.25f * (i + 1)); }
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = acc3[r] = 0.0f;
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc3, 0, 0, 0);
    }
    float s = 0.0f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// fp8 e4m3 (OCP) encode of small values that are exactly representable
static uint8_t enc_e4m3(float v) {
    if (v == 0.0f) return 0;
    const uint8_t sign = v < 0 ? 0x80 : 0;
    v = fabsf(v);
    int e;
    const float m = frexpf(v, &e);  // v = m 2^e, m in [0.5, 1)
    int E = e - 1 + 7;              // biased exponent of 1.xxx 2^(e-1)
    int M = (int)lrintf((m * 2.0f - 1.0f) * 8.0f);
    if (E < 1) { M = (int)lrintf(v / ldexpf(1.0f, -9)); E = 0; }  // subnormal: M 2^-9
    return sign | (uint8_t)(E << 3) | (uint8_t)(M & 7);
}

int main() {
    // ---- (1) layout + (2) scale semantics
    std::vector<float> A(32 * 64), B(64 * 32);
    srand(3);
    for (auto& v : A) v = (float)(rand() % 9 - 4);          // -4 .. 4
    for (auto& v : B) v = (float)(rand() % 7 - 3) * 0.5f;   // -1.5 .. 1.5
    std::vector<uint8_t> a(64 * 32), b(64 * 32), sa(64, 127), sb(64, 127);
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
            a[l * 32 + j] = enc_e4m3(A[(l & 31) * 64 + 32 * (l >> 5) + j]);
            b[l * 32 + j] = enc_e4m3(B[(32 * (l >> 5) + j) * 32 + (l & 31)]);
        }
    uint8_t *da, *db, *dsa, *dsb;
    float* dd;
    CK(hipMalloc(&da, a.size())); CK(hipMalloc(&db, b.size())); CK(hipMalloc(&dsa, 64)); CK(hipMalloc(&dsb, 64)); CK(hipMalloc(&dd, 64 * 16 * 4));
    CK(hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice));
    std::vector<float> d(64 * 16);
    // which (row / column, K half) does the scale byte of lane L apply to? Scale ONE lane by 4 and compare D with the unit-scale D: the rows
    // (A side) / columns (B side) that moved, and whether they moved by 3 x the partial sum over k < 32 or k >= 32
    auto run = [&]() {
        CK(hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
        CK(hipMemcpy(d.data(), dd, d.size() * 4, hipMemcpyDeviceToHost));
    };
    run();
    const std::vector<float> d0 = d;
    double worst0 = 0;
    std::vector<double> part(2 * 32 * 32), quart(4 * 32 * 32);  // [lane half][row][col]; [lane half * 2 + byte half][row][col]
    for (int row = 0; row < 32; ++row)
        for (int col = 0; col < 32; ++col)
            for (int k = 0; k < 64; ++k) {
                part[((k >> 5) * 32 + row) * 32 + col] += (double)A[row * 64 + k] * B[k * 32 + col];
                quart[((k >> 4) * 32 + row) * 32 + col] += (double)A[row * 64 + k] * B[k * 32 + col];
            }
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
            worst0 = fmax(worst0, fabs(part[row * 32 + col] + part[(32 + row) * 32 + col] - d0[l * 16 + r]));
        }
    printf("unit scales: max |D - ref| = %g (operand layout: lane l byte j <-> k = 32 (l >> 5) + j for A rows / B columns l & 31; C/D as every 32x32 MFMA)\n", worst0);
    for (int side = 0; side < 2; ++side)
        for (int L : {0, 5, 31, 32, 37, 63}) {
            std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
            (side ? sb : sa)[L] = 129;
            run();
            int moved = 0, first = -1, last = -1, as_half0 = 0, as_half1 = 0, other = 0;
            int subset_hits[16] = {0};
            for (int idx = 0; idx < 32; ++idx) {  // idx = row (A side) or column (B side)
                bool any = false;
                for (int o = 0; o < 32; ++o) {
                    const int row = side ? o : idx, col = side ? idx : o;
                    // find D[row][col]
                    const int l = col + 32 * ((row >> 2) & 1), r = (row & 3) + 4 * (row >> 3);
                    const double diff = d[l * 16 + r] - d0[l * 16 + r];
                    if (diff != 0) {
                        any = true;
                        if (fabs(diff - 3 * part[row * 32 + col]) < 1e-3) ++as_half0;
                        else if (fabs(diff - 3 * part[(32 + row) * 32 + col]) < 1e-3) ++as_half1;
                        else ++other;
                        for (int sub = 1; sub < 16; ++sub) {  // which quarters (lane half, byte half) of the operand were scaled?
                            double e = 0;
                            for (int q = 0; q < 4; ++q) if (sub >> q & 1) e += 3 * quart[(q * 32 + row) * 32 + col];
                            if (fabs(diff - e) < 1e-3) ++subset_hits[sub];
                        }
                    }
                }
                if (any) { ++moved; if (first < 0) first = idx; last = idx; }
            }
            int best = 1;
            for (int sub = 1; sub < 16; ++sub) if (subset_hits[sub] > subset_hits[best]) best = sub;
            printf("%c scale x4 on lane %2d: %s %d moved; best-fitting scaled part: quarters {%s%s%s%s} of the operand (lane half, byte half) - explains %d of %d moved elements\n",
                   side ? 'B' : 'A', L, side ? "column" : "row", first, best & 1 ? " (0,0)" : "", best & 2 ? " (0,1)" : "", best & 4 ? " (1,0)" : "", best & 8 ? " (1,1)" : "",
                   subset_hits[best], as_half0 + as_half1 + other);
            (void)moved; (void)last;
        }
    {   // the discovered semantics, checked exactly with random scales 2^-2 .. 2^2 on every lane
        for (int l = 0; l < 64; ++l) { sa[l] = (uint8_t)(125 + rand() % 5); sb[l] = (uint8_t)(125 + rand() % 5); }
        run();
        double worst = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
                double ref = 0;
                for (int lh = 0; lh < 2; ++lh)
                    for (int j = 0; j < 32; ++j) {  // operand byte j of lane half lh: scale from lane (row | col) + 32 * (j >> 4)
                        const int k = 32 * lh + j;  // (the probe's own K numbering: byte j of lane half lh)
                        ref += (double)A[row * 64 + k] * ldexp(1.0, sa[row + 32 * (j >> 4)] - 127) * B[k * 32 + col] * ldexp(1.0, sb[col + 32 * (j >> 4)] - 127);
                    }
                worst = fmax(worst, fabs(ref - d[l * 16 + r]));
            }
        printf("random scales, semantics 'lane r scales bytes 0-15 of lanes r and r+32, lane r+32 bytes 16-31': max |D - ref| = %g (%s)\n", worst, worst == 0 ? "EXACT" : "MISMATCH");
    }
    {   // ---- the 16x16x128 form (the accumulator shape of the 8-phase kernels): rows / columns l & 15, four lane groups of 32 K bytes each
        std::vector<float> A2(16 * 128), B2(128 * 16);
        for (auto& v : A2) v = (float)(rand() % 9 - 4);
        for (auto& v : B2) v = (float)(rand() % 7 - 3) * 0.5f;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 32; ++j) {
                a[l * 32 + j] = enc_e4m3(A2[(l & 15) * 128 + 32 * (l >> 4) + j]);
                b[l * 32 + j] = enc_e4m3(B2[(32 * (l >> 4) + j) * 16 + (l & 15)]);
            }
        CK(hipMemcpy(da, a.data(), a.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size(), hipMemcpyHostToDevice));
        std::vector<float> e(64 * 4), e0;
        auto run16 = [&]() {
            CK(hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(one_mfma16, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
            CK(hipMemcpy(e.data(), dd, e.size() * 4, hipMemcpyDeviceToHost));
        };
        std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
        run16();
        e0 = e;
        std::vector<double> eighth(8 * 16 * 16);  // [lane group * 2 + byte half][row][col]
        for (int row = 0; row < 16; ++row)
            for (int col = 0; col < 16; ++col)
                for (int k = 0; k < 128; ++k) eighth[((k >> 4) * 16 + row) * 16 + col] += (double)A2[row * 128 + k] * B2[k * 16 + col];
        double w16 = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * (l >> 4) + r, col = l & 15;
                double ref = 0;
                for (int q = 0; q < 8; ++q) ref += eighth[(q * 16 + row) * 16 + col];
                w16 = fmax(w16, fabs(ref - e0[l * 4 + r]));
            }
        printf("16x16x128, unit scales: max |D - ref| = %g (lane l byte j <-> row / column l & 15, k = 32 (l >> 4) + j; D[row = 4 (l >> 4) + r][col = l & 15])\n", w16);
        for (int side = 0; side < 2; ++side)
            for (int L : {3, 19, 35, 51}) {
                std::fill(sa.begin(), sa.end(), 127); std::fill(sb.begin(), sb.end(), 127);
                (side ? sb : sa)[L] = 129;
                run16();
                int hits[256] = {0}, moved = 0, idx_moved = -1;
                for (int row = 0; row < 16; ++row)
                    for (int col = 0; col < 16; ++col) {
                        const int l = col + 16 * (row >> 2), r = row & 3;
                        const double diff = e[l * 4 + r] - e0[l * 4 + r];
                        if (diff == 0) continue;
                        ++moved; idx_moved = side ? col : row;
                        for (int sub = 1; sub < 256; ++sub) {
                            double x = 0;
                            for (int q = 0; q < 8; ++q) if (sub >> q & 1) x += 3 * eighth[(q * 16 + row) * 16 + col];
                            if (fabs(diff - x) < 1e-3) ++hits[sub];
                        }
                    }
                int best = 1;
                for (int sub = 1; sub < 256; ++sub) if (hits[sub] > hits[best]) best = sub;
                printf("16x16x128 %c scale x4 on lane %2d: %s %d moved; scaled part = (lane group, byte half) {", side ? 'B' : 'A', L, side ? "column" : "row", idx_moved);
                for (int q = 0; q < 8; ++q) if (best >> q & 1) printf(" (%d,%d)", q >> 1, q & 1);
                printf(" } - explains %d of %d moved elements\n", hits[best], moved);
            }
    }
    // ---- (3) issue rate
    float* dout;
    const int blocks = 256 * 8, threads = 256, iters = 4000;
    CK(hipMalloc(&dout, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char* name, auto launch, double flop_per_instr) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        const double instr = (double)blocks * (threads / 64) * iters * 4;
        printf("%-34s %8.3f ms  %8.1f TFLOP/s\n", name, ms, instr * flop_per_instr / (ms * 1e-3) / 1e12);
    };
    time_it("v_mfma_f32_32x32x16_f16", [&] { hipLaunchKernelGGL(rate_f16, dim3(blocks), dim3(threads), 0, 0, dout, iters); }, 2.0 * 32 * 32 * 16);
    time_it("scaled 32x32x64 fp8 x fp8", [&] { hipLaunchKernelGGL((rate_scaled<0>), dim3(blocks), dim3(threads), 0, 0, dout, iters); }, 2.0 * 32 * 32 * 64);
    time_it("scaled 32x32x64 fp6 x fp6", [&] { hipLaunchKernelGGL((rate_scaled<2>), dim3(blocks), dim3(threads), 0, 0, dout, iters); }, 2.0 * 32 * 32 * 64);
    time_it("scaled 32x32x64 fp4 x fp4", [&] { hipLaunchKernelGGL((rate_scaled<4>), dim3(blocks), dim3(threads), 0, 0, dout, iters); }, 2.0 * 32 * 32 * 64);
    return 0;
}

// Probe: what does the MI355X sustain on bf16 MFMA once the GEMM's other ingredients are added one at a time, and does a wave tile with
// fewer LDS fragment reads per MFMA (128x128 per wave, 4 waves of 512 registers) sustain more than the 8-wave 128x64 form? All variants run
// random data on all 256 CUs for ~0.3 s each (the chip settles at its power limit: MI355X_MICROARCH.md "DVFS give-back"), so the numbers are
// what the power / issue budget allows, not what the instruction timings add up to.
//
//   v0  8 waves, 64 MFMA 16x16x32 per "K tile", operands constant in registers                       (MFMA pipe + register file only)
//   v1  v0 + the 24 ds_read_b128 per wave and K tile of a 128x64 wave tile (A 16, B 8), double-buffered fragments
//   v2  v1 + 8 LDS-DMA instructions (1 KiB each) per wave and K tile from an L2-resident panel         (= gemm8's per-K-tile traffic)
//   v3  4 waves, 128x128 per wave: 128 MFMA per K tile and 32 ds_read_b128 (A 16, B 16)                (2/3 of v1's reads per MFMA)
//   v4  v3 + 16 LDS-DMA instructions per wave and K tile                                               (same bytes per CU as v2)
//   v5 / v6  v1 / v3 with 32x32x16 MFMAs (same reads; half the MFMA instructions and operand-register reads per flop)
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_power.hip -o tools/probes/_bin/mfma_power && tools/probes/_bin/mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

// -DPROBE_F16: the same loops on fp16 operands (v_mfma_*_f16; round 5: is the fp16 build's 3-4 % slower GEMM loop the MFMA pipe's power, or the kernels?)
#ifdef PROBE_F16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;
#define MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define MFMA32 __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
#define MFMA16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define MFMA32 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define OPAQUE(x) asm volatile("" : "+v"(x))

__device__ __forceinline__ void dma1k(const char* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// RB x CB blocks of 16x16 per wave (8x4 = 128x64, 8x8 = 128x128), two k-steps of 32 per K tile
template <int RB, int CB, bool LDS, int NDMA, int NT>
__global__ __launch_bounds__(NT, 1) void k16(const char* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // fill 128 KiB of LDS with the random panel (fragment reads below take whatever is there)
    for (int i = tid * 16; i < 131072; i += NT * 16) *(bf16x8*)(smem + i) = *(const bf16x8*)(src + i);
    __syncthreads();
    f32x4 acc[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    bf16x8 a[2][RB], b[2][CB];
    int off = (wave * 4096 + lane * 16) & 32767;
    auto load = [&](int buf, int base) {
#pragma unroll
        for (int i = 0; i < RB; ++i) a[buf][i] = *(const bf16x8*)(smem + base + off + i * 2048);
#pragma unroll
        for (int j = 0; j < CB; ++j) b[buf][j] = *(const bf16x8*)(smem + 65536 + base + off + j * 2048);
    };
    const char* gp = src + (size_t)blockIdx.x * 65536 + lane * 16;
    if (LDS) load(0, 0);
    else {
#pragma unroll
        for (int i = 0; i < RB; ++i) a[0][i] = a[1][i] = *(const bf16x8*)(src + lane * 16 + i * 1024);
#pragma unroll
        for (int j = 0; j < CB; ++j) b[0][j] = b[1][j] = *(const bf16x8*)(src + 32768 + lane * 16 + j * 1024);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (LDS) {
                OPAQUE(off);
                load(ks ^ 1, ks ? 0 : 16384);
            }
            if (NDMA) {
#pragma unroll
                for (int d = 0; d < NDMA / 2; ++d) dma1k(gp + ((it * NDMA + ks * (NDMA / 2) + d) & 63) * 1024, smem + 32768 + wave * 1024);
            }
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j) acc[i][j] = MFMA16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
        }
    }
    f32x4 s = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j) s += acc[i][j];
    out[(size_t)blockIdx.x * NT + tid] = s[0] + s[1] + s[2] + s[3];
}

// the same wave tiles out of 32x32x16 MFMAs: RB x CB blocks of 32x32, four k-steps of 16 per K tile, one b128 read feeds TWO k-steps
template <int RB, int CB, int NT>
__global__ __launch_bounds__(NT, 1) void k32(const char* src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid * 16; i < 131072; i += NT * 16) *(bf16x8*)(smem + i) = *(const bf16x8*)(src + i);
    __syncthreads();
    f32x16 acc[RB][CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
    // per k-step of 16: a lane holds 8 bf16 of one row -> RB + CB b128 reads per k-step, 4 k-steps per K tile: (RB + CB) * 4 reads,
    // the same count as the 16x16x32 form of the same wave tile ((2 RB + 2 CB) * 2)
    bf16x8 a[2][RB], b[2][CB];
    int off = (wave * 4096 + lane * 16) & 32767;
    auto load = [&](int buf, int base) {
#pragma unroll
        for (int i = 0; i < RB; ++i) a[buf][i] = *(const bf16x8*)(smem + base + off + i * 2048);
#pragma unroll
        for (int j = 0; j < CB; ++j) b[buf][j] = *(const bf16x8*)(smem + 65536 + base + off + j * 2048);
    };
    load(0, 0);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            OPAQUE(off);
            load((ks & 1) ^ 1, ((ks + 1) & 3) * 8192);
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j) acc[i][j] = MFMA32(a[ks & 1][i], b[ks & 1][j], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[(size_t)blockIdx.x * NT + tid] = s;
}

template <typename K>
static void run(const char* what, K kern, int nt, double flops_per_iter_per_wg, const char* src, float* out) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int grid = 256;
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), 131072, 0, src, out, iters);
    hipDeviceSynchronize();
    float ms = 0;
    // ~0.3 s of back-to-back launches so the clock settles, the last 10 launches timed
    for (int rep = 0; rep < 2; ++rep) {
        const int launches = rep == 0 ? 30 : 10;
        hipEventRecord(e0, 0);
        for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(kern, dim3(grid), dim3(nt), 131072, 0, src, out, iters * 4);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        ms /= launches;
    }
    const double tf = flops_per_iter_per_wg * iters * 4 * grid / (ms * 1e-3) * 1e-12;
    const double mfma_cycles = flops_per_iter_per_wg / (4 * 1024.0);  // 1024 flops per cycle and SIMD at the dense bf16 rate
    printf("%-78s %7.1f TFLOP/s  (%.3f of 2.5 PF)  %6.0f ns per K tile (MFMA-bound at 2.4 GHz: %.0f)\n", what, tf, tf / 2500.0,
           ms * 1e6 / (iters * 4), mfma_cycles / 2.4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("  !! %s\n", hipGetErrorString(e));
}

int main() {
    const size_t bytes = 256 * 65536 + 131072;
    std::vector<unsigned short> h(bytes / 2);
    srand(7);
    for (auto& v : h) {
        float f = (float)rand() / RAND_MAX * 2.0f - 1.0f;
#ifdef PROBE_F16
        _Float16 hf = (_Float16)f;
        memcpy(&v, &hf, 2);
#else
        unsigned u;
        memcpy(&u, &f, 4);
        v = (unsigned short)(u >> 16);
#endif
    }
    char* src;
    float* out;
    hipMalloc(&src, bytes);
    hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
    const double f8 = 2.0 * 256 * 256 * 64;  // one 256x256x64 K tile per workgroup and iteration
    run("v0 8 waves x 128x64, MFMA 16x16x32 only (operands in registers)", k16<8, 4, false, 0, 512>, 512, f8, src, out);
    run("v1 v0 + 24 ds_read_b128 per wave and K tile", k16<8, 4, true, 0, 512>, 512, f8, src, out);
    run("v2 v1 + 8 LDS-DMA KiB per wave and K tile (gemm8's traffic)", k16<8, 4, true, 8, 512>, 512, f8, src, out);
    run("v3 4 waves x 128x128, 32 ds_read_b128 per wave and K tile", k16<8, 8, true, 0, 256>, 256, f8, src, out);
    run("v4 v3 + 16 LDS-DMA KiB per wave and K tile", k16<8, 8, true, 16, 256>, 256, f8, src, out);
    run("v5 8 waves x 128x64 out of 32x32x16 MFMAs, 24 ds_read_b128", k32<4, 2, 512>, 512, f8, src, out);
    run("v6 4 waves x 128x128 out of 32x32x16 MFMAs, 32 ds_read_b128", k32<4, 4, 256>, 256, f8, src, out);
    run("v0 again (drift check)", k16<8, 4, false, 0, 512>, 512, f8, src, out);
    return 0;
}

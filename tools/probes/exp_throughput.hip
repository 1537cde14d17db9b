// Micro-probe for the attention softmax question: is v_exp_f32 expensive enough on gfx950 that moving part of the
// exponentials onto the FMA pipe (range reduction + polynomial exp2, FlashAttention-4 style) would pay?
// Three bodies over the same register-resident data, many waves per SIMD and one wave per SIMD:
//   hw    : e = v_exp_f32(s * log2e - m)                         (1 fma + 1 transcendental per element, the kernel's current form)
//   poly3 : n = floor(x); f = x - n; p = c0 + f (c1 + f (c2 + f c3)); e = ldexp(p, n)   (fma-only, degree 3: rel. error ~1e-4, enough for bf16 P)
//   fma   : 1 fma per element (the floor: what the loop costs without any exponential)
// Prints cycles per element per wave. Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/_bin/exp_throughput tools/probes/exp_throughput.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void body(float* out, long long* cyc, int iters, float m) {
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = -0.01f * (float)((threadIdx.x * 7 + i * 13) & 255);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float x = __builtin_fmaf(v[i], 1.4426950408889634f, -m);
            if (MODE == 0) {
                v[i] = __builtin_amdgcn_exp2f(x) - 1.0f;
            } else if (MODE == 1) {
                const float n = __builtin_floorf(x);
                const float f = x - n;
                float p = __builtin_fmaf(f, 0.0790199f, 0.2240818f);
                p = __builtin_fmaf(f, p, 0.6960656f);
                p = __builtin_fmaf(f, p, 1.0f);
                v[i] = __builtin_amdgcn_ldexpf(p, (int)n) - 1.0f;
            } else {
                v[i] = x * 0.5f;
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
double run(int grid, int iters, float* dout, long long* dcyc) {
    hipLaunchKernelGGL(body<MODE>, dim3(grid), dim3(256), 0, 0, dout, dcyc, iters, 0.25f);
    CK(hipDeviceSynchronize());
    long long* h = (long long*)malloc(sizeof(long long) * grid * 4);
    CK(hipMemcpy(h, dcyc, sizeof(long long) * grid * 4, hipMemcpyDeviceToHost));
    double tot = 0;
    for (int i = 0; i < grid * 4; ++i) tot += (double)h[i];
    free(h);
    return tot / (grid * 4) / ((double)iters * 32);
}

int main() {
    float* dout; long long* dcyc;
    CK(hipMalloc(&dout, 8192 * 256 * 4)); CK(hipMalloc(&dcyc, 8192 * 4 * 8));
    const char* names[3] = {"hw v_exp_f32 + fma", "poly3 exp2 (fma only)", "fma only"};
    for (int occ = 0; occ < 2; ++occ) {
        const int grid = occ == 0 ? 256 : 2048;  // 1 wave per SIMD vs 8 waves per SIMD
        double c[3];
        run<0>(grid, 64, dout, dcyc);
        c[0] = run<0>(grid, 512, dout, dcyc); c[1] = run<1>(grid, 512, dout, dcyc); c[2] = run<2>(grid, 512, dout, dcyc);
        for (int k = 0; k < 3; ++k)
            printf("%-24s %s: %.2f cycles per element per wave (x%.2f of the hw form)\n", names[k], occ == 0 ? "1 wave/SIMD " : "8 waves/SIMD", c[k], c[k] / c[0]);
    }
    return 0;
}

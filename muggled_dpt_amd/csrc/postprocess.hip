// Depth post-processing on the GPU: the step immediately after the DPT forward in the reference's demos
// (muggled_dpt/demo_helpers/postprocess.py:22-29 scale_prediction, :63-74 normalize_01, :79-91 convert_to_uint8;
// run_3dviewer.py:576-590 24-bit packing). HBM-bound streaming kernels: fp32 in, fp32 / u8 out, no host round trip of the
// full-resolution fp32 map. min/max travel through a 2-float device buffer (no sync).

#include "mdpt_kernels.h"
#include "mdpt_prof.h"

namespace {

// order-preserving float <-> uint mapping so that atomicMin/atomicMax on unsigned work for any sign
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__global__ void minmax_init_kernel(unsigned* mm) {
    mm[0] = 0xffffffffu;  // running min (ordered domain)
    mm[1] = 0u;           // running max
}

// torch's .min() / .max() PROPAGATE NaN (normalize_01 of a map with a NaN gives an all-NaN map) while fminf / fmaxf drop it: a thread
// that saw a NaN says so, and the block then pins the running min to ordered 0 and the running max to ordered ~0, both of which
// ord2f() maps back to NaN bit patterns.
__device__ __forceinline__ void block_minmax(float lo, float hi, bool saw_nan, unsigned* mm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o));
        hi = fmaxf(hi, __shfl_xor(hi, o));
    }
    const bool wave_nan = __any(saw_nan);
    __shared__ float slo[4], shi[4];
    __shared__ int snan[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { slo[wave] = lo; shi[wave] = hi; snan[wave] = wave_nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool any_nan = wave_nan;
        for (int w = 1; w < 4; ++w) { lo = fminf(lo, slo[w]); hi = fmaxf(hi, shi[w]); any_nan |= snan[w] != 0; }
        atomicMin(mm + 0, any_nan ? 0u : f2ord(lo));
        atomicMax(mm + 1, any_nan ? 0xffffffffu : f2ord(hi));
    }
}

__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ in, size_t n, unsigned* mm) {
    float lo = INFINITY, hi = -INFINITY;
    bool saw_nan = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = in[i];
        saw_nan |= v != v;
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    block_minmax(lo, hi, saw_nan, mm);
}

__global__ void minmax_finish_kernel(const unsigned* mm, float* out) {
    out[0] = ord2f(mm[0]);
    out[1] = ord2f(mm[1]);
}

// F.interpolate(x[:, None], size=(oh, ow), mode="bilinear") (align_corners=False, no antialias): src = max(0, s*(dst+0.5)-0.5)
// Optionally folds the min/max reduction of the OUTPUT into the same pass (for convert_to_uint8(scale_prediction(x))).
__global__ __launch_bounds__(256) void scale_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int ih, int iw,
                                                             int oh, int ow, unsigned* mm) {
    const float sy = (float)ih / (float)oh, sx = (float)iw / (float)ow;
    const size_t total = (size_t)B * oh * ow;
    float lo = INFINITY, hi = -INFINITY;
    bool saw_nan = false;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int ox = (int)(idx % ow), oy = (int)((idx / ow) % oh);
        const size_t b = idx / ((size_t)ow * oh);
        const float fy = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.0f), fx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.0f);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < ih - 1), x1 = x0 + (x0 < iw - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* p = in + b * (size_t)ih * iw;
        const float v = (1.0f - ly) * ((1.0f - lx) * p[(size_t)y0 * iw + x0] + lx * p[(size_t)y0 * iw + x1]) +
                        ly * ((1.0f - lx) * p[(size_t)y1 * iw + x0] + lx * p[(size_t)y1 * iw + x1]);
        out[idx] = v;
        saw_nan |= v != v;
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    }
    if (mm) block_minmax(lo, hi, saw_nan, mm);
}

// mode 0: fp32 (x - min) / (max - min); mode 1: u8 = trunc(255 * norm) (Tensor.byte()); mode 2: BGRA u8 = 24-bit
// round-half-even(16777215 * norm) split into bytes (B = low, G = mid, R = high; alpha left 0), lossy: high byte only.
// minmax == null: the input is used as is (metric models skip normalize_01, run_3dviewer.py:577-578).
template <int MODE>
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ in, const float* __restrict__ minmax, void* out, size_t n,
                                                        int lossy) {
    const float lo = minmax ? minmax[0] : 0.0f, hi = minmax ? minmax[1] : 1.0f;
    const float range = hi - lo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = minmax ? (in[i] - lo) / range : in[i];
        if (MODE == 0) {
            ((float*)out)[i] = v;  // NaN (NaN input, or max == min) propagates like the reference's fp32 result
            continue;
        }
        // integer outputs: a NaN (max == min: 0 / 0; NaN anywhere in the map) becomes 0 - converting NaN to an integer is undefined in
        // C and implementation-defined in torch's .byte(); values are clamped to the representable range before the conversion
        v = v == v ? fminf(fmaxf(v, 0.0f), 1.0f) : 0.0f;
        if (MODE == 1) {
            ((unsigned char*)out)[i] = (unsigned char)(int)(255.0f * v);
        } else {
            const int q = (int)rintf(16777215.0f * v);
            uchar4 px;
            px.x = lossy ? 0 : (unsigned char)(q & 255);
            px.y = lossy ? 0 : (unsigned char)((q >> 8) & 255);
            px.z = (unsigned char)((q >> 16) & 255);
            px.w = 0;
            ((uchar4*)out)[i] = px;
        }
    }
}

inline int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

int mdpt_launch_post_minmax(const float* in, size_t n, float* minmax_out, unsigned* scratch2, hipStream_t stream) {
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, stream, scratch2);
    hipLaunchKernelGGL(minmax_kernel, dim3(grid_for(n)), dim3(256), 0, stream, in, n, scratch2);
    hipLaunchKernelGGL(minmax_finish_kernel, dim3(1), dim3(1), 0, stream, scratch2, minmax_out);
    return (int)hipGetLastError();
}

int mdpt_launch_post_scale(const float* in, float* out, int B, int ih, int iw, int oh, int ow, float* minmax_out, unsigned* scratch2,
                           hipStream_t stream) {
    MdptProfScope prof("scale_bilinear_kernel", 0.0, stream);
    if (minmax_out) hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, stream, scratch2);
    hipLaunchKernelGGL(scale_bilinear_kernel, dim3(grid_for((size_t)B * oh * ow)), dim3(256), 0, stream, in, out, B, ih, iw, oh, ow,
                       minmax_out ? scratch2 : nullptr);
    if (minmax_out) hipLaunchKernelGGL(minmax_finish_kernel, dim3(1), dim3(1), 0, stream, scratch2, minmax_out);
    return (int)hipGetLastError();
}

int mdpt_launch_post_normalize(const float* in, const float* minmax, void* out, size_t n, int mode, int lossy, hipStream_t stream) {
    MdptProfScope prof("normalize_kernel", 0.0, stream);
    if (mode == 0) hipLaunchKernelGGL(normalize_kernel<0>, dim3(grid_for(n)), dim3(256), 0, stream, in, minmax, out, n, lossy);
    else if (mode == 1) hipLaunchKernelGGL(normalize_kernel<1>, dim3(grid_for(n)), dim3(256), 0, stream, in, minmax, out, n, lossy);
    else if (mode == 2) hipLaunchKernelGGL(normalize_kernel<2>, dim3(grid_for(n)), dim3(256), 0, stream, in, minmax, out, n, lossy);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}

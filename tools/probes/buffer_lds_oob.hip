// Probe: does `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer descriptor) write ZEROS for lanes whose offset fails the
// descriptor's bounds check, and is the scalar offset (soffset) part of that check? (conv3h.hip stages its halo patch this way: lanes
// outside the image are given an out-of-range offset instead of a zero-page pointer.)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/buffer_lds_oob.hip -o tools/probes/_bin/buffer_lds_oob && tools/probes/_bin/buffer_lds_oob
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void k(const float* src, float* out, int nbytes, unsigned soff, int mode) {
    __shared__ __attribute__((aligned(16))) float smem[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) smem[i] = -1.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
    unsigned off = threadIdx.x * 16;
    if (mode == 1 && (threadIdx.x & 1)) off = 0xFFFFFFF0u;           // odd lanes out of range through voffset
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, off, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}

int main() {
    const int N = 4096;  // floats in the source; descriptor covers the first 1024 bytes (256 floats) in some runs
    std::vector<float> h(N);
    for (int i = 0; i < N; ++i) h[i] = 1.0f + i;
    float *d, *o;
    hipMalloc(&d, N * 4); hipMalloc(&o, 256 * 4);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    std::vector<float> r(256);
    auto run = [&](const char* what, int nbytes, unsigned soff, int mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, nbytes, soff, mode);
        hipMemcpy(r.data(), o, 256 * 4, hipMemcpyDeviceToHost);
        printf("%-70s lane0 %.0f lane1 %.0f lane2 %.0f lane3 %.0f ... lane62 %.0f lane63 %.0f\n", what, r[0], r[4], r[8], r[12], r[248], r[252]);
    };
    run("in range (expect 1 5 9 13 .. 249 253)", N * 4, 0, 0);
    run("odd lanes voffset 0xFFFFFFF0 (zeros expected on odd lanes)", N * 4, 0, 1);
    run("descriptor 512 B: lanes >= 32 out of range via num_records", 512, 0, 0);
    run("descriptor 1024 B, soffset 512: is soffset bounds-checked? (lanes>=32)", 1024, 512, 0);
    run("odd lanes voffset OOB + soffset 128", N * 4, 128, 1);
    return 0;
}

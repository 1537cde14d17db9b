// Micro-probe: what does one 64-lane global store instruction cost on gfx950 as a function of how many rows (cache lines) it
// touches? Emulates a GEMM tile epilogue: every workgroup (8 waves) owns a 256-row x W-byte block of a row-major matrix whose
// rows are `stride` bytes apart; a wave owns 32 rows and writes them with dwordx4 stores, each instruction covering R rows x
// (1024 / R) contiguous bytes. Prints cycles per store instruction per wave / per CU and the aggregate write rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/_bin/store_pattern tools/probes/store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int R, int W>
__global__ __launch_bounds__(512) void store_kernel(char* out, long stride, int tiles_per_row, int reps, long buf_rows, long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int SEG = 1024 / R;                 // contiguous bytes per row per instruction
    constexpr int SEGS = W / SEG;                 // instructions to cover one row group
    constexpr int LPR = SEG / 16;                 // lanes per row
    constexpr int NI = 32 * W / 1024;             // store instructions per wave per tile
    float4 v = make_float4(lane, wave, blockIdx.x, 1.f);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        long tile = (long)rep * gridDim.x + blockIdx.x;
        long row0 = ((tile / tiles_per_row) * 256) % buf_rows + wave * 32;
        long col0 = (tile % tiles_per_row) * W;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int rg = i / SEGS, sg = i % SEGS;
            long row = row0 + rg * R + lane / LPR;
            long col = col0 + sg * SEG + (lane % LPR) * 16;
            *reinterpret_cast<float4*>(out + row * stride + col) = v;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int R, int W>
void run(char* buf, long stride, long buf_rows, int grid, int reps, long long* dcyc) {
    int tiles_per_row = (int)(stride / W);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    store_kernel<R, W><<<grid, 512>>>(buf, stride, tiles_per_row, 2, buf_rows, dcyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    store_kernel<R, W><<<grid, 512>>>(buf, stride, tiles_per_row, reps, buf_rows, dcyc);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> c(grid * 8);
    CK(hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto x : c) mean += x; mean /= c.size();
    constexpr int NI = 32 * W / 1024;
    double per_wave = mean / (double(reps) * NI);
    double bytes = double(grid) * reps * 256.0 * W;
    printf("W=%4d B  rows/instr=%2d (%4d B/row)  grid=%4d: %7.1f memtime ticks per store per wave, %6.2f per CU-store; %7.1f us, %6.2f TB/s\n",
           W, R, 1024 / R, grid, per_wave, per_wave / 8.0, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
}


// GEMM-epilogue-like pattern: 256 x W-byte tile per workgroup, wave (grp = w>>2, wc = w&3).
//   PAT 0: each store = 16 rows x 64 B at column wc*64 (+ qn*256): the two halves of a 128-byte line come from DIFFERENT waves
//   PAT 1: each store =  8 rows x 128 B at column (wc>>1)*128 (+ qn*256), rows split by (wc&1): every store writes whole lines
template <int PAT>
__global__ __launch_bounds__(512) void tile_store_kernel(char* out, long stride, int tiles_per_row, int reps, long buf_rows, long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = wave >> 2, wc = wave & 3;
    float4 v = make_float4(lane, wave, blockIdx.x, 1.f);
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        long tile = (long)rep * gridDim.x + blockIdx.x;
        long row0 = ((tile / tiles_per_row) * 256) % buf_rows;
        long col0 = (tile % tiles_per_row) * 512;
#pragma unroll
        for (int qn = 0; qn < 2; ++qn)
#pragma unroll
            for (int qm = 0; qm < 2; ++qm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    long row, col;
                    if (PAT == 0) {
                        row = row0 + qm * 128 + grp * 64 + i * 16 + (lane & 15);
                        col = col0 + qn * 256 + wc * 64 + (lane >> 4) * 16;
                    } else {
                        row = row0 + qm * 128 + grp * 64 + i * 16 + (wc & 1) * 8 + (lane >> 3);
                        col = col0 + qn * 256 + (wc >> 1) * 128 + (lane & 7) * 16;
                    }
                    *reinterpret_cast<float4*>(out + row * stride + col) = v;
                }
    }
    __builtin_amdgcn_s_waitcnt(0);
    long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int PAT>
void run_tile(char* buf, long stride, long buf_rows, int grid, int reps, long long* dcyc) {
    int tiles_per_row = (int)(stride / 512);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    tile_store_kernel<PAT><<<grid, 512>>>(buf, stride, tiles_per_row, 2, buf_rows, dcyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    tile_store_kernel<PAT><<<grid, 512>>>(buf, stride, tiles_per_row, reps, buf_rows, dcyc);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> c(grid * 8);
    CK(hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto x : c) mean += x; mean /= c.size();
    printf("tile pattern %d (%s) grid=%4d reps=%3d: %7.1f ticks per store per wave, %6.2f per CU-store; %7.1f us, %6.2f TB/s\n", PAT,
           PAT == 0 ? "16 rows x 64 B, line halves from two waves" : "8 rows x 128 B, whole lines", grid, reps, mean / (reps * 16.0),
           mean / (reps * 16.0) / 8.0, ms * 1e3, double(grid) * reps * 256.0 * 512.0 / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    int reps = argc > 1 ? atoi(argv[1]) : 32;
    const long stride = 4096, buf_rows = 256L * 1024;   // 1 GiB window
    char* buf; long long* dcyc;
    CK(hipMalloc(&buf, stride * (buf_rows + 256))); CK(hipMalloc(&dcyc, 4096 * 8 * 8));
    CK(hipMemset(buf, 0, stride * buf_rows));
    for (int grid : {8, 256}) for (int r : {1, 4, 32}) { run_tile<0>(buf, stride, buf_rows, grid, r, dcyc); run_tile<1>(buf, stride, buf_rows, grid, r, dcyc); }
    for (int grid : {8, 32, 256, 1024}) {
        run<2, 512>(buf, stride, buf_rows, grid, reps, dcyc);
        run<4, 512>(buf, stride, buf_rows, grid, reps, dcyc);
        run<8, 512>(buf, stride, buf_rows, grid, reps, dcyc);
        run<16, 512>(buf, stride, buf_rows, grid, reps, dcyc);
        run<32, 512>(buf, stride, buf_rows, grid, reps, dcyc);
        run<64, 512>(buf, stride, buf_rows, grid, reps, dcyc);
        run<1, 1024>(buf, stride, buf_rows, grid, reps, dcyc);
        run<8, 1024>(buf, stride, buf_rows, grid, reps, dcyc);
        run<16, 1024>(buf, stride, buf_rows, grid, reps, dcyc);
    }
    return 0;
}

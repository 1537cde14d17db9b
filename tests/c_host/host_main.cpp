// A torch-free host of libmdpt: plain HIP runtime calls + the C ABI of include/mdpt.h, nothing else.
//   host_main <weights.bin> <input.bin> <output.bin>
// weights.bin: int32 cfg[11] {F, heads, blocks, reasm[4], base_gh, base_gw, fusion_ch, patch} int32 precision, int32 n,
//              then n x { int32 name_len, name bytes, int32 ndim, int64 shape[ndim], float data[prod(shape)] }
// input.bin:   int32 B, H, W then float image[B][3][H][W]   (RGB, normalised)
// output.bin:  float depth[B][H][W]
// Exercised by tests/test_gpu_c_host.py (build: hipcc host_main.cpp -I include -L csrc -lmdpt).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "mdpt.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s (%s:%d)\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)
#define MDCK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "libmdpt error %d: %s (%s:%d)\n", r_, mdpt_last_error(), __FILE__, __LINE__); return 3; } } while (0)

struct HostTensor { std::vector<int64_t> shape; std::vector<float> data; };

template <class T> static bool rd(FILE* f, T* v, size_t n = 1) { return fread(v, sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s weights.bin input.bin output.bin\n", argv[0]); return 1; }
    if (mdpt_abi_version() != MDPT_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    FILE* fw = fopen(argv[1], "rb");
    if (!fw) { perror("weights"); return 1; }
    int32_t c[11], precision, n;
    if (!rd(fw, c, 11) || !rd(fw, &precision) || !rd(fw, &n)) return 1;
    mdpt_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.features_per_token = c[0]; cfg.num_heads = c[1]; cfg.num_blocks = c[2];
    for (int i = 0; i < 4; ++i) cfg.reassembly_features[i] = c[3 + i];
    cfg.base_patch_grid_h = c[7]; cfg.base_patch_grid_w = c[8]; cfg.fusion_channels = c[9]; cfg.patch_size_px = c[10];
    cfg.precision = precision; cfg.family = MDPT_FAMILY_DAV2;
    std::map<std::string, HostTensor> host;
    for (int i = 0; i < n; ++i) {
        int32_t len, ndim;
        if (!rd(fw, &len)) return 1;
        std::string name(len, '\0');
        if (!rd(fw, &name[0], len) || !rd(fw, &ndim)) return 1;
        HostTensor t;
        t.shape.resize(ndim);
        if (ndim && !rd(fw, t.shape.data(), ndim)) return 1;
        size_t cnt = 1;
        for (auto s : t.shape) cnt *= (size_t)s;
        t.data.resize(cnt);
        if (!rd(fw, t.data.data(), cnt)) return 1;
        host[name] = std::move(t);
    }
    fclose(fw);

    mdpt_handle* h = nullptr;
    MDCK(mdpt_create(&cfg, &h));
    hipStream_t stream;
    HIPCK(hipStreamCreate(&stream));
    std::vector<void*> dev_weights;
    for (int i = 0; i < mdpt_num_weights(h); ++i) {
        const char* name = mdpt_weight_name(h, i);
        auto it = host.find(name);
        if (it == host.end()) { fprintf(stderr, "weights.bin lacks %s\n", name); return 1; }
        void* d = nullptr;
        HIPCK(hipMalloc(&d, it->second.data.size() * 4));
        HIPCK(hipMemcpy(d, it->second.data.data(), it->second.data.size() * 4, hipMemcpyHostToDevice));
        dev_weights.push_back(d);
        MDCK(mdpt_bind_weight(h, name, d, MDPT_DTYPE_F32, (int32_t)it->second.shape.size(), it->second.shape.data()));
    }
    size_t packed_bytes = 0;
    MDCK(mdpt_packed_bytes(h, &packed_bytes));
    void* packed = nullptr;
    HIPCK(hipMalloc(&packed, packed_bytes));
    MDCK(mdpt_finalize(h, packed, packed_bytes, stream));
    HIPCK(hipStreamSynchronize(stream));
    for (void* d : dev_weights) HIPCK(hipFree(d));  // only read during mdpt_finalize

    FILE* fi = fopen(argv[2], "rb");
    if (!fi) { perror("input"); return 1; }
    int32_t B, H, W;
    if (!rd(fi, &B) || !rd(fi, &H) || !rd(fi, &W)) return 1;
    std::vector<float> img((size_t)B * 3 * H * W), depth((size_t)B * H * W);
    if (!rd(fi, img.data(), img.size())) return 1;
    fclose(fi);
    void *d_img = nullptr, *d_depth = nullptr, *ws = nullptr;
    size_t ws_bytes = 0;
    MDCK(mdpt_workspace_bytes(h, B, H, W, &ws_bytes));
    HIPCK(hipMalloc(&d_img, img.size() * 4));
    HIPCK(hipMalloc(&d_depth, depth.size() * 4));
    HIPCK(hipMalloc(&ws, ws_bytes));
    HIPCK(hipMemcpyAsync(d_img, img.data(), img.size() * 4, hipMemcpyHostToDevice, stream));
    MDCK(mdpt_forward(h, d_img, MDPT_DTYPE_F32, B, H, W, d_depth, MDPT_DTYPE_F32, ws, ws_bytes, stream));
    HIPCK(hipMemcpyAsync(depth.data(), d_depth, depth.size() * 4, hipMemcpyDeviceToHost, stream));
    HIPCK(hipStreamSynchronize(stream));
    FILE* fo = fopen(argv[3], "wb");
    if (!fo || fwrite(depth.data(), 4, depth.size(), fo) != depth.size()) return 1;
    fclose(fo);
    mdpt_destroy(h);
    printf("C_HOST_OK B=%d H=%d W=%d weights=%d packed=%zu workspace=%zu\n", B, H, W, n, packed_bytes, ws_bytes);
    return 0;
}

// A torch-free host of libmdpt: plain HIP runtime calls + the C ABI of include/mdpt.h, nothing else. It runs a `.mdpt` model file (the
// deployment artefact written by DPTModel.export / muggled_dpt_amd/export.py - all four model families, any arithmetic mode, parameters in
// fp32 / bf16 / fp16) on tensors or on uint8 images of ANY legal size: the dynamic H / W axes of the reference's ONNX export
// (experiments/export_onnx.py:139-147) are a property of the handle here.
//
//   host_main model.mdpt input.bin output.bin                       tensor in -> depth out (DPTModel.forward, dpt_model.py:61-83)
//   host_main model.mdpt --image image.bin output.bin [max_side [square]]
//                                                                   uint8 BGR image in -> depth out (DPTModel.inference, dpt_model.py:87-109:
//                                                                   prepare_image on the device with the file's normalisation and size rule)
// input.bin:   int32 B, H, W, dtype (MDPT_DTYPE_*), then image[B][3][H][W] in that dtype (RGB, normalised)
// image.bin:   int32 h, w, then uint8 bgr[h][w][3]
// output.bin:  int32 B, H, W, dtype, then depth[B][H][W] in the model file's parameter dtype (what the reference returns, dpt_model.py:105-107)
// Several inputs may follow each other: `host_main model.mdpt in1.bin out1.bin in2.bin out2.bin ...` runs them on ONE handle (different sizes
// included: no re-finalisation). Exercised by tests/test_gpu_c_host.py (build: hipcc host_main.cpp -I include -L csrc -lmdpt).
#include <hip/hip_runtime.h>
#include <cmath>
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

struct HostTensor { int32_t dtype; std::vector<int64_t> shape; std::vector<char> data; };

struct ModelFile {
    mdpt_config cfg;
    int32_t passes[16], wrc, latency, tiling, default_side;
    float mean[3], std[3];
    std::map<std::string, HostTensor> tensors;
    int32_t param_dtype = MDPT_DTYPE_F32;
};

template <class T> static bool rd(FILE* f, T* v, size_t n = 1) { return fread(v, sizeof(T), n, f) == n; }
static bool align16(FILE* f) { const long p = ftell(f); return fseek(f, (16 - p % 16) % 16, SEEK_CUR) == 0; }
static size_t dtype_bytes(int dt) { return dt == MDPT_DTYPE_F32 ? 4 : 2; }

static int read_model(const char* path, ModelFile* m) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    char magic[8];
    int32_t abi, cfg_bytes, json_bytes, n;
    if (!rd(f, magic, 8) || memcmp(magic, "MDPTMDL1", 8) || !rd(f, &abi) || !rd(f, &cfg_bytes)) { fprintf(stderr, "%s: not an mdpt model file\n", path); return 1; }
    if (abi != MDPT_ABI_VERSION || cfg_bytes != (int32_t)sizeof(mdpt_config)) { fprintf(stderr, "%s: written for ABI %d, this host is ABI %d\n", path, abi, MDPT_ABI_VERSION); return 1; }
    if (!rd(f, &m->cfg) || !rd(f, m->passes, 16) || !rd(f, &m->wrc) || !rd(f, &m->latency) || !rd(f, m->mean, 3) || !rd(f, m->std, 3) ||
        !rd(f, &m->tiling) || !rd(f, &m->default_side) || !rd(f, &json_bytes) || fseek(f, json_bytes, SEEK_CUR) != 0 || !rd(f, &n)) return 1;
    for (int i = 0; i < n; ++i) {
        int32_t len, ndim;
        int64_t nbytes;
        if (!rd(f, &len)) return 1;
        std::string name(len, '\0');
        HostTensor t;
        if (!rd(f, &name[0], len) || !rd(f, &t.dtype) || !rd(f, &ndim)) return 1;
        t.shape.resize(ndim);
        if ((ndim && !rd(f, t.shape.data(), ndim)) || !rd(f, &nbytes) || !align16(f)) return 1;
        t.data.resize((size_t)nbytes);
        if (nbytes && !rd(f, t.data.data(), (size_t)nbytes)) return 1;
        if (!align16(f)) return 1;
        m->param_dtype = t.dtype;
        m->tensors[name] = std::move(t);
    }
    fclose(f);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s model.mdpt (input.bin | --image image.bin) output.bin [max_side [square]] ...\n", argv[0]); return 1; }
    if (mdpt_abi_version() != MDPT_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    ModelFile mf;
    if (read_model(argv[1], &mf)) return 1;

    mdpt_handle* h = nullptr;
    MDCK(mdpt_create(&mf.cfg, &h));
    for (int c = 0; c < MDPT_NUM_CLASSES; ++c)
        if (mf.passes[c]) MDCK(mdpt_set_class_passes(h, c, mf.passes[c]));
    if (mf.wrc >= 0) MDCK(mdpt_set_weight_rounding_compensation(h, mf.wrc));
    if (mf.latency) MDCK(mdpt_set_latency_mode(h, 1));
    hipStream_t stream;
    HIPCK(hipStreamCreate(&stream));
    std::vector<void*> dev_weights;
    for (int i = 0; i < mdpt_num_weights(h); ++i) {
        const char* name = mdpt_weight_name(h, i);
        auto it = mf.tensors.find(name);
        if (it == mf.tensors.end()) { fprintf(stderr, "the model file lacks %s\n", name); return 1; }
        void* d = nullptr;
        HIPCK(hipMalloc(&d, it->second.data.size() ? it->second.data.size() : 4));
        HIPCK(hipMemcpy(d, it->second.data.data(), it->second.data.size(), hipMemcpyHostToDevice));
        dev_weights.push_back(d);
        MDCK(mdpt_bind_weight(h, name, d, it->second.dtype, (int32_t)it->second.shape.size(), it->second.shape.data()));
    }
    size_t packed_bytes = 0;
    MDCK(mdpt_packed_bytes(h, &packed_bytes));
    void* packed = nullptr;
    HIPCK(hipMalloc(&packed, packed_bytes));
    MDCK(mdpt_finalize(h, packed, packed_bytes, stream));
    HIPCK(hipStreamSynchronize(stream));
    for (void* d : dev_weights) HIPCK(hipFree(d));  // only read during mdpt_finalize

    void* ws = nullptr;
    size_t ws_cap = 0;
    int runs = 0;
    for (int a = 2; a < argc;) {
        const bool from_image = !strcmp(argv[a], "--image");
        if (from_image) ++a;
        if (a + 1 >= argc) { fprintf(stderr, "missing output file\n"); return 1; }
        const char* in_path = argv[a];
        const char* out_path = argv[a + 1];
        a += 2;
        int32_t B = 1, H = 0, W = 0, in_dtype = mf.param_dtype;
        void* d_img = nullptr;
        FILE* fi = fopen(in_path, "rb");
        if (!fi) { perror(in_path); return 1; }
        if (from_image) {
            int max_side = mf.default_side, square = 1;
            if (a < argc && argv[a][0] >= '0' && argv[a][0] <= '9') { max_side = atoi(argv[a++]); if (a < argc && (argv[a][0] == '0' || argv[a][0] == '1') && !argv[a][1]) square = atoi(argv[a++]); }
            int32_t ih, iw;
            if (!rd(fi, &ih) || !rd(fi, &iw)) return 1;
            std::vector<unsigned char> px((size_t)ih * iw * 3);
            if (!rd(fi, px.data(), px.size())) return 1;
            // the reference's size rule (patch_embed.py:116-130): Python's round() is round-half-to-even = nearbyint in the default rounding mode
            const int largest = ih > iw ? ih : iw;
            const double scale = (double)max_side / largest;
            const int th = square ? largest : ih, tw = square ? largest : iw;
            auto snap = [&](int side) { const long r = (long)std::nearbyint(side * scale / mf.tiling); return (int)((r < 1 ? 1 : r) * mf.tiling); };
            H = snap(th); W = snap(tw);
            void* d_px = nullptr;
            HIPCK(hipMalloc(&d_px, px.size()));
            HIPCK(hipMalloc(&d_img, (size_t)3 * H * W * dtype_bytes(in_dtype)));
            HIPCK(hipMemcpyAsync(d_px, px.data(), px.size(), hipMemcpyHostToDevice, stream));
            MDCK(mdpt_prepare_image(d_px, ih, iw, d_img, in_dtype, H, W, mf.mean, mf.std, MDPT_INTERP_BILINEAR, stream));
            HIPCK(hipStreamSynchronize(stream));
            HIPCK(hipFree(d_px));
        } else {
            if (!rd(fi, &B) || !rd(fi, &H) || !rd(fi, &W) || !rd(fi, &in_dtype)) return 1;
            std::vector<char> img((size_t)B * 3 * H * W * dtype_bytes(in_dtype));
            if (!rd(fi, img.data(), img.size())) return 1;
            HIPCK(hipMalloc(&d_img, img.size()));
            HIPCK(hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice));
        }
        fclose(fi);
        const int32_t out_dtype = mf.param_dtype;
        std::vector<char> depth((size_t)B * H * W * dtype_bytes(out_dtype));
        void* d_depth = nullptr;
        size_t ws_bytes = 0;
        MDCK(mdpt_workspace_bytes(h, B, H, W, &ws_bytes));
        if (ws_bytes > ws_cap) {
            if (ws) HIPCK(hipFree(ws));
            HIPCK(hipMalloc(&ws, ws_bytes));
            ws_cap = ws_bytes;
        }
        HIPCK(hipMalloc(&d_depth, depth.size()));
        MDCK(mdpt_forward(h, d_img, in_dtype, B, H, W, d_depth, out_dtype, ws, ws_cap, stream));
        HIPCK(hipMemcpyAsync(depth.data(), d_depth, depth.size(), hipMemcpyDeviceToHost, stream));
        HIPCK(hipStreamSynchronize(stream));
        HIPCK(hipFree(d_img));
        HIPCK(hipFree(d_depth));
        FILE* fo = fopen(out_path, "wb");
        const int32_t hdr[4] = {B, H, W, out_dtype};
        if (!fo || fwrite(hdr, 4, 4, fo) != 4 || fwrite(depth.data(), 1, depth.size(), fo) != depth.size()) return 1;
        fclose(fo);
        printf("C_HOST_OK B=%d H=%d W=%d dtype=%d workspace=%zu%s\n", B, H, W, out_dtype, ws_bytes, from_image ? " (from a uint8 image)" : "");
        ++runs;
    }
    mdpt_destroy(h);
    printf("C_HOST_DONE runs=%d weights=%zu packed=%zu family=%d precision=%d\n", runs, mf.tensors.size(), packed_bytes, mf.cfg.family, mf.cfg.precision);
    return 0;
}

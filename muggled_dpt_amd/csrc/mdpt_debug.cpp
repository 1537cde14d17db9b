// libmdpt: test / measurement hooks of include/mdpt.h (used by tests/ and tools/ only) and the lazily resolved RCCL all-gather wrapper.
#include "mdpt_internal.h"

extern "C" {

// ---- test hooks (tests/ only): truncate the encoder after (block, step) and read raw internal buffers as fp32
int mdpt_debug_set_stop(mdpt_handle* h, int32_t block, int32_t step) {
    if (!h) return fail(MDPT_E_INVALID, "null handle");
    h->dbg_block = block; h->dbg_step = step;
    return 0;
}

int mdpt_debug_set_reassemble_overlap(mdpt_handle* h, int32_t on) {
    if (!h) return fail(MDPT_E_INVALID, "null handle");
    if (on < 0 || on > 2) return fail(MDPT_E_INVALID, "reassemble overlap: 0 = off, 1 = the library's rule, 2 = always");
    h->overlap_reasm = on;
    return 0;
}

// The power-of-two weight scale of the fp16 build (GemmParams::wscale) has two equally valid roundings of the same weights: the shipped rule scales a
// folded matrix only when its largest entry is below 2^-5, "all" scales every one. The tolerance claims of the mixed mode are asserted under BOTH
// (tests/test_gpu_precision_modes.py). Takes effect at the next mdpt_finalize.
int mdpt_debug_set_wscale_policy(mdpt_handle* h, int32_t all) {
    if (!h) return fail(MDPT_E_INVALID, "null handle");
    h->wscale_all = all ? 1 : 0;
    h->finalized = false;
    return 0;
}

int mdpt_debug_set_side_stream_priority(mdpt_handle* h, int32_t prio) {
    if (!h || prio < -1 || prio > 1) return fail(MDPT_E_INVALID, "null handle or priority class outside -1 .. 1");
    if (h->side_ncand) return fail(MDPT_E_STATE, "the side stream exists already (set the class before the first forward)");
    h->side_prio = prio;
    return 0;
}

int mdpt_debug_set_side_stream_probe(mdpt_handle* h, int32_t on) {
    if (!h) return fail(MDPT_E_INVALID, "null handle");
    if (h->side_ncand) return fail(MDPT_E_STATE, "the side stream exists already (set this before the first forward)");
    h->side_probe = on ? 1 : 0;
    return 0;
}

int mdpt_debug_side_stream_info(mdpt_handle* h, int32_t* candidates, int32_t* rejected) {
    if (!h || !candidates || !rejected) return fail(MDPT_E_INVALID, "null argument");
    *candidates = h->side_ncand;
    *rejected = h->side_rejected;
    return 0;
}

int mdpt_debug_set_ksplit_min(mdpt_handle* h, int32_t min_k_tiles, int32_t four_k_tiles) {
    if (!h || min_k_tiles < 2 || four_k_tiles < 4) return fail(MDPT_E_INVALID, "null handle or thresholds below 2 / 4 K tiles");
    h->ks_min_ktiles = min_k_tiles;
    h->ks_big_ktiles = four_k_tiles;
    return 0;
}

int mdpt_debug_read(mdpt_handle* h, const char* name, void* out_f32, size_t out_floats, void* workspace, size_t workspace_bytes,
                    void* stream) {
    if (!h || !name || !out_f32) return fail(MDPT_E_INVALID, "null argument");
    if (!h->has_last && !h->has_stage_plan) return fail(MDPT_E_STATE, "mdpt_debug_read needs a preceding mdpt_forward");
    if (h->swin) return fail(MDPT_E_UNSUPPORTED, "mdpt_debug_read: internal buffer names are defined for the ViT families only");
    Ctx c;
    c.h = h; c.p = h->has_last ? h->last_plan : h->stage_plan; c.ws = (char*)workspace; c.s = (hipStream_t)stream;
    CHK(check_ws(h, c.p, workspace, workspace_bytes));
    const Plan& p = c.p;
    const size_t rows = (size_t)p.B * p.npad;
    const std::string n = name;
    const size_t* planes = nullptr;
    size_t f32_off = SIZE_MAX, elems = 0;
    size_t bf16_only[2] = {SIZE_MAX, SIZE_MAX};  // a bf16 map without a lo plane
    const bool bf16_head = head_tail_fused(h) && mdpt_head_tail_scale_ok(8 * p.gh, 8 * p.gw, p.H, p.W);
    const size_t px[4] = {(size_t)16 * p.Np, (size_t)4 * p.Np, (size_t)p.Np, (size_t)p.Np / 4};
    if (n == "im2col") { planes = p.im2col; elems = (size_t)p.B * p.Np * h->Kpatch; }
    else if (n == "pos") { f32_off = p.pos; elems = (size_t)p.Np * h->F; }
    else if (n == "resid") { f32_off = p.resid; elems = rows * h->F; }
    else if (n == "xn") { planes = p.xn; elems = rows * h->F; }
    else if (n == "q") { planes = p.q; elems = (size_t)p.B * h->heads * p.npad * 64; }
    else if (n == "k") { planes = p.k; elems = (size_t)p.B * h->heads * p.npad * 64; }
    else if (n == "vt") { planes = p.vt; elems = (size_t)p.B * h->heads * 64 * p.npadv; }
    else if (n == "att") { planes = p.att; elems = rows * h->F; }
    else if (n == "hbuf") { planes = p.hbuf; elems = rows * 4 * h->F; }
    else if (n == "h1") {
        // bf16 mode with the fused head tail: the first conv writes a bf16 map into the fp32 map's buffer (run_head)
        elems = (size_t)p.B * 64 * p.Np * h->C2p;
        if (bf16_head) { bf16_only[0] = p.h1; if (h->terms(CLS_HEAD_TAIL) == 2) bf16_only[1] = p.h1 + elems * 2; planes = bf16_only; } else { f32_off = p.h1; }
    }
    else if (n == "h1u") {
        if (bf16_head) return fail(MDPT_E_STATE, "h1u does not exist on the fused head-tail path (the upsampled map only ever lives in LDS tiles)");
        planes = p.h1u; elems = (size_t)p.B * p.H * p.W * h->C2p;
    }
    else if (n == "fused") {
        // bf16 mode: the forward may have folded the x2 upsample into the head's first conv; rebuild the map the head saw (same arithmetic)
        if (head_upsamples_bf16(h))
            CHK(OPLC(mdpt_launch_upsample_bf16, c.at<op_t>(p.flo[0]), c.pl(p.fused).hi, p.B, 4 * p.gh, 4 * p.gw, 8 * p.gh, 8 * p.gw, h->Cp, c.s));
        planes = p.fused; elems = (size_t)p.B * 64 * p.Np * h->Cp;
    }
    else if (n == "u0") { planes = p.u0; elems = (size_t)p.B * px[0] * h->hidp[0]; }
    else if (n == "u1") { planes = p.u1; elems = (size_t)p.B * px[1] * h->hidp[1]; }
    else if (n == "d3") { planes = p.d3; elems = (size_t)p.B * px[3] * h->hidp[3]; }
    else if (n.size() == 2 && n[0] == 't' && n[1] >= '0' && n[1] <= '3') { const int i = n[1] - '0'; planes = p.t[i]; elems = (size_t)p.B * p.Np * h->hidp[i]; }
    else if (n.size() == 4 && n.compare(0, 3, "flo") == 0 && n[3] >= '0' && n[3] <= '3') {
        const int i = n[3] - '0';
        elems = (size_t)p.B * px[i] * h->Cp;
        if (i == 0 && head_upsamples_bf16(h)) { bf16_only[0] = p.flo[0]; planes = bf16_only; } else { f32_off = p.flo[i]; }  // level 0: bf16 map (run_fusion(c, true))
    }
    else if (n.size() == 3 && n.compare(0, 2, "xf") == 0 && n[2] >= '0' && n[2] <= '3') { const int i = n[2] - '0'; f32_off = p.x_f32[i]; elems = (size_t)p.B * px[i] * h->Cp; }
    else if (n.size() == 3 && n.compare(0, 2, "a1") == 0 && n[2] >= '0' && n[2] <= '3') { const int i = n[2] - '0'; planes = p.a1[i]; elems = (size_t)p.B * px[i] * h->Cp; }
    else if (n.size() == 3 && n.compare(0, 2, "b2") == 0 && n[2] >= '0' && n[2] <= '3') { const int i = n[2] - '0'; planes = p.b2[i]; elems = (size_t)p.B * px[i] * h->Cp; }
    else return fail(MDPT_E_INVALID, "unknown debug buffer \"%s\"", name);
    if (out_floats < elems) return fail(MDPT_E_WORKSPACE, "debug buffer %s needs %zu floats, got %zu", name, elems, out_floats);
    // reuse the token exporter as a flat converter: B=1, N=npad=elems/F' with F'=4 keeps indices simple
    if (planes) {
        Planes pl = c.pl(planes);
        CHK(OPLC(mdpt_launch_tokens_export, pl.hi, pl.lo, nullptr, (float*)out_f32, 1, (int)(elems / 4), (int)(elems / 4), 4, 0, c.s, pl.lo && planes != bf16_only ? pl.f8 : 0));
    } else {
        CHK(OPLC(mdpt_launch_tokens_export, nullptr, nullptr, c.at<float>(f32_off), (float*)out_f32, 1, (int)(elems / 4), (int)(elems / 4), 4, 0, c.s));
    }
    return 0;
}

// ---- test/bench hook: the plain dense GEMM kernel on caller-provided bf16 operands (out_f32[M,N] = A[M,K] W[N,K]^T)
int mdpt_debug_gemm(const void* a_bf16, const void* w_bf16, void* out_f32, void* out_bf16, int32_t M, int32_t N, int32_t K,
                    int32_t tile, int32_t iters, void* stream, void* dbg_times) {
    if (!a_bf16 || !w_bf16 || (!out_f32 && !out_bf16)) return fail(MDPT_E_INVALID, "null argument");
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.A_hi = (const op_t*)a_bf16; g.W_hi = (const op_t*)w_bf16;
    g.M = M; g.N = N; g.K = K; g.lda = K; g.npass = 1;
    g.zero_page = (const op_t*)w_bf16;
    g.amode = MDPT_A_DENSE; g.ekind = MDPT_E_GENERIC; g.tile = tile & 255;
    g.act = (tile >> 8) & 3;  // bits 8-9 of `tile`: epilogue activation (MDPT_ACT_*), for epilogue-cost measurements
    g.out_f32 = (float*)out_f32; g.out_hi = (op_t*)out_bf16; g.ldc = N; g.ldr = N;
    g.dbg_times = (unsigned long long*)dbg_times;
    if ((tile >> 10) & 1) {  // bit 10: in-place residual epilogue (proj / fc2 form); bias and gamma are read from the out_bf16 buffer
        if (!out_f32 || !out_bf16) return fail(MDPT_E_INVALID, "residual mode needs both output buffers");
        g.bias = (const float*)out_bf16; g.gamma = (const float*)out_bf16 + N; g.resid = (const float*)out_f32; g.out_hi = nullptr;
    }
    if ((tile >> 11) & 1) {  // bit 11: residual-initialised accumulators (the encoder's proj / fc2 form): out = (out + A W^T) + bias, in place
        if (!out_f32 || !out_bf16) return fail(MDPT_E_INVALID, "residual mode needs both output buffers");
        g.bias = (const float*)out_bf16; g.resid = (const float*)out_f32; g.out_hi = nullptr; g.acc_init = 1;
    }
    for (int i = 0; i < iters; ++i) CHK(OPLG(mdpt_launch_gemm, g, (hipStream_t)stream));
    return 0;
}

// ---- test/bench hook: one 3x3 stride-1 conv Cin -> Cout (256 | 128) on caller-provided operands (bf16 NHWC input, MDPT_PACK_CONV3 weights [Cout][9 Cin]):
//      path 0 = the implicit-GEMM kernels of gemm.hip (tile = MDPT_TILE_*), path 1 = the halo-staged kernel of conv3h.hip.
//      out = [skip +] conv + [bias] [+ up2(up)] -> out_f32 (optional) and out_bf16 (ReLU'd if relu_bf16); both paths use the same arithmetic
int mdpt_debug_conv3(const void* in_bf16, const void* w_packed_bf16, const void* bias_f32, const void* skip_f32, const void* up_f32, int32_t Hu,
                     int32_t Wu, void* out_f32, void* out_bf16, int32_t relu_bf16, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                     int32_t path, int32_t tile, int32_t iters, void* stream, void* dbg_times, const void* in_lo_bf16, const void* w_lo_bf16,
                     void* out_lo_bf16) {
    if (Cout != 256 && Cout != 128) return fail(MDPT_E_INVALID, "Cout must be 256 or 128");
    if (w_lo_bf16 && !in_lo_bf16) return fail(MDPT_E_INVALID, "a lo plane of the weights needs one of the input (three passes); the input alone = two passes");
    if (!in_bf16 || !w_packed_bf16 || (!out_bf16 && !out_f32)) return fail(MDPT_E_INVALID, "null argument");
    static op_t* zero_page = nullptr;  // test hook only: allocated once, never freed
    if (!zero_page) {
        if (hipMalloc((void**)&zero_page, 256) != hipSuccess || hipMemset(zero_page, 0, 256) != hipSuccess) return fail(MDPT_E_STATE, "zero page allocation failed");
    }
    if (path == 2 || path == 3) {
        // upsampled input: in_bf16 is the SOURCE map [B, Hu, Wu, Cin]; the conv runs on its bilinear upsample to H x W.
        // path 2 = interpolated inside the halo-staged kernel; path 3 = stand-alone bf16 upsample into out_lo_bf16 (scratch [B, H, W, Cin])
        // followed by the implicit-GEMM conv (tile) - the two must agree bit for bit
        if (Cout != 128 || !out_bf16 || Hu < 2 || Wu < 2) return fail(MDPT_E_INVALID, "upsampled-input form: 128 output channels, bf16 output");
        if (path == 2) {
            Conv3hParams q;
            memset(&q, 0, sizeof(q));
            q.up_in = (const op_t*)in_bf16; q.Hs = Hu; q.Ws = Wu; q.w = (const op_t*)w_packed_bf16; q.bias = (const float*)bias_f32;
            q.out_bf = (op_t*)out_bf16; q.B = B; q.H = H; q.W = W; q.Cin = Cin; q.Cout = 128;
            q.dbg_times = (unsigned long long*)dbg_times;
            if (!OPLG(mdpt_conv3h_supported, q)) return fail(MDPT_E_UNSUPPORTED, "conv3h does not cover this combination");
            for (int i = 0; i < iters; ++i) CHK(OPLG(mdpt_launch_conv3h, q, (hipStream_t)stream));
            return 0;
        }
        if (!out_lo_bf16) return fail(MDPT_E_INVALID, "path 3 needs a scratch map in out_lo_bf16");
        for (int i = 0; i < iters; ++i) {
            CHK(OPLG(mdpt_launch_upsample_bf16, (const op_t*)in_bf16, (op_t*)out_lo_bf16, B, Hu, Wu, H, W, Cin, (hipStream_t)stream));
            GemmParams g;
            memset(&g, 0, sizeof(g));
            g.A_hi = (const op_t*)out_lo_bf16; g.W_hi = (const op_t*)w_packed_bf16;
            g.M = B * H * W; g.N = 128; g.K = 9 * Cin; g.lda = Cin; g.npass = 1; g.zero_page = zero_page;
            g.amode = MDPT_A_CONV3; g.ekind = MDPT_E_GENERIC; g.tile = tile;
            g.Hi = H; g.Wi = W; g.Cin = Cin; g.Ho = H; g.Wo = W; g.cstride = 1;
            g.bias = (const float*)bias_f32; g.out_hi = (op_t*)out_bf16; g.ldc = 128; g.ldr = 128;
            CHK(OPLG(mdpt_launch_gemm, g, (hipStream_t)stream));
        }
        return 0;
    }
    if (path == 1) {
        Conv3hParams q;
        memset(&q, 0, sizeof(q));
        q.in = (const op_t*)in_bf16; q.w = (const op_t*)w_packed_bf16; q.bias = (const float*)bias_f32; q.skip = (const float*)skip_f32;
        q.in_lo = (const op_t*)in_lo_bf16; q.w_lo = (const op_t*)w_lo_bf16; q.out_bf_lo = (op_t*)out_lo_bf16;
        q.up_src = (const float*)up_f32; q.Hu = Hu; q.Wu = Wu; q.out_f32 = (float*)out_f32; q.out_bf = (op_t*)out_bf16; q.relu_bf = relu_bf16;
        q.B = B; q.H = H; q.W = W; q.Cin = Cin; q.Cout = Cout;
        q.dbg_times = (unsigned long long*)dbg_times;
        if (!OPLG(mdpt_conv3h_supported, q)) return fail(MDPT_E_UNSUPPORTED, "conv3h does not cover this combination");
        for (int i = 0; i < iters; ++i) CHK(OPLG(mdpt_launch_conv3h, q, (hipStream_t)stream));
        return 0;
    }
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.A_hi = (const op_t*)in_bf16; g.W_hi = (const op_t*)w_packed_bf16; g.A_lo = (const op_t*)in_lo_bf16; g.W_lo = (const op_t*)w_lo_bf16;
    g.M = B * H * W; g.N = Cout; g.K = 9 * Cin; g.lda = Cin; g.npass = in_lo_bf16 ? (w_lo_bf16 ? 3 : 2) : 1;  // (input lo plane without a weight lo plane: the two-pass form)
    g.zero_page = zero_page;
    g.amode = MDPT_A_CONV3; g.ekind = MDPT_E_GENERIC; g.tile = tile;
    g.Hi = H; g.Wi = W; g.Cin = Cin; g.Ho = H; g.Wo = W; g.cstride = 1;
    g.bias = (const float*)bias_f32; g.resid = (const float*)skip_f32; g.ldr = Cout;
    g.up_src = (const float*)up_f32; g.Hu = Hu; g.Wu = Wu;
    g.out_f32 = (float*)out_f32; g.out_hi = (op_t*)out_bf16; g.out_lo = (op_t*)out_lo_bf16; g.relu_bf16 = relu_bf16; g.ldc = Cout;
    g.dbg_times = (unsigned long long*)dbg_times;
    for (int i = 0; i < iters; ++i) CHK(OPLG(mdpt_launch_gemm, g, (hipStream_t)stream));
    return 0;
}

// ---- test hook: the fused attention kernel on caller-provided head-major operands (bf16 mode, head dim 64, no bias):
//      Q, K [B, heads, npad, 64] (Q pre-scaled by 1/8), Vt [B, heads, 64, npadv] (pad columns zero) -> out [B * npad, heads * 64]
int mdpt_debug_attention(const void* q_bf16, const void* k_bf16, const void* vt_bf16, void* out_bf16, int32_t B, int32_t heads, int32_t N,
                         int32_t npad, int32_t npadv, int32_t iters, void* stream) {
    if (!q_bf16 || !k_bf16 || !vt_bf16 || !out_bf16) return fail(MDPT_E_INVALID, "null argument");
    AttnParams a;
    memset(&a, 0, sizeof(a));
    a.q_hi = (const op_t*)q_bf16; a.k_hi = (const op_t*)k_bf16; a.vt_hi = (const op_t*)vt_bf16; a.out_hi = (op_t*)out_bf16;
    a.B = B; a.heads = heads; a.N = N; a.npad = npad; a.npadv = npadv; a.F = heads * 64;
    for (int i = 0; i < iters; ++i) CHK(OPLG(mdpt_launch_attention, a, (hipStream_t)stream));
    return 0;
}

// ---- RCCL all-gather wrapper (resolved lazily so the library itself has no link-time dependency on RCCL)
int mdpt_allgather(void* comm, const void* send_dev, void* recv_dev, size_t count_per_rank, int32_t dtype, void* stream) {
    typedef int (*allgather_fn)(const void*, void*, size_t, int, void*, void*);
    static allgather_fn fn = nullptr;
    if (!fn) {
        void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) return fail(MDPT_E_STATE, "cannot load librccl.so: %s", dlerror());
        fn = (allgather_fn)dlsym(lib, "ncclAllGather");
        if (!fn) return fail(MDPT_E_STATE, "ncclAllGather not found in librccl.so");
    }
    // ncclDataType_t: ncclFloat16 = 6, ncclFloat32 = 7, ncclBfloat16 = 9 (rccl.h)
    if (dtype != MDPT_DTYPE_F32 && dtype != MDPT_DTYPE_BF16 && dtype != MDPT_DTYPE_F16) return fail(MDPT_E_INVALID, "bad dtype %d", dtype);
    const int nccl_type = dtype == MDPT_DTYPE_F32 ? 7 : (dtype == MDPT_DTYPE_BF16 ? 9 : 6);
    const int rc = fn(send_dev, recv_dev, count_per_rank, nccl_type, comm, stream);
    if (rc != 0) return fail(MDPT_E_STATE, "ncclAllGather failed with code %d", rc);
    return 0;
}

}  // extern "C"

#!/usr/bin/env python3
"""A/B of the two in-place residual GEMM forms on the ViT-L encoder shapes (M = 41728 = 32 x 1304 rows; proj: N = K = 1024, fc2: K = 4096):
  resid : accumulate from zero, epilogue reads the residual back, scales, adds, stores   (round-1 form, gemm8_kernel<0,0,2>)
  rinit : accumulators start at the residual tile (loads in flight under the first K tiles), epilogue = bias add + store (gemm8_kernel<0,0,6>)
Interleaved rounds in ONE process (cdna_hip_programming.md rule 24), uniform random [-1,1) operands, per-tile s_memtime stamps of one launch."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native

lib = native.load()
stream = torch.cuda.current_stream().cuda_stream
M = int(sys.argv[1]) if len(sys.argv) > 1 else 41728
ROUNDS, ITERS = 6, 10
for (N, K) in ((1024, 1024), (1024, 4096)):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    out = torch.randn(M, N, device="cuda", dtype=torch.float32)
    aux = torch.zeros(max(M * N, 4 * N), device="cuda", dtype=torch.bfloat16)
    aux.view(torch.float32)[:N] = torch.linspace(-1, 1, N, device="cuda")
    aux.view(torch.float32)[N:2 * N] = 1e-3  # gamma of the resid form: tiny, so that repeated in-place updates stay finite
    flags = {"resid": 5 | (1 << 10), "rinit": 5 | (1 << 11)}
    times = {k: [] for k in flags}
    for r in range(ROUNDS):
        for mode, fl in flags.items():
            native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), out.data_ptr(), aux.data_ptr(), M, N, K, fl, 2, stream, None))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), out.data_ptr(), aux.data_ptr(), M, N, K, fl, ITERS, stream, None))
            e1.record(); torch.cuda.synchronize()
            times[mode].append(e0.elapsed_time(e1) * 1e3 / ITERS)
            out.normal_()
    for mode, ts in times.items():
        med, mn = float(np.median(ts)), float(np.min(ts))
        print(f"M={M} N={N} K={K} {mode:5s}: median {med:7.1f} us ({2.0 * M * N * K / med / 1e6:6.0f} TF = {2.0 * M * N * K / med / 1e6 / 2500:.3f} of peak)  min {mn:7.1f} us   rounds {['%.1f' % t for t in ts]}", flush=True)
    nblk = ((M + 255) // 256) * (N // 256)
    for mode, fl in flags.items():
        dbg = torch.zeros(nblk * 6 + nblk * 16, dtype=torch.int64, device="cuda")
        native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), out.data_ptr(), aux.data_ptr(), M, N, K, fl, 1, stream, dbg.data_ptr()))
        torch.cuda.synchronize()
        d = dbg.cpu().numpy().astype(np.int64)[:nblk * 6].reshape(nblk, 6)
        pro, loop, epi = (d[:, 1] - d[:, 0]), (d[:, 2] - d[:, 1]), (d[:, 3] - d[:, 2])
        print(f"   stamps {mode:5s}: prologue {pro.mean():7.0f}  loop {loop.mean():8.0f} ({loop.mean() / (K // 64):6.0f} per K tile)  epilogue {epi.mean():7.0f}  "
              f"tile total {(d[:, 3] - d[:, 0]).mean():8.0f} cycles", flush=True)
        out.normal_()

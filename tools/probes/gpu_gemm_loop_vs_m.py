#!/usr/bin/env python3
"""Does the 8-phase GEMM main loop slow down when the A operand has to come from HBM? Per-tile s_memtime stamps (prologue / loop / epilogue)
of the fc1 / fc2 / proj shapes at row counts whose A panel is L2-resident, MALL-resident or neither."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native

lib = native.load()
stream = torch.cuda.current_stream().cuda_stream
for (N, K, tag) in [(4096, 1024, "fc1"), (1024, 4096, "fc2"), (1024, 1024, "proj")]:
    for M in (2048, 8192, 41728, 166912):
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        outb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        nblk = ((M + 255) // 256) * ((N + 255) // 256)
        native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), None, outb.data_ptr(), M, N, K, 5, 2, stream, None))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), None, outb.data_ptr(), M, N, K, 5, 5, stream, None))
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        dbg = torch.zeros(nblk * 6 + nblk * 16, dtype=torch.int64, device="cuda")
        native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), None, outb.data_ptr(), M, N, K, 5, 1, stream, dbg.data_ptr()))
        torch.cuda.synchronize()
        d = dbg.cpu().numpy().astype(np.int64)[:nblk * 6].reshape(nblk, 6)
        pro, loop, epi = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
        tf = 2.0 * M * N * K / us * 1e-6
        print(f"{tag:5s} M={M:6d} ({nblk:5d} tiles, A {M * K * 2 / 1e6:6.0f} MB): {us:8.1f} us {tf:6.0f} TFLOP/s | ticks/tile: prologue {pro.mean():6.0f}  "
              f"loop {loop.mean():7.0f} ({loop.mean() / (K // 64):5.0f}/K-tile, p10 {np.percentile(loop, 10) / (K // 64):5.0f} p90 {np.percentile(loop, 90) / (K // 64):5.0f})  "
              f"epilogue {epi.mean():6.0f}", flush=True)
        del a, w, outb

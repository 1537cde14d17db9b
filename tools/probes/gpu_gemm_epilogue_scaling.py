#!/usr/bin/env python3
"""Does the GEMM epilogue cost depend on how many workgroups run it at the same time? (store-burst hypothesis)
Same 256x256x64 kernel, K = 1024, N = 1024; M chosen so that 4 / 32 / 128 / 256 / 652 workgroups run."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native

lib = native.load()
stream = torch.cuda.current_stream().cuda_stream
N, K = 1024, 1024
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for M in (256, 16384, 41728):
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    outb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for mode in ("f32", "bf16", "gelu", "resid", "rinit"):
        nblk = ((M + 255) // 256) * (N // 256)
        dbg = torch.zeros(nblk * 6 + nblk * 16, dtype=torch.int64, device="cuda")
        o32 = out.data_ptr() if mode in ("f32", "resid", "rinit") else None
        o16 = outb.data_ptr() if mode in ("bf16", "gelu", "resid", "rinit") else None
        tl = tile | (2 << 8 if mode == "gelu" else 0) | (1 << 10 if mode == "resid" else 0) | (1 << 11 if mode == "rinit" else 0)
        if mode in ("resid", "rinit"):
            outb.zero_(); out.zero_()
        native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), o32, o16, M, N, K, tl, 2, stream, None))
        torch.cuda.synchronize()
        native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), o32, o16, M, N, K, tl, 1, stream, dbg.data_ptr()))
        torch.cuda.synchronize()
        d = dbg.cpu().numpy().astype(np.int64)[:nblk * 6].reshape(nblk, 6)
        pro, loop, epi = (d[:, 1] - d[:, 0]), (d[:, 2] - d[:, 1]), (d[:, 3] - d[:, 2])
        issue = d[:, 4] - d[:, 2]
        if mode == "resid":
            e = dbg.cpu().numpy().astype(np.int64)[nblk * 6:].reshape(nblk, 16)
            st = e[:, 1:6] - e[:, 0:5]
            print("      resid phases (wave 0 mean): issue L01 %5.0f | wait L01 %5.0f | compute + issue L23,S01 %5.0f | wait L23 %5.0f | compute + issue S23 %5.0f"
                  % tuple(st.mean(axis=0)), flush=True)
        print(f"M={M:6d} blocks={nblk:4d} out={mode:4s}: prologue {pro.mean():7.0f}  loop/kstep {loop.mean()/16:6.0f}  epilogue mean {epi.mean():7.0f} "
              f"p10 {np.percentile(epi,10):6.0f} p50 {np.percentile(epi,50):6.0f} p90 {np.percentile(epi,90):6.0f}  (issue-only, wave 0: {issue.mean():7.0f})", flush=True)

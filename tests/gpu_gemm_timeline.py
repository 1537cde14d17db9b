#!/usr/bin/env python3
"""Per-workgroup phase timeline of the dense GEMM kernel (s_memtime stamps): where does a block spend its time?"""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native

lib = native.load()
stream = torch.cuda.current_stream().cuda_stream
for (M, N, K, tag) in [(41728, 1024, 1024, "proj"), (41728, 4096, 1024, "fc1"), (41728, 1024, 4096, "fc2")]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    outb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for tile, (bm, bn) in ((2, (256, 256)), (4, (256, 128)), (1, (128, 128))):
        for mode in ("f32", "bf16"):
            nblk = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
            dbg = torch.zeros(nblk * 6, dtype=torch.int64, device="cuda")
            o32 = out.data_ptr() if mode == "f32" else None
            o16 = outb.data_ptr() if mode == "bf16" else None
            native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), o32, o16, M, N, K, tile, 2, stream, None))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), o32, o16, M, N, K, tile, 1, stream, dbg.data_ptr()))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3
            d = dbg.cpu().numpy().reshape(nblk, 6).astype(np.int64)
            pro = (d[:, 1] - d[:, 0]).astype(np.float64)
            loop = (d[:, 2] - d[:, 1]).astype(np.float64)
            epi = (d[:, 3] - d[:, 2]).astype(np.float64)
            ksteps = K // (32 if tile == 4 else 64)
            # per-XCC span (each XCD has its own s_memtime base): kernel duration in ticks on that XCD
            xcc = d[:, 4] & 15
            span = np.mean([d[xcc == x, 3].max() - d[xcc == x, 0].min() for x in set(xcc)])
            print(f"{tag:5s} tile={tile} out={mode}: {us:7.1f} us = {span:9.0f} ticks ({span/us:6.1f} ticks/us), {nblk:5d} blocks | ticks/block: "
                  f"prologue {pro.mean():7.0f}  loop {loop.mean():8.0f} ({loop.mean()/ksteps:6.0f}/kstep)  epilogue {epi.mean():7.0f} "
                  f"(p10 {np.percentile(epi,10):6.0f} p90 {np.percentile(epi,90):6.0f}) total {(pro+loop+epi).mean():8.0f}", flush=True)

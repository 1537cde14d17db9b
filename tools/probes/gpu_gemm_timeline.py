#!/usr/bin/env python3
"""Per-workgroup phase timeline of the dense GEMM kernel (s_memtime stamps): where does a block spend its time?"""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native

lib = native.load()
stream = torch.cuda.current_stream().cuda_stream
for (M, N, K, tag) in [(41728, 1024, 1024, "proj"), (41728, 4096, 1024, "fc1"), (41728, 1024, 4096, "fc2")]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    outb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for tile, (bm, bn) in ((2, (256, 256)), (5, (256, 256))):
        for mode in ("f32", "bf16"):
            nblk = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
            dbg = torch.zeros(nblk * 6 + nblk * 16, dtype=torch.int64, device="cuda")
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
            full = dbg.cpu().numpy().astype(np.int64)
            d = full[:nblk * 6].reshape(nblk, 6)
            q = full[nblk * 6:].reshape(nblk, 2, 8)
            pro = (d[:, 1] - d[:, 0]).astype(np.float64)
            loop = (d[:, 2] - d[:, 1]).astype(np.float64)
            epi = (d[:, 3] - d[:, 2]).astype(np.float64)
            ksteps = K // (32 if tile in (4, 5) else 64)
            # per-XCC span (each XCD has its own s_memtime base): kernel duration in ticks on that XCD
            xcc = d[:, 4] & 15
            span = np.mean([d[xcc == x, 3].max() - d[xcc == x, 0].min() for x in set(xcc)])
            print(f"{tag:5s} tile={tile} out={mode}: {us:7.1f} us = {span:9.0f} ticks ({span/us:6.1f} ticks/us), {nblk:5d} blocks | ticks/block: "
                  f"prologue {pro.mean():7.0f}  loop {loop.mean():8.0f} ({loop.mean()/ksteps:6.0f}/kstep)  epilogue {epi.mean():7.0f} "
                  f"(p10 {np.percentile(epi,10):6.0f} p90 {np.percentile(epi,90):6.0f}) total {(pro+loop+epi).mean():8.0f}", flush=True)
            for wv in (0, 1):
                if tile == 5:
                    qq = q[:, wv, :4].astype(np.float64)
                    ok = qq[:, 0] > 0
                    dd = np.diff(qq[ok], axis=1)
                    print(f"        mid slab, wave {'0' if wv == 0 else '4'}: C(prev, grp1 only) {dd[:,0].mean():6.0f} | L-phase {dd[:,1].mean():6.0f} | C(grp0 only) {dd[:,2].mean():6.0f}", flush=True)
                    continue
                qq = q[:, wv, :5].astype(np.float64)
                ok = qq[:, 0] > 0
                dd = np.diff(qq[ok], axis=1)
                print(f"        mid K-step, wave {'0' if wv == 0 else 'NW/2'}: wait+barrier {dd[:,0].mean():6.0f} | issue DMA {dd[:,1].mean():6.0f} | first LDS data {dd[:,2].mean():6.0f} | MFMA issue {dd[:,3].mean():6.0f} | sum {dd.sum(axis=1).mean():6.0f}", flush=True)

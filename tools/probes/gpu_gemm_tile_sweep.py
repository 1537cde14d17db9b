#!/usr/bin/env python3
"""Which main-loop variant wins for which problem size? (feeds the AUTO rule in gemm.hip launch_tile)
tile 1 = 128x128x64 lockstep, 5 = 256x256x64 8-phase, 6 = 64x64x64."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native
lib = native.load()
stream = torch.cuda.current_stream().cuda_stream
for M in (577, 1304, 2608, 4616, 9232, 10376, 20752):
    for (N, K) in ((1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096), (384, 384), (1536, 384)):
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        res = {}
        for tile in (1, 5, 6):
            native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, tile, 3, stream, None))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, tile, 20, stream, None))
            e1.record(); torch.cuda.synchronize()
            res[tile] = e0.elapsed_time(e1) * 50
        best = min(res, key=res.get)
        t256 = ((M + 255) // 256) * ((N + 255) // 256); t128 = ((M + 127) // 128) * ((N + 127) // 128)
        print(f"M={M:6d} N={N:5d} K={K:5d} tiles256={t256:4d} tiles128={t128:5d}: " + "  ".join(f"t{t}={v:7.1f}us" for t, v in res.items()) + f"  best t{best}", flush=True)

#!/usr/bin/env python3
"""All main-loop variants on the ViT-L encoder GEMM shapes (half batch M = 20752 and full batch M = 41504): TFLOP/s per variant.
tile 1 = 128x128x64 lockstep (2 WG/CU), 2 = 256x256x64 lockstep, 4 = 256x128x32 3-deep ring (2 WG/CU), 5 = 256x256x64 8-phase."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native
lib = native.load()
stream = torch.cuda.current_stream().cuda_stream
tiles = [int(t) for t in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 4, 5]
for M in (20752, 41504):
    for (N, K) in ((1024, 1024), (3072, 1024), (4096, 1024), (1024, 4096)):
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        res = {}
        for tile in tiles:
            native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, tile, 3, stream, None))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, tile, 20, stream, None))
            e1.record(); torch.cuda.synchronize()
            res[tile] = e0.elapsed_time(e1) * 50
        print(f"M={M:6d} N={N:5d} K={K:5d}: " + "  ".join(f"t{t}={v:7.1f}us ({2.0 * M * N * K / v / 1e6:6.0f} TF)" for t, v in res.items()), flush=True)

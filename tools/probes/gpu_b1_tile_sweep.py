#!/usr/bin/env python3
"""Batch-1 GEMM shapes (M = one image's token rows) on every main-loop variant, bf16 output and the residual-initialised fp32 form:
which tile should the latency rule pick per (N, K)?  tile 1 = 128x128, 2 = 256x256 lockstep, 4 = 256x128x32, 5 = 8-phase 256x256, 6 = 64x64."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native
lib = native.load()
stream = torch.cuda.current_stream().cuda_stream
shapes = [(1304, 3072, 1024, "qkv L"), (1304, 1024, 1024, "proj L"), (1304, 4096, 1024, "fc1 L"), (1304, 1024, 4096, "fc2 L"),
          (1304, 1152, 384, "qkv S"), (1304, 384, 384, "proj S"), (1304, 1536, 384, "fc1 S"), (1304, 384, 1536, "fc2 S"),
          (584, 3072, 1024, "qkv beit"), (584, 1024, 4096, "fc2 beit")]
for (M, N, K, tag) in shapes:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    rinit = tag.startswith(("proj", "fc2"))
    out16 = torch.zeros(max(M * N, 4 * N), device="cuda", dtype=torch.bfloat16)
    out32 = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    res = {}
    for tile in (6, 1, 2, 4, 5):  # (6 = 64x64: a 3-deep ring from 32 K tiles on, 2-deep below)
        flags = tile | ((1 << 11) if rinit else 0)
        args = (a.data_ptr(), w.data_ptr(), out32.data_ptr() if rinit else None, out16.data_ptr(), M, N, K, flags)
        native.check(lib, lib.mdpt_debug_gemm(*args, 3, stream, None))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.check(lib, lib.mdpt_debug_gemm(*args, 50, stream, None))
        e1.record(); torch.cuda.synchronize()
        res[tile] = e0.elapsed_time(e1) * 20
    best = min(res, key=res.get)
    print(f"{tag:9s} M={M:5d} N={N:5d} K={K:5d} {'rinit' if rinit else 'bf16 '}: " + "  ".join(f"t{t}={v:6.1f}us" for t, v in res.items()) +
          f"  best t{best} ({2e-6 * M * N * K / res[best]:.0f} TF/s)", flush=True)

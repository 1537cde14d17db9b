#!/usr/bin/env python3
"""SwinV2-L stage 0 / 1 GEMM shapes at batch 16 (K = 192: 3 K tiles, the 8-phase kernel handles pairs only) on every main-loop variant, plus
K padded to 256: which tile should the rule fall back to, and does zero-padding K pay?  bf16 out; tile ids as in gpu_b1_tile_sweep.py."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native
lib = native.load()
stream = torch.cuda.current_stream().cuda_stream
shapes = [(147456, 576, 192, "s0 qkv"), (147456, 576, 256, "s0 qkv, K padded"), (147456, 768, 192, "s0 fc1"), (147456, 768, 256, "s0 fc1, K padded"),
          (147456, 192, 192, "s0 proj"), (147456, 192, 256, "s0 proj, K padded"), (147456, 192, 768, "s0 fc2"),
          (36864, 1152, 384, "s1 qkv"), (36864, 1536, 384, "s1 fc1"), (36864, 384, 384, "s1 proj"), (36864, 384, 1536, "s1 fc2")]
for (M, N, K, tag) in shapes:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    out16 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    for tile in (1, 2, 4, 5, 6):
        args = (a.data_ptr(), w.data_ptr(), None, out16.data_ptr(), M, N, K, tile)
        native.check(lib, lib.mdpt_debug_gemm(*args, 2, stream, None))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.check(lib, lib.mdpt_debug_gemm(*args, 10, stream, None))
        e1.record(); torch.cuda.synchronize()
        res[tile] = e0.elapsed_time(e1) * 100
    best = min(res, key=res.get)
    mb = (M * K * 2 + N * K * 2 + M * N * 2) / 1e6
    print(f"{tag:20s} M={M:6d} N={N:5d} K={K:5d}: " + "  ".join(f"t{t}={v:7.1f}us" for t, v in res.items()) + f"  best t{best}  ({mb:.0f} MB: {mb / res[best]:.2f} TB/s)", flush=True)

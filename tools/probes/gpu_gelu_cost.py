#!/usr/bin/env python3
"""What does the fused GELU cost in the fc1 epilogue? Same GEMM (M=41504, N=4096, K=1024, bf16 out, 8-phase tile) with act = none / relu / gelu."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native
lib = native.load()
stream = torch.cuda.current_stream().cuda_stream
M, N, K = 41504, 4096, 1024
a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16) * 0.05
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for act, name in ((0, "none"), (1, "relu"), (2, "gelu")):
    tile = 5 | (act << 8)
    native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, tile, 3, stream, None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, tile, 20, stream, None))
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"act={name:5s}: {us:7.1f} us  {2.0 * M * N * K / us * 1e-6:7.1f} TF", flush=True)

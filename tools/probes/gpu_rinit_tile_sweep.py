#!/usr/bin/env python3
"""The residual GEMMs (out = (resid + A W^T) + bias, fp32 residual stream in place) of ViT-L batch 32 on every tile variant, alone on the
GPU: proj (K = 1024) has a memory floor of 78 us (427 MB), so a form with two workgroups per CU might beat the 8-phase tile there."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native

lib = native.load()
stream = torch.cuda.current_stream().cuda_stream
names = {5: "8-phase 256x256", 2: "lockstep 256x256", 4: "lockstep 256x128x32 (2 WG/CU)", 1: "lockstep 128x128 (2 WG/CU)", 6: "lockstep 64x64"}
for (M, N, K, tag) in [(41728, 1024, 1024, "proj"), (41728, 1024, 4096, "fc2"), (20864, 1024, 1024, "proj, half batch")]:
    a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device="cuda") * 2 - 1) / K ** 0.5).to(torch.bfloat16)
    out32 = torch.randn(M, N, device="cuda")
    aux = torch.zeros(max(M * N, 4 * N), device="cuda", dtype=torch.bfloat16)
    for tile in (5, 2, 4, 1, 6):
        flags = tile | (1 << 11)
        native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), out32.data_ptr(), aux.data_ptr(), M, N, K, flags, 3, stream, None))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), out32.data_ptr(), aux.data_ptr(), M, N, K, flags, 20, stream, None))
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{tag:18s} M={M} K={K}  {names[tile]:32s} {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:7.0f} TFLOP/s", flush=True)

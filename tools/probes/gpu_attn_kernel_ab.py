#!/usr/bin/env python3
"""The bare attention kernel at the ViT-L shapes (mdpt_debug_attention: 32 x 16 heads x 1297 tokens x 64; 8 x 16 x 5477 x 64), 30 launches
back to back, median of 5 rounds. Environment switches of a -DMDPT_DEBUG_SWITCHES build select the variant (MDPT_ATTN_PRIO, MDPT_ATTN_WIDE)."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native
lib = native.load()
stream = torch.cuda.current_stream().cuda_stream
for (B, H, N) in ((32, 16, 1297), (8, 16, 5477)):
    npad, npadv = (N + 7) // 8 * 8, (N + 63) // 64 * 64
    g = torch.Generator().manual_seed(0)
    qf = torch.randn(B, H, npad, 64, generator=g)
    cmax = os.environ.get("MDPT_ATTN_CMAX") is not None  # log2-domain scores: Q pre-scaled by log2(e) / 8 instead of 1 / 8
    q = (qf * (0.125 * 1.4426950408889634 if cmax else 0.125)).to(torch.bfloat16).cuda()
    k = torch.randn(B, H, npad, 64, generator=g).to(torch.bfloat16).cuda()
    vt = torch.randn(B, H, 64, npadv, generator=g)
    vt[..., N:] = 0
    vt = vt.to(torch.bfloat16).cuda()
    out = torch.empty(B * npad, H * 64, device="cuda", dtype=torch.bfloat16)
    run = lambda it: native.check(lib, lib.mdpt_debug_attention(q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), B, H, N, npad, npadv, it, stream))
    run(3); torch.cuda.synchronize()
    res = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(30); e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 30)
    # accuracy against an fp64 softmax of image 0, head 0 (same rounded K / V, Q from the fp32 values)
    kk, vv = k[0, 0, :N].double().cpu(), vt[0, 0, :, :N].double().cpu()
    att = torch.softmax((qf[0, 0, :N].double() * 0.125) @ kk.T, dim=-1) @ vv.T
    err = float((out[:N, :64].double().cpu() - att).abs().max() / att.abs().max())
    flops = 4.0 * B * H * N * N * 64
    us = float(np.median(res))
    print(f"B={B} N={N}: {us:8.1f} us  ({flops / us * 1e-6:6.0f} TFLOP/s = {flops / us * 1e-6 / 2500:.3f} of peak)  checksum {float(out.float().abs().sum()):.6e}  err vs fp64 {err:.2e}", flush=True)

#!/usr/bin/env python3
"""Where does the halo-staged conv kernel start to beat the implicit-GEMM path at SMALL batches (the conv3h dispatch threshold in
mdpt_stages.cpp conv3_to_fusion / run_head)? 256 -> 256 at 144^2 / 72^2 and the head's 256 -> 128 at 288^2, batch 1 / 2 / 4 / 8, bf16 + ReLU form."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from muggled_dpt_amd import native
from test_gpu_conv3h import _pack

lib = native.load()
stream = torch.cuda.current_stream().cuda_stream
for (H, Cout) in ((144, 256), (72, 256), (288, 128)):
    for B in (1, 2, 4, 8):
        g = torch.Generator().manual_seed(0)
        Cin = 256
        x = torch.randn(B, H, H, Cin, generator=g).to(torch.bfloat16).cuda()
        wp = _pack(torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(torch.bfloat16).cuda()
        bias = torch.randn(Cout, generator=g).cuda()
        obf = torch.empty(B, H, H, Cout, device="cuda", dtype=torch.bfloat16)

        def launch(path, iters):
            native.check(lib, lib.mdpt_debug_conv3(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), None, None, 0, 0, None, obf.data_ptr(), int(Cout == 256), B, H, H, Cin, Cout,
                                                   path, 0, iters, stream, None, None, None, None))
        res = {0: [], 1: []}
        for path in (0, 1):
            launch(path, 3)
        torch.cuda.synchronize()
        for rnd in range(5):
            for path in (0, 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); launch(path, 10); e1.record(); torch.cuda.synchronize()
                res[path].append(e0.elapsed_time(e1) * 100)
        tiles = B * ((H + 15) // 16) ** 2
        print(f"{H}x{H} -> {Cout}, B={B} ({tiles:4d} conv3h tiles, {(B * H * H + 255) // 256:5d} x256 rows): implicit-gemm {np.median(res[0]):7.1f} us   conv3h {np.median(res[1]):7.1f} us", flush=True)

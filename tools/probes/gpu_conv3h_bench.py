#!/usr/bin/env python3
"""Halo-staged conv kernel vs the implicit-GEMM path on the decoder's 256-channel shapes: time per launch (interleaved rounds) and the
per-tile s_memtime phase stamps (prologue / main loop / epilogue) of both kernels."""
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
VARIANTS = {"bf16+relu": (False, False, True, False), "skip->bf16": (True, False, False, False), "f32+bf16relu": (False, True, True, False),
            "skip+up->f32+bf16": (True, True, True, True)}
shapes = [(32, 144, 144, 256, 256)] + ([(32, 288, 288, 256, 128)] if "--head" in sys.argv else [])
for (B, H, W, Cin, Cout) in shapes:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).cuda()
    wp = _pack(torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(torch.bfloat16).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    skip = torch.randn(B, H, W, Cout, generator=g).cuda() if Cout == 256 else None
    up = torch.randn(B, H // 2, W // 2, Cout, generator=g).cuda() if Cout == 256 else None
    obf = torch.empty(B, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    o32 = torch.empty(B, H, W, Cout, device="cuda", dtype=torch.float32) if Cout == 256 else None
    flops = 2.0 * B * H * W * Cout * 9 * Cin
    for name, (has_skip, want_f32, relu, has_up) in (VARIANTS.items() if Cout == 256 else [("bias->bf16 (head conv1)", (False, False, False, False))]):
        def launch(path, iters, dbg=None):
            native.check(lib, lib.mdpt_debug_conv3(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), skip.data_ptr() if has_skip else None,
                                                   up.data_ptr() if has_up else None, H // 2 if has_up else 0, W // 2 if has_up else 0,
                                                   o32.data_ptr() if want_f32 else None, obf.data_ptr(), int(relu), B, H, W, Cin, Cout, path, 0, iters, stream,
                                                   dbg.data_ptr() if dbg is not None else None, None, None, None))
        res = {0: [], 1: []}
        for path in (0, 1):
            launch(path, 2)
        torch.cuda.synchronize()
        for rnd in range(5):
            for path in (0, 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); launch(path, 4); e1.record(); torch.cuda.synchronize()
                res[path].append(e0.elapsed_time(e1) * 1e3 / 4)
        line = f"{B}x{H}x{W}x{Cin} {name:20s}:"
        for path, tag in ((0, "implicit-gemm"), (1, "conv3h")):
            us = float(np.median(res[path]))
            line += f"  {tag} {us:7.1f} us ({flops / us * 1e-6:6.0f} TFLOP/s, {flops / us * 1e-6 / 2500:.3f})"
        print(line, flush=True)
        for path, tag in ((0, "implicit-gemm"), (1, "conv3h")):
            nblk = B * ((H + 15) // 16) * ((W + 15) // 16) if path == 1 else (B * H * W + 255) // 256
            dbg = torch.zeros(nblk * 6 + nblk * 16, dtype=torch.int64, device="cuda")
            launch(path, 1, dbg); torch.cuda.synchronize()
            d = dbg.cpu().numpy().astype(np.int64)[:nblk * 6].reshape(nblk, 6)
            ok = d[:, 3] > d[:, 0]
            d = d[ok]
            pro, loop, epi = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
            print(f"        {tag:14s} ticks/tile: prologue {pro.mean():7.0f}  loop {loop.mean():8.0f} ({loop.mean() / (9 * Cin // 64):6.0f}/K-tile)  "
                  f"epilogue {epi.mean():7.0f} (issued after {(d[:, 4] - d[:, 2]).mean() if path == 1 else float('nan'):7.0f})  total {(pro + loop + epi).mean():8.0f}", flush=True)

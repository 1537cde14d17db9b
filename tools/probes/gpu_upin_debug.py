#!/usr/bin/env python3
"""Where does the upsampled-input conv3h form differ from upsample + implicit GEMM? (debugging aid)"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from muggled_dpt_amd import native
from test_gpu_conv3h import _pack
lib = native.load()
B, H, W, Cin = 1, 32, 32, 256
Hs, Ws = H // 2, W // 2
g = torch.Generator().manual_seed(3)
src = torch.randn(B, Hs, Ws, Cin, generator=g).to(torch.bfloat16)
mode = sys.argv[1] if len(sys.argv) > 1 else "rand"
w = torch.zeros(128, Cin, 3, 3)
if mode == "tap":  # output channel n = tap t, input channel c: out[n] = in[tap n % 9 shifted][channel n] -> exposes the halo image directly
    for n in range(128):
        w[n, n % Cin, (n % 9) // 3, (n % 9) % 3] = 1.0
else:
    w = torch.randn(128, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
w = w.to(torch.bfloat16)
bias = torch.zeros(128)
sd, wp, bd = src.cuda(), _pack(w.float()).to(torch.bfloat16).cuda(), bias.cuda()
stream = torch.cuda.current_stream().cuda_stream
def run(path, tile):
    out = torch.full((B, H, W, 128), float("nan"), device="cuda", dtype=torch.bfloat16)
    scratch = torch.full((B, H, W, Cin), float("nan"), device="cuda", dtype=torch.bfloat16)
    native.check(lib, lib.mdpt_debug_conv3(sd.data_ptr(), wp.data_ptr(), bd.data_ptr(), None, None, Hs, Ws, None, out.data_ptr(), 0, B, H, W, Cin, 128, path, tile, 1, stream, None, None, None, scratch.data_ptr()))
    torch.cuda.synchronize()
    return out.float().cpu(), scratch.float().cpu()
a, _ = run(2, 0)
b, up = run(3, 6)
d = (a != b)
print("mismatching elements:", int(d.sum()), "of", d.numel(), "max abs diff", float((a - b).abs().max()))
idx = d.nonzero()
print("first mismatches (b, y, x, n):", idx[:20].tolist())
if len(idx):
    ys = idx[:, 1].unique().tolist(); xs = idx[:, 2].unique().tolist(); ns = idx[:, 3].unique().tolist()
    print("rows", ys[:40]); print("cols", xs[:40]); print("channels", ns[:40], "count", len(ns))
    for (bb, y, x, n) in idx[:8].tolist():
        print((y, x, n), "conv3h", float(a[bb, y, x, n]), "ref", float(b[bb, y, x, n]))

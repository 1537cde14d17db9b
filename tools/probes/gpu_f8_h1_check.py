#!/usr/bin/env python3
"""Head conv 1 with fp8 cross terms: the fp32 map it writes (debug buffer h1) against the CPU emulation of exactly that conv, element by element:
which output channels / pixels deviate when the lockstep GEMM tiles run it (small batch)?"""
import os, sys, torch
import torch.nn.functional as TF
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests", "precision_budget"))
import emulate_operand_rounding as emu
from tests.test_gpu_f8_cross import _model
from muggled_dpt_amd import native
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = int(sys.argv[2]) if len(sys.argv) > 2 else 4
model, cfg, w = _model({"head": native.PASSES_2F8, "head_tail": 3})
S = 8 * g
fused = torch.randn(batch, 256, S, S, generator=torch.Generator().manual_seed(9)) * 1.5
y = model.head(fused.cuda())
eng = model._get_engine()
out = torch.empty(batch * S * S * 128, device="cuda", dtype=torch.float32)
ws_ptr, ws_bytes = eng.workspace(batch, (g * 14, g * 14))
native.check(eng.lib, eng.lib.mdpt_debug_read(eng.handle, b"h1", out.data_ptr(), out.numel(), ws_ptr, ws_bytes, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
h1 = out.cpu().reshape(batch, S, S, 128).permute(0, 3, 1, 2)[:1]
W, b = w["head.spatial_upsampler.0.weight"], w["head.spatial_upsampler.0.bias"]
x = fused[:1]
xh, wh = emu.rnd(x, "f16"), emu.rnd(W, "f16")
main = TF.conv2d(xh, wh, b, padding=1)
cross = TF.conv2d(emu.sf8_act(x - xh, 16), emu.sf8_weight(wh, 0), None, padding=1)
ref = main + cross
d = (h1 - ref).abs()
print(f"batch {batch} map {S}x{S}: |h1 - (main + fp8 cross)| max {float(d.max()):.3e} mean {float(d.mean()):.3e}; |cross| max {float(cross.abs().max()):.3e} mean {float(cross.abs().mean()):.3e}; |h1 - main| max {float((h1 - main).abs().max()):.3e}")
per_ch = d.amax(dim=(0, 2, 3))
top = torch.topk(per_ch, 8)
print("  worst output channels:", [(int(i), f"{float(v):.2e}") for v, i in zip(top.values, top.indices)])
per_px = d.amax(dim=(0, 1))
tp = torch.topk(per_px.flatten(), 8)
print("  worst pixels:", [((int(i) // S, int(i) % S), f"{float(v):.2e}") for v, i in zip(tp.values, tp.indices)])
frac = float((d > 10 * d.mean()).float().mean())
print(f"  elements with error > 10 x mean: {100 * frac:.3f} %")
# is the deviating part explained by ONE input channel block / tap? error against cross terms computed without saturation / with other roundings
for name, alt in (("cross with e5m2 residue NOT saturated", TF.conv2d(((x - xh).double() * 65536).float().clamp(-1e9, 1e9).mul(1 / 65536).float(), emu.sf8_weight(wh, 0), None, padding=1)),):
    print(f"  vs {name}: max {float((h1 - main - alt).abs().max()):.3e}")
per_x = d.amax(dim=(0, 1, 2))
print("  per-x max error:", " ".join(f"{float(v):.0e}" for v in per_x))
per_y = d.amax(dim=(0, 1, 3))
print("  per-y max error:", " ".join(f"{float(v):.0e}" for v in per_y[:16]))
# which input channel blocks / taps explain the deviation at the worst element?
i = int(torch.argmax(d)); n_, y_, x_ = (i // (S * S)) % 128, (i // S) % S, i % S
print(f"  worst element: channel {n_} pixel ({y_}, {x_}): h1 {float(h1[0, n_, y_, x_]):.6f} ref {float(ref[0, n_, y_, x_]):.6f} main {float(main[0, n_, y_, x_]):.6f} cross {float(cross[0, n_, y_, x_]):.3e}")
lo = emu.sf8_act(x - xh, 16); w8 = emu.sf8_weight(wh, 0)
xp = TF.pad(lo, (1, 1, 1, 1))
for cb in range(2):
    for tap in range(9):
        ky, kx = tap // 3, tap % 3
        part = float((xp[0, cb * 128:(cb + 1) * 128, y_ + ky, x_ + kx] * w8[n_, cb * 128:(cb + 1) * 128, ky, kx]).sum())
        print(f"    block {cb} tap {tap}: {part:+.3e}", end="")
    print()
print(f"  deviation h1 - ref = {float(h1[0, n_, y_, x_] - ref[0, n_, y_, x_]):+.3e}")
# which (channel block, tap) partial sums are missing at the deviating rows? least squares of the deviation on the 18 partial products
parts = []
for cb in range(2):
    for tap in range(9):
        ky, kx = tap // 3, tap % 3
        xs = xp[:, cb * 128:(cb + 1) * 128, ky:ky + S, kx:kx + S]
        parts.append(TF.conv2d(xs, w8[:, cb * 128:(cb + 1) * 128, ky:ky + 1, kx:kx + 1]))
P = torch.stack(parts, dim=-1)[0]          # [128, S, S, 18]
dev = (h1 - ref)[0]                         # [128, S, S]
cols = [c for c in range(S) if float(per_x[c]) > 1e-5]
sel = torch.zeros(S, S, dtype=torch.bool); sel[1:, cols] = True
A_ = P[:, sel].reshape(-1, 18).double(); b_ = dev[:, sel].reshape(-1).double()
coef = torch.linalg.lstsq(A_, b_.unsqueeze(1)).solution[:, 0]
print("  deviating columns:", cols)
print("  least-squares share of each (block, tap) partial sum in the deviation (-1 = missing):")
print("   block 0:", " ".join(f"{float(v):+.2f}" for v in coef[:9]))
print("   block 1:", " ".join(f"{float(v):+.2f}" for v in coef[9:]))
# ... and which 16-channel chunks of (block 0, tap 0)?
parts = [TF.conv2d(xp[:, c * 16:(c + 1) * 16, 0:S, 0:S], w8[:, c * 16:(c + 1) * 16, 0:1, 0:1]) for c in range(8)]
P8 = torch.stack(parts, dim=-1)[0]
for cset in ([27], [31]):
    sel = torch.zeros(S, S, dtype=torch.bool); sel[1:, cset] = True
    A_ = P8[:, sel].reshape(-1, 8).double(); b_ = dev[:, sel].reshape(-1).double()
    coef = torch.linalg.lstsq(A_, b_.unsqueeze(1)).solution[:, 0]
    print(f"  column {cset}: share of the eight 16-channel chunks of (block 0, tap 0):", " ".join(f"{float(v):+.2f}" for v in coef))

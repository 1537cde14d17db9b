#!/usr/bin/env python3
"""Where does image 0 of a batch first differ (bitwise) from the same image run alone? (GPU diagnosis helper)"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict, native
from tests.helpers import synthetic_model
from tools.probes.gpu_diagnose import dbg_read, set_stop

name, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dtype = torch.bfloat16 if len(sys.argv) < 5 or sys.argv[4] == "bf16" else torch.float32
osd, cfg, w = synthetic_model(name, 0)
_, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
model = model.to("cuda", dtype)
x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(7)).to("cuda", dtype)
F, heads = cfg["features_per_token"], cfg["num_heads"]
N = (S // 14) ** 2 + 1
npad = (N + 7) // 8 * 8
npadv = (N + 63) // 64 * 64
steps = [(0, "xn", F), (1, "q", None), (1, "k", None), (1, "vt", None), (2, "att", F), (3, "resid", F), (4, "xn", F), (5, "hbuf", 4 * F), (6, "resid", F)]
for blk in (0, cfg["num_blocks"] - 1):
    for step, buf, width in steps:
        outs = []
        for xb in (x, x[:1]):
            b = xb.shape[0]
            set_stop(model, blk, step)
            model(xb)
            torch.cuda.synchronize()
            if buf in ("q", "k"):
                t = dbg_read(model, buf, b * heads * npad * 64, b, (S, S)).view(b, heads, npad, 64)[0, :, :N]
            elif buf == "vt":
                t = dbg_read(model, buf, b * heads * 64 * npadv, b, (S, S)).view(b, heads, 64, npadv)[0, :, :, :N]
            else:
                t = dbg_read(model, buf, b * npad * width, b, (S, S)).view(b, npad, width)[0, :N]
            outs.append(t)
        d = (outs[0] - outs[1]).abs()
        print(f"block {blk} step {step} {buf:6s}: equal={torch.equal(outs[0], outs[1])} max|diff|={float(d.max()):.3e} n_diff={int((d > 0).sum())}", flush=True)
set_stop(model, -1, -1)
y = model(x); y1 = model(x[:1])
d = (y[0].float() - y1[0].float()).abs()
print("final depth: equal", torch.equal(y[0], y1[0]), "max diff", float(d.max()), "n_diff", int((d > 0).sum()))
for i, nm in enumerate(["tap0", "tap1", "tap2", "tap3", "reasm0", "reasm1", "reasm2", "reasm3", "fused"]):
    model(x); tb = model.debug_taps(B, (S, S))
    model(x[:1]); t1 = model.debug_taps(1, (S, S))
    a = (tb["stages"] + tb["reasm"] + [tb["fused"]])[i][0]; b_ = (t1["stages"] + t1["reasm"] + [t1["fused"]])[i][0]
    print(nm, "equal", torch.equal(a, b_), "max diff", float((a - b_).abs().max()))

#!/usr/bin/env python3
"""Fusion-stage locator: compare internal decoder buffers of image 0 between (batch B, tile tb) and (batch 1, tile t1)."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
from tests.helpers import synthetic_model
from tools.probes.gpu_diagnose import dbg_read

name, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
osd, cfg, w = synthetic_model(name, 0)
x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(7)).to("cuda", torch.bfloat16)
g = S // 14
Cp = (cfg["fusion_channels"] + 63) // 64 * 64
px = [16 * g * g, 4 * g * g, g * g, g * g // 4]

def run(batch, tile):
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    model = model.to("cuda", torch.bfloat16)
    model.set_gemm_tile(tile)
    xb = x[:batch]
    model(xb); torch.cuda.synchronize()
    out = {}
    for lv in (3, 2, 1, 0):
        for nm in ("a1", "xf", "b2", "flo"):
            if nm in ("a1", "xf") and lv == 3:
                continue
            out[f"{nm}{lv}"] = dbg_read(model, f"{nm}{lv}", batch * px[lv] * Cp, batch, (S, S)).view(batch, -1)[0].clone()
    return out

for (ba, ta, bb, tb) in ((B, 0, 1, 0), (B, 1, 1, 1), (B, 2, 1, 2), (1, 1, 1, 2), (B, 1, B, 2)):
    A, Bq = run(ba, ta), run(bb, tb)
    print(f"--- (B={ba},tile={ta}) vs (B={bb},tile={tb})")
    for k in A:
        d = (A[k] - Bq[k]).abs()
        print(f"   {k}: equal={torch.equal(A[k], Bq[k])} max|diff|={float(d.max()):.3e} n_diff={int((d>0).sum())}/{d.numel()}", flush=True)

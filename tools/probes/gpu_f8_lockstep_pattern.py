#!/usr/bin/env python3
"""Where does the lockstep fp8 path deviate from the emulation? Head stage at batch 2 (lockstep conv) against the emulation, error map statistics."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests", "precision_budget"))
import emulate_operand_rounding as emu
from oracle import dpt_oracle as orc
from tests.test_gpu_f8_cross import _model, _policy
from muggled_dpt_amd import native
emu.FINE = True
model, cfg, w = _model({"head": native.PASSES_2F8, "head_tail": 3})
for batch, g in ((2, 4), (2, 8), (1, 16), (32, 8)):
    fused = torch.randn(batch, 256, 8 * g, 8 * g, generator=torch.Generator().manual_seed(9)) * 1.5
    pol = _policy(emu, head_conv1="f16x2a@sf8", head_conv2="f16x3")
    ref = emu.emulated_call(orc.head, w, pol, w, cfg, fused[:1])
    y = model.head(fused.cuda()).cpu()[:1]
    d = (y - ref).abs()[0]
    mx = float(ref.abs().max())
    H = d.shape[0]
    b = max(2, H // 8)
    print(f"batch {batch} map {8*g}x{8*g}: max err {float(d.max())/mx:.2e}; border band {float(torch.cat([d[:b].flatten(), d[-b:].flatten(), d[:, :b].flatten(), d[:, -b:].flatten()]).max())/mx:.2e}; interior {float(d[b:-b, b:-b].max())/mx:.2e}; mean {float(d.mean())/mx:.2e}")
    rows = d.max(dim=1).values / mx
    print("   row maxima:", " ".join(f"{float(v):.0e}" for v in rows[:: max(1, H // 16)]))

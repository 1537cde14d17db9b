#!/usr/bin/env python3
"""Per-phase s_memtime stamps of head_tail_kernel / head_tail2_kernel (MDPT_HEAD_DBG=1 debug hook; library built with
MDPT_EXTRA_HIPCC_FLAGS=-DMDPT_DEBUG_SWITCHES) on the headline shape: ViT-L head, 288^2 -> 504^2.
   python tools/probes/gpu_head_tail_phases.py [batch] [bf16 | mixed]"""
import os, sys, time
os.environ["MDPT_HEAD_DBG"] = "1"
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
osd = make_synthetic_original_state_dict("vitl", 0)
_, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
dt = torch.bfloat16 if mode == "bf16" else torch.float32
model = model.to("cuda", dt)
if mode != "bf16":
    model.set_precision(mode)
fused = torch.randn(int(sys.argv[1]) if len(sys.argv) > 1 else 16, 256, 288, 288, device="cuda", dtype=dt)
for _ in range(3):
    y = model.head(fused)
torch.cuda.synchronize()

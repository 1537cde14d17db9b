#!/usr/bin/env python3
"""One-off full-size check at the other BASELINE size: ViT-L, 1036x1036 (5477 tokens), one image, both arithmetic modes vs the CPU oracle."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components, get_model_config_from_state_dict
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
from oracle import dpt_oracle

torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
osd = make_synthetic_original_state_dict("vitl", 0)
cfg = get_model_config_from_state_dict(osd)
w = flatten_components(convert_state_dict_keys(cfg, osd))
x = torch.randn(1, 3, 1036, 1036, generator=torch.Generator().manual_seed(1))
t0 = time.perf_counter()
ref = dpt_oracle.forward(w, cfg, x)
print(f"oracle: {time.perf_counter() - t0:.1f} s on {torch.get_num_threads()} threads, depth max {float(ref.max()):.3f}", flush=True)
for dtype in (torch.float32, torch.bfloat16):
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    y = model.to("cuda", dtype)(x.to("cuda", dtype)).float().cpu()
    rel = float((y.double() - ref.double()).abs().max() / ref.double().abs().max())
    print(f"{dtype}: rel err vs CPU fp32 oracle = {rel:.3e}", flush=True)
    del model

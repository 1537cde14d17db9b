#!/usr/bin/env python3
"""One-off: the headline workload (ViT-L, 504x504, batch 32) - every image of the batch against the CPU oracle, both modes."""
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
x = torch.randn(32, 3, 504, 504, generator=torch.Generator().manual_seed(1))
t0 = time.perf_counter()
ref = torch.cat([dpt_oracle.forward(w, cfg, x[i:i + 4]) for i in range(0, 32, 4)])
print(f"oracle: {time.perf_counter() - t0:.1f} s for 32 images", flush=True)
for dtype in (torch.float32, torch.bfloat16):
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    y = model.to("cuda", dtype)(x.to("cuda", dtype)).float().cpu()
    per = [(float((y[i].double() - ref[i].double()).abs().max() / ref[i].double().abs().max())) for i in range(32)]
    print(f"{dtype}: rel err per image: max {max(per):.3e}  median {sorted(per)[16]:.3e}  min {min(per):.3e}", flush=True)
    del model

#!/usr/bin/env python3
"""Fused head tail (head.hip) vs the oracle head on random fused maps: error map statistics by position inside the 16x16 tiles."""
import os, sys
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components, get_model_config_from_state_dict
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
from oracle import dpt_oracle

for name, gh, gw in (("tiny", 4, 6), ("vits", 6, 6), ("vitl", 4, 4)):
    osd = make_synthetic_original_state_dict(name, 0)
    cfg = get_model_config_from_state_dict(osd)
    w = flatten_components(convert_state_dict_keys(cfg, osd))
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    model = model.to("cuda", torch.bfloat16)
    C = cfg["fusion_channels"]
    fused = torch.randn(2, C, 8 * gh, 8 * gw, generator=torch.Generator().manual_seed(3))
    ref = dpt_oracle.head(w, cfg, fused)
    got = model.head(fused.to("cuda", torch.bfloat16)).float().cpu()
    err = (got - ref).abs()
    scale = float(ref.abs().max())
    print(f"{name}: C={C} out {tuple(got.shape)} max|ref| {scale:.3f} rel err max {float(err.max()) / scale:.3e} mean {float(err.mean()) / scale:.3e}")
    e = err[0].numpy() / scale
    H, W = e.shape
    ty, tx = np.unravel_index(np.argmax(e), e.shape)
    print(f"   worst pixel ({ty},{tx}) in-tile ({ty % 16},{tx % 16}); rows with err>5e-2: {np.where(e.max(axis=1) > 5e-2)[0][:20]} cols: {np.where(e.max(axis=0) > 5e-2)[0][:20]}")

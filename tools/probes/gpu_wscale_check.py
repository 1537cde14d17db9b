#!/usr/bin/env python3
"""fp16 weight-scale check on a model whose residual stream is as small as its layer-scaled updates (image, patch bias, cls, pos and gammas x f):
depth error vs the CPU oracle per mode, and block 1's proj / fc2 updates fp16x3 / fp16 against bf16x3."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from tests.test_gpu_precision_modes import _resid_after
from tests.helpers import seeded_input, synthetic_model, rel_err
from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components, get_model_config_from_state_dict
from oracle import dpt_oracle
osd0, _, _ = synthetic_model("vits", 0)
x0 = seeded_input((2, 3, 140, 112), 23)
for f in (1.0, 1e-2, 1e-3, 1e-4):
    osd = {k: (v * f if (".ls1.gamma" in k or ".ls2.gamma" in k or k in ("pretrained.cls_token", "pretrained.pos_embed", "pretrained.patch_embed.proj.bias")) else v.clone()) for k, v in osd0.items()}
    x = x0 * f
    cfg = get_model_config_from_state_dict(osd)
    w = flatten_components(convert_state_dict_keys(cfg, osd))
    ref = dpt_oracle.forward(w, cfg, x)
    upd, errs = {}, {}
    for prec in ("bf16x3", "fp16x3", "fp16", "mixed"):
        _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
        model = model.to("cuda", torch.float32); model.set_precision(prec)
        errs[prec] = rel_err(model(x.cuda()).cpu(), ref)
        for name, before, after in (("proj", 2, 3), ("fc2", 5, 6)):
            upd[(prec, name)] = (_resid_after(model, x.cuda(), 1, after).double() - _resid_after(model, x.cuda(), 1, before).double())
        del model
    print(f"scale {f:g}: depth rel err " + " ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    for name in ("proj", "fc2"):
        r = upd[("bf16x3", name)]
        print(f"   {name}: |update| max {float(r.abs().max()):.3e}  fp16x3 vs bf16x3 {float((upd[('fp16x3', name)] - r).abs().max() / r.abs().max()):.3e}  "
              f"fp16 {float((upd[('fp16', name)] - r).abs().max() / r.abs().max()):.3e}", flush=True)

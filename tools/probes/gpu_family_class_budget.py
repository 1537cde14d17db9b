#!/usr/bin/env python3
"""BEiT-L / SwinV2-L (BASELINE configs[4], the reference fixtures of tests/golden/): error against the fixture and maps/s at batch 16 of the mixed
mode with one decoder class at a time moved between 1 / 2 / 3 passes - the per-family budget behind mdpt_default_mixed_passes' family rows.
   python tools/probes/gpu_family_class_budget.py [beitl swinl vitl]"""
import os, sys, time
import numpy as np
import torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bench
from helpers import seeded_input

FIX = {"beitl": ("beit_large_384", 384, 16), "swinl": ("swin2_large_384", 384, 16), "vitl": ("vitl504", 504, 32)}
R04 = {"fusion": 3, "fusion_proj": 3, "head": 3, "head_tail": 3}
VARIANTS = [("shipped", {}), ("round 4 (3 passes)", R04), ("fusion=3", {"fusion": 3}), ("head=3", {"head": 3}), ("head_tail=3", {"head_tail": 3}),
            ("fusion=3 head=3", {"fusion": 3, "head": 3}), ("fusion=3 head_tail=3", {"fusion": 3, "head_tail": 3}), ("head=3 head_tail=3", {"head": 3, "head_tail": 3}),
            ("reasm=2", {"reasm": 2}), ("fusion_in=2", {"fusion_in": 2}), ("fusion_in=3", {"fusion_in": 3}), ("proj=3", {"proj": 3}), ("attn=3", {"attn": 3}),
            ("fusion_in=3 proj=3", {"fusion_in": 3, "proj": 3})]
if os.environ.get("MDPT_BUDGET_ENCODER"):  # encoder classes one at a time (what buys margin per ms on this family)
    VARIANTS = [("shipped", {})] + [(f"{c}={n}", {c: n}) for c in ("qkv", "proj", "fc1", "fc2") for n in (2, 3)] + [("attn=3", {"attn": 3}), ("fusion_in=3", {"fusion_in": 3})]
if os.environ.get("MDPT_BUDGET_R06"):  # round 6: the fp8 table (shipped) against round 5's, and what buys margin under BOTH roundings of the fp16 weight scale
    R05 = {"reasm": 3, "fusion": 3, "fusion_proj": 3, "head": 3, "head_tail": 3}
    VARIANTS = [("shipped (fp8 cross terms)", {}), ("round 5's table (fp16 planes)", R05), ("fusion_in=4", {"fusion_in": 4}), ("fusion_in=5", {"fusion_in": 5}), ("fc1=2", {"fc1": 2}),
                ("fc2=2", {"fc2": 2}), ("fc1=2 fc2=2", {"fc1": 2, "fc2": 2}), ("fc1=3", {"fc1": 3}), ("fc2=3", {"fc2": 3}), ("qkv=2", {"qkv": 2}), ("fc2=2 fusion_in=4", {"fc2": 2, "fusion_in": 4}),
                ("fc1=3 fusion_in=4", {"fc1": 3, "fusion_in": 4}), ("fc1=3 fc2=3", {"fc1": 3, "fc2": 3}), ("fc1=3 proj=3", {"fc1": 3, "proj": 3}), ("fc1=3 qkv=3", {"fc1": 3, "qkv": 3})]
    if os.environ.get("MDPT_BUDGET_R06") == "2":
        VARIANTS = [VARIANTS[0]] + VARIANTS[-4:]
    if os.environ.get("MDPT_BUDGET_R06") == "3":  # where BEiT-L's error comes from: classes moved to 3 passes in groups
        ENC = {"qkv": 3, "attn": 3, "proj": 3, "fc1": 3, "fc2": 3}
        DEC = {"reasm": 3, "fusion": 3, "fusion_in": 3, "fusion_proj": 3, "head": 3, "head_tail": 3}
        VARIANTS = [("shipped", {}), ("attn=3", {"attn": 3}), ("qkv=3 attn=3", {"qkv": 3, "attn": 3}), ("encoder 3", ENC), ("decoder 3 (fp16 planes)", DEC),
                    ("encoder 3, decoder 3", {**ENC, **DEC}), ("fc1=3 fc2=3", {"fc1": 3, "fc2": 3}), ("fc1=3 fc2=3 proj=3", {"fc1": 3, "fc2": 3, "proj": 3}),
                    ("fc1=3 fc2=3 qkv=3", {"fc1": 3, "fc2": 3, "qkv": 3}), ("qkv=3 proj=3", {"qkv": 3, "proj": 3})]
BOTH = bool(os.environ.get("MDPT_BUDGET_R06"))
for name in (sys.argv[1:] or ["beitl", "swinl"]):
    fixture, size, batch = FIX[name]
    g = np.load(os.path.join(REPO, "tests", "golden", fixture + ".npz"))
    model, _ = bench.make_model_and_weights(name)
    # the fixture's weights (seed from the fixture, as the tests do)
    if name != "vitl":
        from muggled_dpt_amd import make_beit_dpt_from_midas_v31_state_dict, make_swinv2_dpt_from_midas_v31_state_dict
        from muggled_dpt_amd.synthetic import make_synthetic_beit_state_dict, make_synthetic_swinv2_state_dict
        make, synth = (make_beit_dpt_from_midas_v31_state_dict, make_synthetic_beit_state_dict) if name == "beitl" else (make_swinv2_dpt_from_midas_v31_state_dict, make_synthetic_swinv2_state_dict)
        _, model = make(synth(fixture, int(g["weight_seed"])))
    model = model.to("cuda", torch.float32)
    x1 = seeded_input((1, 3, size, size), int(g["input_seed"]))
    ref = torch.from_numpy(g["depth_strided"]).double()
    xb = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(1)).cuda()
    print(f"== {name} ({fixture}), batch {batch}")
    for label, passes in VARIANTS:
        model.set_precision("mixed")
        model.set_class_passes(passes)
        other = ""
        if BOTH:  # the other valid rounding of the weight scale (mdpt_debug_set_wscale_policy)
            model._debug_set_wscale_policy(True)
            d = model(x1.cuda()).cpu()[:, ::4, ::4].double() - ref
            other = f"   | every folded matrix scaled: max {float(d.abs().max() / ref.abs().max()):.3e}  rms {float(d.pow(2).mean().sqrt() / ref.abs().max()):.3e}"
            model._debug_set_wscale_policy(False)
        y = model(x1.cuda()).cpu()
        d = y[:, ::4, ::4].double() - ref
        err, rms = float(d.abs().max() / ref.abs().max()), float(d.pow(2).mean().sqrt() / ref.abs().max())
        for _ in range(2):
            model(xb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            model(xb)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8
        print(f"  {label:30s} max {err:.3e}  rms {rms:.3e}   {dt * 1e3:7.2f} ms  {batch / dt:7.1f} maps/s{other}", flush=True)

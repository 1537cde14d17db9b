#!/usr/bin/env python3
"""SwinV2-L / BEiT-L 384 fixtures: error of the encoder taps and of the depth map against the reference fixture with and without the token-mean compensation,
fp16 and mixed modes (does the compensation help the encoder of this family, and does it show in the map?)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import make_beit_dpt_from_midas_v31_state_dict, make_swinv2_dpt_from_midas_v31_state_dict
from muggled_dpt_amd.synthetic import make_synthetic_beit_state_dict, make_synthetic_swinv2_state_dict
from tests.helpers import seeded_input

gold = os.path.join(REPO, "tests", "golden")
for fixture, make, synth in (("swin2_large_384", make_swinv2_dpt_from_midas_v31_state_dict, make_synthetic_swinv2_state_dict),
                             ("beit_large_384", make_beit_dpt_from_midas_v31_state_dict, make_synthetic_beit_state_dict)):
    g = np.load(os.path.join(gold, fixture + ".npz"))
    osd = synth(fixture, int(g["weight_seed"]))
    x = seeded_input((1, 3, 384, 384), int(g["input_seed"]))
    ref = torch.from_numpy(g["depth_strided"]).double()
    _, model = make(osd)
    model = model.to("cuda", torch.float32)
    for precision in ("fp16", "mixed", "fp16x3"):
        for comp in ((None, False) if precision != "fp16x3" else (None,)):
            model.set_precision(precision)
            model.set_weight_rounding_compensation(comp)
            y = model(x.cuda()).cpu()
            d = y[:, ::4, ::4].double() - ref
            taps = model.debug_taps(1, (384, 384))
            terr = []
            for i in range(4):
                crop = torch.from_numpy(g[f"tap{i}_crop"]).double()
                t = taps["stages"][i][:, :64, :64].cpu().double()
                terr.append(float((t - crop).pow(2).mean().sqrt() / crop.pow(2).mean().sqrt()))
            print(f"{fixture:16s} {precision:7s} comp={'on ' if comp is None else 'off'} depth max {float(d.abs().max() / ref.abs().max()):.3e} rms {float(d.pow(2).mean().sqrt() / ref.abs().max()):.3e}"
                  "   tap rel rms " + " ".join(f"{e:.3e}" for e in terr), flush=True)
    del model
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""VERDICT r05 "missing" item 2: no real checkpoint exists offline, and the tolerance-meeting mode has ~20 % margin on N(0, 1/fan_in) weights.
This probe perturbs the synthetic ViT-L checkpoint toward the statistics real DINOv2 / Depth-Anything-V2 weights are known for - ALL AT ONCE,
which the two stand-in tests (small layer scales; massive activation channels) do one at a time on ViT-S:
  * layer-scale gammas log-uniform over 1e-4 ... 1 per channel (2 % of the channels at 1e-5); second variant: 1e-2 ... 3 (blocks that write
    MORE into the stream than the plain synthetic U(0.5, 1)),
  * LayerNorm weights log-normal (sigma 0.5) with 1 % of the channels x 8 / x 0.05,
  * heavy-tailed matrices: every encoder Linear's rows scaled by exp(N(0, 0.3)), one entry in 1000 x 6,
  * "massive activations": two residual channels pushed to +150 / -90 by fc2's bias in block 4, on top of +40 / -25 in the position embedding,
  * a DC offset and a 3x contrast spread across the images of the batch.
Every arithmetic mode against the CPU fp32 oracle, ViT-L at 504x504, batch 2, for a few seeds; plain synthetic weights beside it for scale.
Usage: python tools/probes/gpu_realistic_stats_check.py [seeds ...]"""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components, get_model_config_from_state_dict
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict, realistic_statistics
from oracle import dpt_oracle

torch.set_num_threads(max(1, min(32, (os.cpu_count() or 2) // 2)))
seeds = [int(a) for a in sys.argv[1:]] or [0, 1]
size = int(os.environ.get("MDPT_RS_SIZE", "504"))
for seed in seeds:
    for kind in ("plain", "realistic", "realistic, gammas 1e-2 ... 3"):
        osd = make_synthetic_original_state_dict("vitl", seed)
        if kind == "realistic":
            osd = realistic_statistics(osd, seed)
        elif kind != "plain":
            osd = realistic_statistics(osd, seed, 1e-2, 3.0)
        cfg = get_model_config_from_state_dict(osd)
        w = flatten_components(convert_state_dict_keys(cfg, osd))
        x = torch.randn(2, 3, size, size, generator=torch.Generator().manual_seed(100 + seed))
        if kind != "plain":
            x[0] = x[0] * 0.4 + 1.2
            x[1] = x[1] * 1.2 - 0.8
        t0 = time.perf_counter()
        ref = dpt_oracle.forward(w, cfg, x)
        assert bool(torch.isfinite(ref).all())
        line = [f"seed {seed} {kind:28s}: depth max {float(ref.max()):8.3f} (oracle {time.perf_counter() - t0:.0f} s)"]
        _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
        model = model.to("cuda", torch.float32)
        for prec in ("bf16x3", "fp16x3", "mixed", "fp16", "bf16"):
            model.set_precision(prec)
            y = model(x.cuda()).float().cpu()
            d = (y.double() - ref.double())
            per_img = [float(d[i].abs().max() / ref[i].double().abs().max()) for i in range(2)]
            rms = float(d.pow(2).mean().sqrt() / ref.double().abs().max())
            line.append(f"{prec} {max(per_img):.2e} (rms {rms:.1e})")
            if prec == "mixed" and os.environ.get("MDPT_RS_DETAIL"):
                for i in range(2):
                    e = d[i].abs()
                    iy, ix = divmod(int(e.argmax()), e.shape[-1])
                    print(f"      image {i}: ref min {float(ref[i].min()):.3f} max {float(ref[i].max()):.3f} mean {float(ref[i].mean()):.3f}; mixed max abs err {float(e.max()):.2e} at ({iy},{ix}) "
                          f"where ref = {float(ref[i, iy, ix]):.3f}; abs-err rms {float(d[i].pow(2).mean().sqrt()):.2e}", flush=True)
        print("   ".join(line), flush=True)
        del model

#!/usr/bin/env python3
"""Batch-1 step time (bf16, 200 steps, three interleaved rounds): default path, latency mode without the K split (threshold out of reach),
and latency mode with the K split of the residual GEMMs (two / four ranges on the 64x64 tile) from several K-tile thresholds. mdpt_debug_set_ksplit_min is the test hook that moves the threshold."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
from muggled_dpt_amd import native

for name, size in (("vitl", 504), ("vits", 504), ("beitl", 384)):
    model, _ = bench.make_model_and_weights(name)
    model = model.to("cuda", torch.bfloat16)
    x = torch.randn(1, 3, size, size, generator=torch.Generator().manual_seed(11)).to("cuda", torch.bfloat16)
    model(x)
    eng = model._get_engine()
    rows = {}
    for rnd in range(3):
        model.set_latency_mode(False)
        rows.setdefault("default", []).append(bench.time_model(model, x, 200)[0] * 1e3)
        model.set_latency_mode(True)
        OFF = 1 << 30
        for thr, big in ((OFF, OFF), (16, OFF), (16, 64), (16, 24), (16, 16), (8, 16)):
            native.check(eng.lib, eng.lib.mdpt_debug_set_ksplit_min(eng.handle, thr, big))
            label = "latency, no K split" if thr == OFF else f"latency, x2 from {thr} K tiles, x4 from {big if big < OFF else 'never'}"
            rows.setdefault(label, []).append(bench.time_model(model, x, 200)[0] * 1e3)
    for k, v in rows.items():
        print(f"{name:6s} {k:62s} " + " ".join(f"{t:.3f}" for t in v) + f"   min {min(v):.3f} ms", flush=True)
    del model

#!/usr/bin/env python3
"""Batch-1 step time in the default and in the opt-in latency mode (ViT-L / ViT-S / BEiT-L, bf16, 200 steps): for tools/probes/ab_libs.sh."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
out = []
for name, size in (("vitl", 504), ("vits", 504), ("beitl", 384)):
    model, _ = bench.make_model_and_weights(name)
    model = model.to("cuda", torch.bfloat16)
    x = torch.randn(1, 3, size, size, generator=torch.Generator().manual_seed(11)).to("cuda", torch.bfloat16)
    dt, _ = bench.time_model(model, x, 200)
    model.set_latency_mode(True)
    dl, _ = bench.time_model(model, x, 200)
    out.append(f"{name} {dt * 1e3:.3f} / latency mode {dl * 1e3:.3f} ms")
    del model
print("   ".join(out), flush=True)

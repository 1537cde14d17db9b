#!/usr/bin/env python3
"""What mdpt_set_nonfinite_propagation costs (one B-word memset + one small launch per forward): batch-1 step time of ViT-S / ViT-L and the
batch-32 ViT-L step with the switch on / off, interleaved, min of three rounds of 200 / 20 steps."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
for name, size, B, steps in (("vits", 504, 1, 200), ("vitl", 504, 1, 200), ("vitl", 504, 32, 20)):
    model, _ = bench.make_model_and_weights(name)
    model = model.to("cuda", torch.bfloat16)
    x = torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(11)).to("cuda", torch.bfloat16)
    best = {True: 1e9, False: 1e9}
    for _ in range(3):
        for on in (True, False):
            model.set_nonfinite_propagation(on)
            dt, _ = bench.time_model(model, x, steps)
            best[on] = min(best[on], dt)
    print(f"{name} {size} B={B}: on {best[True] * 1e3:.4f} ms   off {best[False] * 1e3:.4f} ms   (+{(best[True] / best[False] - 1) * 100:.2f} %)", flush=True)
    del model

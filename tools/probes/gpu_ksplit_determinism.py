#!/usr/bin/env python3
"""Run-to-run determinism stress of latency mode: N forwards of ViT-L / BEiT-L at batch 1 must all give the bits of the first one. Written for the
in-kernel K-split reduction that round 4 built and removed (profiles/r04_b1_ksplit_sweep.txt): it is what showed 1-10 % of the forwards reading
partial sums that had not arrived yet."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for name, size in (("vitl", 504), ("beitl", 384)):
    for dtype in (torch.bfloat16,):
        model, _ = bench.make_model_and_weights(name)
        model = model.to("cuda", dtype)
        model.set_latency_mode(True)
        x = torch.randn(1, 3, size, size, generator=torch.Generator().manual_seed(11)).to("cuda", dtype)
        x2 = torch.randn(1, 3, size // 2 // 112 * 112 + 112, size, generator=torch.Generator().manual_seed(12)).to("cuda", dtype)
        y0 = model(x).clone()
        bad = 0
        for i in range(n):
            if i % 7 == 3:
                model(x2)
            y = model(x)
            if not torch.equal(y, y0):
                bad += 1
                d = (y.float() - y0.float()).abs()
                print(f"  {name} {dtype} forward {i}: {int((d > 0).sum())} values differ, max {float(d.max()):.3e}", flush=True)
        print(f"{name} {dtype}: {bad} of {n} forwards differ from the first", flush=True)
        del model

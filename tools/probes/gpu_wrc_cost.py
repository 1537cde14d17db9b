#!/usr/bin/env python3
"""What the token-mean compensation's 192 small launches cost the mixed ViT-L batch-32 step WITH the two-stream batch split (the timed configuration):
the same forward with mdpt_set_weight_rounding_compensation off (its error is not the point here: an upper bound on what batching the launches can buy)."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
from muggled_dpt_amd import native
lib = native.load()
x = torch.randn(32, 3, 504, 504, generator=torch.Generator().manual_seed(1)).cuda()
for wrc in (True, False, True, False):
    for split in (8, 0):
        model, _ = bench.make_model_and_weights("vitl")
        model = model.to("cuda", torch.float32)
        model.set_precision("mixed")
        model.set_weight_rounding_compensation(wrc)
        h = model._get_engine().handle
        native.check(lib, lib.mdpt_set_batch_split(h, split))
        with torch.inference_mode():
            for _ in range(3): model(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): model(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(f"compensation {'on ' if wrc else 'off'}  split {'on ' if split else 'off'}  {dt * 1e3:7.2f} ms  {32 / dt:6.1f} maps/s", flush=True)
        del model; torch.cuda.empty_cache()

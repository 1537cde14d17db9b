#!/usr/bin/env python3
"""Headline step (ViT-L 504x504 batch 32, split batch) and SwinV2-L / BEiT-L batch 16 with the side stream in the highest (-1), default (0) and
lowest (1) priority class, fresh handle each, interleaved rounds; class 0 is measured in a process state where it does not share the caller's queue."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
from muggled_dpt_amd import native
lib = native.load()
for name, size, batch in (("vitl", 504, 32), ("swinl", 384, 16), ("beitl", 384, 16)):
    x = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(11)).to("cuda", torch.bfloat16)
    res = {}
    for rnd in range(2):
        for prio in (0, -1, 1):
            model, _ = bench.make_model_and_weights(name)
            model = model.to("cuda", torch.bfloat16)
            native.check(lib, lib.mdpt_debug_set_side_stream_priority(model._get_engine().handle, prio))
            dt, _ = bench.time_model(model, x, 20)
            res.setdefault(prio, []).append(dt * 1e3)
            del model
    print(name, "  ".join(f"class {p:2d}: {' / '.join(f'{v:.2f}' for v in vs)} ms" for p, vs in sorted(res.items())), flush=True)

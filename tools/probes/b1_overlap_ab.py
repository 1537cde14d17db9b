#!/usr/bin/env python3
"""Batch-1 step time with the reassembly branches behind the encoder (one stream) / beside it (side stream; mdpt_debug_set_reassemble_overlap),
default and latency mode, interleaved rounds on one box: ViT-S, ViT-L 504x504, BEiT-L 384x384, bf16, models built with enable_cache."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
from muggled_dpt_amd import native
lib = native.load()
for name, size in (("vits", 504), ("vitl", 504), ("beitl", 384)):
    model, _ = bench.make_model_and_weights(name, enable_cache=True)
    model = model.to("cuda", torch.bfloat16)
    x = torch.randn(1, 3, size, size, generator=torch.Generator().manual_seed(11)).to("cuda", torch.bfloat16)
    h = model._get_engine().handle
    res = {}
    for rnd in range(3):
        for latency in (False, True):
            model.set_latency_mode(latency)
            for ov in (0, 2):
                native.check(lib, lib.mdpt_debug_set_reassemble_overlap(h, ov))
                dt, _ = bench.time_model(model, x, 300)
                res.setdefault((latency, ov), []).append(dt * 1e3)
    print(name, "  ".join(f"{'latency' if l else 'default'} {'side-stream' if o else 'one-stream'} {min(v):.3f} ms" for (l, o), v in sorted(res.items())), flush=True)
    del model

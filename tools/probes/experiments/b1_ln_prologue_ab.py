#!/usr/bin/env python3
"""(Needs r05_ln_rows_in_gemm.patch applied.) Batch-1 step time with the LayerNorm of QKV / fc1 as a launch of its own / inside the GEMM launch
(MDPT_A_LNROWS; mdpt_debug_set_ln_prologue), default and latency mode, interleaved rounds on one box: ViT-S 504x504 and 252x252, bf16."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, REPO)
import bench
from muggled_dpt_amd import native
lib = native.load()
for name, size in (("vits", 504), ("vits", 252)):
    model, _ = bench.make_model_and_weights(name, enable_cache=True)
    model = model.to("cuda", torch.bfloat16)
    x = torch.randn(1, 3, size, size, generator=torch.Generator().manual_seed(11)).to("cuda", torch.bfloat16)
    h = model._get_engine().handle
    res = {}
    for rnd in range(3):
        for latency in (False, True):
            model.set_latency_mode(latency)
            for on in (0, 1):
                native.check(lib, lib.mdpt_debug_set_ln_prologue(h, on))
                dt, _ = bench.time_model(model, x, 300)
                res.setdefault((latency, on), []).append(dt * 1e3)
    print(name, size, "  ".join(f"{'latency' if l else 'default'} {'in the GEMM' if o else 'own launch'} {min(v):.3f} ms" for (l, o), v in sorted(res.items())), flush=True)
    del model

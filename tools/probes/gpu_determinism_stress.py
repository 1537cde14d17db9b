#!/usr/bin/env python3
"""Race screen for the LDS-DMA pipelines (8-phase GEMM, attention ring): N forwards of the same batch must be bit-identical, with
and without the two-stream batch split, in the bf16, 3-pass float32, mixed and fp16 modes. A DMA that is read before it landed shows up as a rare mismatch."""
import sys, os, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict, native
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
osd = make_synthetic_original_state_dict("vitl", 0)
for dtype, prec in ((torch.bfloat16, None), (torch.float32, None), (torch.float32, "mixed"), (torch.float16, None)):
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    model = model.to("cuda", dtype)
    if prec:
        model.set_precision(prec)  # (round 5: head_tail2_kernel, two-pass conv forms, the compensation's small launches between the big ones)
    x = torch.randn(32 if dtype == torch.bfloat16 else 16, 3, 504, 504, generator=torch.Generator().manual_seed(3)).to("cuda", dtype)
    with torch.inference_mode():
        ref = model(x).clone()
        bad = 0
        for split in (8, 0):
            eng = model._get_engine()
            native.check(eng.lib, eng.lib.mdpt_set_batch_split(eng.handle, split))
            for i in range(n):
                y = model(x)
                if not torch.equal(y, ref):
                    bad += 1
                    print(f"  MISMATCH dtype={dtype} split={split} iter={i}: max abs diff {float((y.float() - ref.float()).abs().max())}", flush=True)
    print(f"{dtype}{' ' + prec if prec else ''}: {2 * n} forwards, {bad} mismatches", flush=True)
    del model
    torch.cuda.empty_cache()

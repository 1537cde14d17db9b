#!/usr/bin/env python3
"""32-bit offset screen: ViT-L at batch 96 (504x504) and batch 24 at 1036x1036 must reproduce, image for image, what small batches give
(every image is independent and all tile rules are bit-compatible): bf16, the 3-pass float32 mode, and - round 5 - the mixed and fp16 modes
(head_tail2_kernel, the two-pass conv forms, the per-image bias tables)."""
import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict

osd = make_synthetic_original_state_dict("vitl", 0)
for dtype, size, big, small, prec in ((torch.bfloat16, 504, 96, 4, None), (torch.float32, 504, 48, 4, None), (torch.bfloat16, 1036, 24, 2, None),
                                      (torch.float32, 504, 96, 4, "mixed"), (torch.float16, 504, 96, 4, None), (torch.float32, 1036, 24, 2, "mixed")):
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    model = model.to("cuda", dtype)
    if prec:
        model.set_precision(prec)
    x = torch.randn(big, 3, size, size, generator=torch.Generator().manual_seed(5)).to("cuda", dtype)
    with torch.inference_mode():
        y_big = model(x)
        bad = 0
        for s in range(0, big, big // 3):
            y_small = model(x[s:s + small].contiguous())
            if not torch.equal(y_small, y_big[s:s + small]):
                bad += 1
                print(f"  MISMATCH {dtype} {size} images {s}..{s + small}: max abs diff {float((y_small.float() - y_big[s:s + small].float()).abs().max())}")
    print(f"{dtype} {prec or ''} {size}x{size} batch {big}: finite={bool(torch.isfinite(y_big.float()).all())} mismatching slices={bad} peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    del model, x, y_big
    torch.cuda.empty_cache()

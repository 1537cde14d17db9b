#!/usr/bin/env python3
"""The REFERENCE's own low-precision error on every fixture configuration (build container only: imports /root/reference).

For each model / input the golden fixtures use, run the imported reference three times on the CPU - float32, and the model and input
cast to bfloat16 and to float16 the way the reference's demos do on a GPU (`model.to(device, dtype)`, demo_helpers/misc.py:61-77,
run_image.py:158) - and record rel = max|y_lowprec - y_fp32| / max|y_fp32| of the depth map. The committed JSON
(tests/golden/reference_lowprec_errors.json) is the yardstick the GPU tests hold the bf16 / fp16 modes against instead of hand-set
tolerances: err(mode) <= max(1.25 * reference's error in that dtype, floor).

usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_reference_lowprec_errors.py [--skip-large]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
cv2_stub = types.ModuleType("cv2")
cv2_stub.COLOR_BGR2RGB = 4
cv2_stub.cvtColor = lambda img, code: np.ascontiguousarray(img[..., ::-1])
sys.modules["cv2"] = cv2_stub
sys.path.insert(0, REF)

from muggled_dpt.make_beit_dpt import make_beit_dpt_from_midas_v31_state_dict  # noqa: E402
from muggled_dpt.make_depthanythingv2_dpt import make_depthanythingv2_dpt_from_original_state_dict  # noqa: E402
from muggled_dpt.make_swinv2_dpt import make_swinv2_dpt_from_midas_v31_state_dict  # noqa: E402

from muggled_dpt_amd.synthetic import (make_synthetic_beit_state_dict, make_synthetic_original_state_dict,  # noqa: E402
                                       make_synthetic_swinv2_state_dict)

GOLD = os.path.join(REPO, "tests", "golden")


def randn(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def cases(skip_large):
    g = lambda name: np.load(os.path.join(GOLD, name))  # noqa: E731
    v2, beit, swin = make_depthanythingv2_dpt_from_original_state_dict, make_beit_dpt_from_midas_v31_state_dict, make_swinv2_dpt_from_midas_v31_state_dict
    out = [("tiny_full", v2, make_synthetic_original_state_dict("tiny", 0), torch.from_numpy(g("tiny_full.npz")["input"])),
           ("tiny_rect", v2, make_synthetic_original_state_dict("tiny", 0), torch.from_numpy(g("tiny_rect.npz")["input"])),
           ("vits504", v2, make_synthetic_original_state_dict("vits", 0), randn((1, 3, 504, 504), 1))]
    gb, gs = g("beit_tiny.npz"), g("swin2_tiny.npz")
    for tag in ("base", "wide", "tall"):
        out.append((f"beit_tiny_{tag}", beit, make_synthetic_beit_state_dict("beit_tiny", int(gb["weight_seed"])), torch.from_numpy(gb[f"{tag}_input"])))
        out.append((f"swin2_tiny_{tag}", swin, make_synthetic_swinv2_state_dict("swin2_tiny", int(gs["weight_seed"])), torch.from_numpy(gs[f"{tag}_input"])))
    if not skip_large:
        out += [("vitl504", v2, make_synthetic_original_state_dict("vitl", 0), randn((1, 3, 504, 504), 1)),
                ("beit_large_384", beit, make_synthetic_beit_state_dict("beit_large_384", 0), randn((1, 3, 384, 384), 1)),
                ("swin2_large_384", swin, make_synthetic_swinv2_state_dict("swin2_large_384", 0), randn((1, 3, 384, 384), 1))]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-large", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(max(1, (os.cpu_count() or 2)))
    path = os.path.join(GOLD, "reference_lowprec_errors.json")
    report = json.load(open(path)) if os.path.exists(path) else {}
    for name, make, osd, x in cases(args.skip_large):
        _, model = make(osd, enable_cache=False, enable_optimizations=True)
        model.eval()
        with torch.inference_mode():
            y32 = model(x).double()
            rec = {"depth_max": float(y32.abs().max())}
            for key, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
                try:
                    y = model.to(dt)(x.to(dt)).double()
                    rec[key] = float((y - y32).abs().max() / y32.abs().max())
                except Exception as e:  # an op without a CPU kernel in that dtype: say so, the tests fall back to their floor
                    rec[key] = None
                    rec[key + "_error"] = f"{type(e).__name__}: {e}"[:200]
                model.to(torch.float32)
        report[name] = rec
        print(name, rec, flush=True)
        with open(path, "w") as fh:
            json.dump(report, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

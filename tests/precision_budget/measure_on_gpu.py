"""Measured error / throughput table of the arithmetic modes on one MI355X (TEST INFRASTRUCTURE: imports the oracle as the checker).

For every policy (a precision mode of include/mdpt.h plus per-class pass counts, mdpt_set_class_passes): the error of images 0 / 7 / 13 /
31 of the seeded batch-32 input against the CPU fp32 oracle (rel = max|y - ref| / max|ref|, fp32 tensors at the boundary so that only the
operand arithmetic is measured) and the depth-maps/s of the batch-32 forward (same two-stream split as bench.py's headline).

    python tests/precision_budget/measure_on_gpu.py --out gpurun_out/precision_budget.json
    python tests/precision_budget/measure_on_gpu.py --render gpurun_out/precision_budget.json > profiles/r04_precision_budget.md
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CLASSES = ("patch", "qkv", "attn", "proj", "fc1", "fc2", "reasm", "fusion", "fusion_proj", "head", "head_tail", "fusion_in")
# round 4's mixed assignment: the whole projection path of the decoder at three passes
R04_MIXED = {"fusion": 3, "fusion_proj": 3, "head": 3, "head_tail": 3}


def policies():
    # the token-mean compensation is on by default in "mixed" and off in single-pass "fp16"; the per-class rows ("fp16c + ...") switch it on, as the
    # table the mixed assignment was derived from did. Pass counts: 3 = both operands split, 2 = activations split (weights one plane)
    out = [("bf16", "bf16", {}), ("fp16 (shipped: no compensation)", "fp16", {}), ("fp16c = fp16 + compensation", "fp16", {"wrc": True}),
           ("mixed (shipped)", "mixed", {}), ("mixed, no compensation", "mixed", {"wrc": False}),
           ("mixed of round 4 (decoder classes at 3 passes)", "mixed", dict(R04_MIXED))]
    # round 5: one decoder class at a time moved from the shipped count to 3 / 2 / 1 passes
    for c, n in (("head_tail", 3), ("head_tail", 1), ("head", 3), ("head", 1), ("fusion", 3), ("fusion", 1), ("fusion_proj", 2), ("fusion_proj", 1),
                 ("reasm", 2), ("reasm", 1), ("patch", 1), ("fusion_in", 2), ("fusion_in", 3)):
        out.append((f"mixed, {c} = {n}", "mixed", {c: n}))
    out += [("mixed, reasm + fusion_proj = 2 (whole decoder activation-split)", "mixed", {"reasm": 2, "fusion_proj": 2, "fusion_in": 2}),
            ("mixed + proj x3", "mixed", {"proj": 3}), ("mixed + proj x2", "mixed", {"proj": 2})]
    # round 6: cross terms on fp8 planes (MDPT_PASSES_2F8 = 4 / _3F8 = 5, csrc/f8_cross.h) - 1.5 / 2 pass-equivalents instead of 2 / 3
    out += [("f8: shipped table, cross terms on fp8 (reasm, fusion_proj 3F8; fusion, head 2F8)", "mixed", {"reasm": 5, "fusion_proj": 5, "fusion": 4, "head": 4}),
            ("f8: reasm, fusion_proj, fusion 3F8; head 2F8", "mixed", {"reasm": 5, "fusion_proj": 5, "fusion": 5, "head": 4}),
            ("f8: reasm, fusion_proj, fusion, head 3F8", "mixed", {"reasm": 5, "fusion_proj": 5, "fusion": 5, "head": 5}),
            ("f8: reasm, fusion_proj, fusion, head 3F8; head_tail 3", "mixed", {"reasm": 5, "fusion_proj": 5, "fusion": 5, "head": 5, "head_tail": 3}),
            ("f8: reasm, fusion_proj, fusion, head 3F8; fusion_in 2F8", "mixed", {"reasm": 5, "fusion_proj": 5, "fusion": 5, "head": 5, "fusion_in": 4}),
            ("f8: reasm, fusion_proj, fusion, head, fusion_in 3F8", "mixed", {"reasm": 5, "fusion_proj": 5, "fusion": 5, "head": 5, "fusion_in": 5}),
            ("f8: reasm, fusion_proj 3F8; fusion 2F8; head 3F8", "mixed", {"reasm": 5, "fusion_proj": 5, "fusion": 4, "head": 5})]
    # round 6: the compensation on a subset of the encoder's four Linear classes (two small launches per compensated Linear and block)
    for sub in (("qkv",), ("qkv", "proj"), ("qkv", "fc1"), ("qkv", "fc2"), ("qkv", "fc1", "fc2"), ("qkv", "proj", "fc1"), ("proj", "fc1", "fc2"), ("fc1", "fc2"), ("proj",), ("fc1",), ("fc2",)):
        out.append(("wrc: " + " + ".join(sub), "mixed", {"wrc": sub}))
    for c in CLASSES:
        out.append((f"fp16c + {c} x3", "fp16", {c: 3, "wrc": True}))
    out += [("fp16c + decoder x3", "fp16", {"reasm": 3, "fusion": 3, "fusion_proj": 3, "fusion_in": 3, "head": 3, "head_tail": 3, "wrc": True}),
            ("fp16c + encoder GEMMs x3", "fp16", {"patch": 3, "qkv": 3, "proj": 3, "fc1": 3, "fc2": 3, "wrc": True}),
            ("fp16x3", "fp16x3", {}), ("bf16x3", "bf16x3", {})]
    return out


def measure(args):
    from helpers import seeded_input, synthetic_model
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    from oracle import dpt_oracle
    osd, cfg, w = synthetic_model(args.model, 0)
    x = seeded_input((args.batch, 3, args.size, args.size), 1)
    idx = list(range(args.batch)) if args.all_images else [i for i in (0, 7, 13, 31) if i < args.batch]
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    ref = dpt_oracle.forward(w, cfg, x[idx])
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    model = model.to("cuda", torch.float32)
    if args.no_split:
        from muggled_dpt_amd import native
        _orig = model._get_engine

        def _no_split_engine():
            eng = _orig()
            native.check(eng.lib, eng.lib.mdpt_set_batch_split(eng.handle, 0))
            return eng
        model._get_engine = _no_split_engine
    xd = x.cuda()
    rows = []
    for label, prec, passes in policies():
        if args.only and not any(s in label for s in args.only):
            continue
        if args.labels and label not in args.labels:
            continue
        passes = dict(passes)
        model.set_weight_rounding_compensation(passes.pop("wrc", None))
        model.set_precision(prec)
        model.set_class_passes(passes)
        y = model(xd)
        torch.cuda.synchronize()
        errs = [float((y[i].cpu().double() - ref[k].double()).abs().max() / ref[k].double().abs().max()) for k, i in enumerate(idx)]
        rms = [float((y[i].cpu().double() - ref[k].double()).pow(2).mean().sqrt() / ref[k].double().abs().max()) for k, i in enumerate(idx)]
        for _ in range(2):
            model(xd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model(xd)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        rows.append({"label": label, "precision": prec, "passes": passes, "rel_err": errs, "rms_err": rms, "ms_per_step": dt * 1e3, "maps_per_s": args.batch / dt})
        shown = errs if len(errs) <= 4 else [max(errs), sorted(errs)[len(errs) // 2], min(errs)]  # (--all-images: worst, median, best)
        print(f"{label:58s} " + " ".join(f"{e:.2e}" for e in shown) + f" | rms {sum(rms) / len(rms):.2e}   {dt * 1e3:7.2f} ms  {args.batch / dt:7.1f} maps/s", flush=True)
    rep = {"model": args.model, "size": args.size, "batch": args.batch, "images": idx, "steps": args.steps, "device": torch.cuda.get_device_name(0), "rows": rows}
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            json.dump(rep, fh, indent=1)


def render(path):
    rep = json.load(open(path))
    print(f"# Precision budget, measured: {rep['model']} {rep['size']}x{rep['size']}, batch {rep['batch']}, one {rep['device']}\n")
    print("rel = max|y - ref| / max|ref| (mean rms: rms(y - ref) / max|ref|, mean over the images) against the CPU fp32 oracle, images " + ", ".join(map(str, rep["images"])) +
          f" of the seeded batch; fp32 tensors at the boundary; {rep['steps']} timed steps per row (`tests/precision_budget/measure_on_gpu.py`).\n")
    print("| policy | " + " | ".join(f"image {i}" for i in rep["images"]) + " | worst | mean rms | ms / step | maps/s |")
    print("|---|" + "---|" * (len(rep["images"]) + 4))
    for r in rep["rows"]:
        rms = r.get("rms_err")
        print(f"| {r['label']} | " + " | ".join(f"{e:.2e}" for e in r["rel_err"]) + f" | **{max(r['rel_err']):.2e}** | " +
              (f"{sum(rms) / len(rms):.2e}" if rms else "-") + f" | {r['ms_per_step']:.2f} | {r['maps_per_s']:.0f} |")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vitl")
    ap.add_argument("--size", type=int, default=504)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--only", nargs="*", default=[], help="rows whose label contains one of these")
    ap.add_argument("--labels", nargs="*", default=[], help="rows with exactly these labels")
    ap.add_argument("--all-images", action="store_true", help="every image of the batch against the oracle (default: images 0, 7, 13, 31)")
    ap.add_argument("--no-split", action="store_true", help="batch split off (per-kernel profiles)")
    ap.add_argument("--out", default="")
    ap.add_argument("--render", default="")
    a = ap.parse_args()
    if a.render:
        render(a.render)
    else:
        measure(a)

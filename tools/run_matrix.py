#!/usr/bin/env python3
"""Secondary measurements on one MI355X: other sizes/models/modes than bench.py's headline, hipGraph replay
for the launch-bound batch-1 case, and the H2D-inclusive rate. Writes gpurun_out/matrix.json (copied to profiles/)."""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict  # noqa: E402
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict  # noqa: E402

GFLOP = {("vitl", 504): 1224.9, ("vitl", 532): 1385.8, ("vitl", 1036): 7424.0, ("vits", 504): 107.3, ("vits", 532): 123.5, ("vits", 1036): 875.2}


def timeit(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    rows = []
    for name in ("vits", "vitl"):
        osd = make_synthetic_original_state_dict(name, 0)
        for dtype, tag in ((torch.bfloat16, "bf16"), (torch.float32, "bf16x3")):
            _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
            model = model.to("cuda", dtype)
            cases = [(1, 504), (32, 504), (32, 532), (8, 1036)] if name == "vitl" else [(1, 504), (32, 504), (8, 1036)]
            for batch, size in cases:
                if tag == "bf16x3" and (size != 504):
                    continue
                x = torch.randn(batch, 3, size, size, device="cuda", dtype=dtype)
                with torch.inference_mode():
                    sec = timeit(lambda: model(x), 10 if batch > 1 else 50)
                    row = {"model": name, "mode": tag, "batch": batch, "tensor": size, "ms": round(sec * 1e3, 3),
                           "maps_per_s": round(batch / sec, 2), "tflops": round(batch / sec * GFLOP[(name, size)] / 1e3, 1)}
                    if batch == 1:
                        # launch-bound: replay the ~215 launches from a captured hipGraph
                        g = torch.cuda.CUDAGraph()
                        s = torch.cuda.Stream()
                        s.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(s):
                            model(x)
                        torch.cuda.current_stream().wait_stream(s)
                        with torch.cuda.graph(g):
                            yg = model(x)
                        sec_g = timeit(g.replay, 100)
                        ok = bool(torch.equal(yg, model(x)))
                        row.update({"graph_ms": round(sec_g * 1e3, 3), "graph_maps_per_s": round(1 / sec_g, 2), "graph_bitwise_equal": ok})
                    if batch == 32 and size == 504 and tag == "bf16":
                        xh = torch.randn(batch, 3, size, size, dtype=dtype).pin_memory()
                        sec_h = timeit(lambda: model(xh.to("cuda", non_blocking=True)), 10)
                        row.update({"h2d_inclusive_ms": round(sec_h * 1e3, 3), "h2d_inclusive_maps_per_s": round(batch / sec_h, 2)})
                print(json.dumps(row), flush=True)
                rows.append(row)
                del x
            del model
            torch.cuda.empty_cache()
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(REPO, "gpurun_out", "matrix.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

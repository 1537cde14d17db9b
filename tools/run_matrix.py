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

# algorithmic GFLOP per map of the reference graph
GFLOP_CFG5 = {"beit_large_384": 516.4, "swin2_large_384": 343.7}
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


def kernel_shares(fn, steps=3):
    """Per-kernel time share of `fn` from the library's HIP-event profiler."""
    import ctypes
    from muggled_dpt_amd import native
    lib = native.load()
    torch.cuda.synchronize()
    lib.mdpt_profile_enable(1)
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    buf = ctypes.create_string_buffer(1 << 16)
    out = {}
    if lib.mdpt_profile_report(buf, len(buf)) == 0:
        prof = json.loads(buf.value.decode())
        tot = sum(k["total_ms"] for k in prof["kernels"]) or 1.0
        out = {k["name"]: [round(k["total_ms"] / tot, 4), round(k["total_ms"] / steps, 3)] for k in prof["kernels"][:10]}
    lib.mdpt_profile_enable(0)
    return out


def config5(rows):
    """BASELINE.json configs[5]: MiDaS v3.1 BEiT-L-384 and SwinV2-L-384, randn(16,3,384,384), one MI355X."""
    from muggled_dpt_amd import make_beit_dpt_from_midas_v31_state_dict, make_swinv2_dpt_from_midas_v31_state_dict
    from muggled_dpt_amd.synthetic import make_synthetic_beit_state_dict, make_synthetic_swinv2_state_dict
    for name, make_sd, make_model in (("beit_large_384", make_synthetic_beit_state_dict, make_beit_dpt_from_midas_v31_state_dict),
                                      ("swin2_large_384", make_synthetic_swinv2_state_dict, make_swinv2_dpt_from_midas_v31_state_dict)):
        osd = make_sd(name, 0)
        for dtype, tag in ((torch.bfloat16, "bf16"), (torch.float32, "bf16x3")):
            _, model = make_model(osd)
            model = model.to("cuda", dtype)
            for batch in (1, 16):
                x = torch.randn(batch, 3, 384, 384, device="cuda", dtype=dtype)
                with torch.inference_mode():
                    sec = timeit(lambda: model(x), 10 if batch > 1 else 30)
                    row = {"model": name, "mode": tag, "batch": batch, "tensor": 384, "ms": round(sec * 1e3, 3),
                           "maps_per_s": round(batch / sec, 2), "tflops": round(batch / sec * GFLOP_CFG5[name] / 1e3, 1)}
                    if batch == 16:
                        row["kernel_share_and_ms"] = kernel_shares(lambda: model(x))
                print(json.dumps(row), flush=True)
                rows.append(row)
                del x
            del model
            torch.cuda.empty_cache()


def main():
    rows = []
    if "--config5" in sys.argv:
        config5(rows)
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        json.dump(rows, open(os.path.join(REPO, "gpurun_out", "matrix_config5.json"), "w"), indent=1)
        return
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
                        # opt-in latency mode (mdpt_set_latency_mode): summation orders that are not batch-invariant in the last bit
                        model.set_latency_mode(True)
                        sec_l = timeit(lambda: model(x), 50)
                        model.set_latency_mode(False)
                        row.update({"latency_mode_ms": round(sec_l * 1e3, 3), "latency_mode_maps_per_s": round(1 / sec_l, 2)})
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

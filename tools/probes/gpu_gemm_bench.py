#!/usr/bin/env python3
"""GEMM kernel micro-benchmark on the encoder's real shapes (run on the MI355X box). Prints TFLOP/s per (shape, tile)
and checks the result against torch.matmul. usage: python tools/probes/gpu_gemm_bench.py [--iters 20]"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from muggled_dpt_amd import native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tiles", default="1,2,5")
    args = ap.parse_args()
    lib = native.load()
    stream = torch.cuda.current_stream().cuda_stream
    shapes = [(41728, 1024, 1024, "proj"), (41728, 3072, 1024, "qkv-shape"), (41728, 4096, 1024, "fc1"), (41728, 1024, 4096, "fc2"),
              (8192, 8192, 8192, "square8k"), (1304, 1024, 1024, "B=1 proj")]
    res = {}
    for M, N, K, tag in shapes:
        a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        w = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
        out = torch.empty(M, N, device="cuda", dtype=torch.float32)
        ref = (a[:512].float() @ w.float().t())
        for tile in [int(t) for t in args.tiles.split(",")]:
            native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), out.data_ptr(), None, M, N, K, tile, 2, stream, None))
            torch.cuda.synchronize()
            err = float((out[:512] - ref).abs().max() / ref.abs().max())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), out.data_ptr(), None, M, N, K, tile, args.iters, stream, None))
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            tf = 2.0 * M * N * K / us * 1e-6
            res[f"{tag} {M}x{N}x{K} tile{tile}"] = {"us": round(us, 1), "tflops": round(tf, 1), "rel_err": err}
            print(f"{tag:10s} M={M} N={N} K={K} tile={tile}: {us:9.1f} us  {tf:7.1f} TF  err {err:.1e}", flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(REPO, "gpurun_out", "gemm_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: depth-maps/sec of the Depth-Anything-V2 ViT-L DPT path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

One "step" = one pass of the hot path (mdpt_forward through the C ABI) over one batch of synthetic images per GPU,
followed (N>1) by the RCCL all-gather of the depth maps. Workload = BASELINE.json configs[2]/[3]:
ViT-L, "518x518" images (= 504x504 model tensor, the reference's own size snapping), bf16 MFMA with fp32 accumulate,
batch 32 per GPU (weak scaling: global batch 32*N), seeded synthetic weights and inputs already resident in HBM.
Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict, native  # noqa: E402
from muggled_dpt_amd.parallel import DataParallelDepth, init_distributed  # noqa: E402
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict  # noqa: E402

# algorithmic GFLOP per depth map (2 FLOP/MAC), SURVEY §8(d) / BASELINE.md §4
GFLOP_PER_MAP = {("vitl", 504): 1224.9, ("vitl", 532): 1385.8, ("vitl", 1036): 7424.0,
                 ("vits", 504): 107.3, ("vits", 532): 123.5, ("vits", 1036): 875.2}
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)


def cpu_baseline(model_name: str, size: int, x_cpu: torch.Tensor, gpu_out: torch.Tensor | None):
    """Oracle (CPU restatement of the reference, kind="port") timed on this box's host cores on a bounded sample."""
    from oracle import dpt_oracle
    from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components, get_model_config_from_state_dict

    threads = max(1, (os.cpu_count() or 2) // 2)  # the reference's own policy (demo_helpers/misc.py:161-166)
    torch.set_num_threads(threads)
    osd = make_synthetic_original_state_dict(model_name, 0)
    cfg = get_model_config_from_state_dict(osd)
    w = flatten_components(convert_state_dict_keys(cfg, osd))
    n_img = 2 if model_name == "vitl" else 8
    dpt_oracle.forward(w, cfg, x_cpu[:1])  # warm-up (thread pool, oneDNN primitive caches)
    t0 = time.perf_counter()
    ref = None
    for i in range(n_img):
        y = dpt_oracle.forward(w, cfg, x_cpu[i % x_cpu.shape[0]: i % x_cpu.shape[0] + 1])
        if i == 0:
            ref = y
    dt = time.perf_counter() - t0
    out = {"value": round(n_img / dt, 4), "unit": "depth-maps/s", "cores": threads, "kind": "port",
           "sample": f"{n_img} images of the same workload, batch 1, fp32, torch {torch.__version__} CPU, {threads} threads "
                     f"(os.cpu_count()={os.cpu_count()})"}
    return out, error_vs(ref, gpu_out), ref


def error_vs(ref: torch.Tensor, gpu_out: torch.Tensor | None):
    """The north-star error figure of image 0: max|y - y_cpu| and the same relative to max|y_cpu|."""
    if gpu_out is None:
        return None
    d = (gpu_out[:1].double().cpu() - ref.double()).abs().max()
    return {"max_abs": float(d), "rel_to_max": float(d / ref.double().abs().max())}


def fp32_class_leg(args, dev, x_cpu, ref):
    """The same workload in the bf16x3 mode (hi + lo bf16 operand planes, 3 MFMA passes, fp32 accumulate): the mode that meets the
    north star's 1e-3 relative tolerance. Reported beside the headline bf16 number, never instead of it."""
    osd = make_synthetic_original_state_dict(args.model, 0)
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    del osd
    model = model.to(dev, torch.float32)
    x = x_cpu.to(dev)
    steps = max(2, min(args.steps, 5))
    with torch.inference_mode():
        y = model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            y = model(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"value": round(args.batch * steps / dt, 3), "unit": "depth-maps/s", "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
            "dtype": "bf16x3 (hi/lo split bf16 MFMA operands, 3 passes, fp32 accumulate)", "error_vs_cpu_fp32": error_vs(ref, y.float())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="vitl", choices=["vitl", "vits"])
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=504, help="model tensor side (a 518x518 image is processed at 504)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3"])
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-launch HIP events in the timed region")
    ap.add_argument("--no-split", action="store_true", help="disable the two-stream half-batch split inside mdpt_forward")
    args = ap.parse_args()

    rank, world, local_rank = init_distributed()
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:  # one process per node checks / builds libmdpt.so, the others load the finished file
        if local_rank == 0:
            native.load()
        dist.barrier()
    dtype = torch.bfloat16 if args.precision == "bf16" else torch.float32

    osd = make_synthetic_original_state_dict(args.model, 0)
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    del osd
    model = model.to(dev, dtype)
    if args.tile:
        model.set_gemm_tile(args.tile)
    if args.no_split:
        from muggled_dpt_amd import native as _native
        _native.check(_native.load(), _native.load().mdpt_set_batch_split(model._get_engine().handle, 0))
    x_cpu = torch.randn(args.batch, 3, args.size, args.size, generator=torch.Generator().manual_seed(1 + rank))
    x = x_cpu.to(dev).to(dtype)
    dp = DataParallelDepth(model, rank, world)
    lib = native.load()

    def barrier():
        if world > 1:
            dist.barrier()

    with torch.inference_mode():
        for _ in range(args.warmup):
            y = dp.forward_shard(x)
        torch.cuda.synchronize()
        barrier()
        if not args.no_profile:
            lib.mdpt_profile_enable(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = dp.forward_shard(x)
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
    prof = prof_alone = None
    if not args.no_profile:
        buf = ctypes.create_string_buffer(1 << 16)
        if lib.mdpt_profile_report(buf, len(buf)) == 0:
            prof = json.loads(buf.value.decode())
        lib.mdpt_profile_enable(0)
        if not args.no_split and rank == 0:
            # The timed region above ran mdpt_forward's default two-stream half-batch split: its kernels overlap pairwise, so a
            # launch's begin->end duration includes the time it shared the GPU with the other half's kernel. For the per-kernel
            # roofline, time the same steps once more with the split off (every kernel alone on the GPU, same stream events).
            handle = model._get_engine().handle
            native.check(lib, lib.mdpt_set_batch_split(handle, 0))
            with torch.inference_mode():  # local forward only (no collective: the other ranks are not in this pass)
                model(x)
                torch.cuda.synchronize()
                lib.mdpt_profile_enable(1)
                for _ in range(args.steps):
                    model(x)
                torch.cuda.synchronize()
            buf = ctypes.create_string_buffer(1 << 16)
            if lib.mdpt_profile_report(buf, len(buf)) == 0:
                prof_alone = json.loads(buf.value.decode())
            lib.mdpt_profile_enable(0)
            native.check(lib, lib.mdpt_set_batch_split(handle, 8))
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        maps = world * args.batch * args.steps
        value = maps / elapsed
        gflop = GFLOP_PER_MAP.get((args.model, args.size))
        line = {
            "metric": "depth-maps/sec @518x518 (504x504 model tensor), DA-V2 %s" % {"vitl": "ViT-L", "vits": "ViT-S"}[args.model],
            "value": round(value, 3), "unit": "depth-maps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "bf16x3(fp32-class)", "data": "synthetic",
            "config": {"workload": f"Depth-Anything-V2 {args.model}, 518x518 image -> {args.size}x{args.size} tensor, batch "
                                   f"{args.batch}/GPU, {args.precision} MFMA operands + fp32 accumulate, mdpt_forward via C ABI"
                                   + (", RCCL all-gather of depth maps" if world > 1 else ""),
                       "global_batch": world * args.batch, "tensor_hw": [args.size, args.size], "parallelism": f"dp{world}",
                       "gemm_tile": args.tile, "batch_split": not args.no_split},
        }
        if gflop:
            line["path_tflops"] = round(value * gflop / 1e3, 2)
            line["path_frac_of_mfma_peak"] = round(value * gflop / 1e3 / (PEAK_BF16_TFLOPS * world), 4)
        if prof and prof["kernels"]:
            def hbm_traffic(kernel_name):
                """HBM bytes per launch of `kernel_name` from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this
                exact workload, corrected per MI355X_MICROARCH.md: profiles/r01_hbm_traffic.md). bench.py cannot run PMC passes itself."""
                path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_hbm_traffic.json")
                if args.model != "vitl" or args.batch != 32 or args.size != 504 or args.precision != "bf16" or not os.path.exists(path):
                    return None
                rec = json.load(open(path)).get(kernel_name)
                return None if rec is None else int((rec["fetch_mb_per_launch"] + rec["write_mb_per_launch"]) * 1e6)

            def roof(pr, how):
                gemms = [k for k in pr["kernels"] if k["gflop"] > 0 and k["name"].startswith("gemm")]
                dom = max(gemms, key=lambda k: k["total_ms"]) if gemms else pr["kernels"][0]
                return {"bound": "mfma", "achieved": dom["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(dom["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": hbm_traffic(dom["name"]), "kernel": dom["name"],
                        "launches": dom["launches"], "avg_us": dom["avg_us"],
                        "gflop_per_launch": round(dom["gflop"] / dom["launches"], 3), "measured": how}
            if prof_alone and prof_alone["kernels"]:
                line["roofline"] = roof(prof_alone, "HIP events, second pass of the same steps with the batch split off (kernel alone on the GPU; "
                                                    "`bench.py --no-split` + rocprofv3 reproduce it)")
                line["roofline_in_timed_region"] = roof(prof, "HIP events in the timed region: two half-batch kernels overlap, durations include sharing")
            else:
                line["roofline"] = roof(prof, "HIP events in the timed region")
            tot = sum(k["total_ms"] for k in prof["kernels"])
            line["kernel_time_share"] = {k["name"]: round(k["total_ms"] / tot, 4) for k in prof["kernels"][:12]}
        else:
            line["roofline"] = None
        if world == 1 and not args.no_cpu_baseline:
            base, err, ref = cpu_baseline(args.model, args.size, x_cpu, y.float())
            line["cpu_baseline"] = base
            line["error_vs_cpu_fp32"] = err
            if args.precision == "bf16":
                del y
                line["fp32_class_mode"] = fp32_class_leg(args, dev, x_cpu, ref)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""CPU reference-path baseline on the GPU box's host cores (SURVEY §8(d) "CPU baseline timed beside it").

Runs the CPU oracle (oracle/dpt_oracle.py: the same op sequence as the reference, fp32, torch.inference_mode) on synthetic
weights/inputs: ViT-S and ViT-L at 504x504, batch 1 and 8, with n = os.cpu_count()//2 threads (the reference's own policy,
demo_helpers/misc.py:161-166) and n = os.cpu_count()//4. Prints one JSON document (also written to gpurun_out/cpu_baseline.json).
The oracle is test infrastructure: this tool times it, nothing in the product path uses it."""
import json
import os
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components, get_model_config_from_state_dict  # noqa: E402
from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict  # noqa: E402
from oracle import dpt_oracle  # noqa: E402


def cpu_model() -> str:
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        for line in out.splitlines():
            if line.startswith("Model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def main():
    rows = []
    ncpu = os.cpu_count() or 1
    for name in ("vits", "vitl"):
        osd = make_synthetic_original_state_dict(name, 0)
        cfg = get_model_config_from_state_dict(osd)
        w = flatten_components(convert_state_dict_keys(cfg, osd))
        del osd
        for threads in (max(1, ncpu // 4), max(1, ncpu // 2)):  # all logical CPUs oversubscribes the box (measured 70x slower)
            torch.set_num_threads(threads)
            for batch in (1, 8):
                x = torch.randn(batch, 3, 504, 504, generator=torch.Generator().manual_seed(1))
                warm, runs = (2, 4) if name == "vits" else (1, 2)
                for _ in range(warm):
                    dpt_oracle.forward(w, cfg, x)
                t0 = time.perf_counter()
                for _ in range(runs):
                    dpt_oracle.forward(w, cfg, x)
                sec = (time.perf_counter() - t0) / runs
                row = {"model": name, "batch": batch, "threads": threads, "ms": round(sec * 1e3, 1), "maps_per_s": round(batch / sec, 3)}
                print(json.dumps(row), flush=True)
                rows.append(row)
    doc = {"cpu": cpu_model(), "os_cpu_count": ncpu, "torch": torch.__version__, "rows": rows}
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(doc, open(os.path.join(REPO, "gpurun_out", "cpu_baseline.json"), "w"), indent=1)
    print(json.dumps({"cpu": doc["cpu"], "os_cpu_count": ncpu}))


if __name__ == "__main__":
    main()

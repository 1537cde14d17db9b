"""Host-side tooling that produces the committed profile summaries: tools/summarize_pmc.py on a synthetic rocprofv3 counter CSV."""
import csv
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_pass(folder, counter, rows):
    os.makedirs(os.path.join(folder, "node"), exist_ok=True)
    with open(os.path.join(folder, "node", "1_counter_collection.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writeheader()
        for disp, kernel, value in rows:
            w.writerow({"Dispatch_Id": disp, "Kernel_Name": kernel, "Counter_Name": counter, "Counter_Value": value})


def test_summarize_pmc_units_and_gfx950_fetch_correction(tmp_path):
    k1 = "void (anonymous namespace)::layernorm_kernel<4>(float const*, float const*, int)"
    k2 = "void (anonymous namespace)::gemm8_kernel<0, 0, 2>(GemmParams)"
    # two dispatches of k1 with several per-XCD rows each (rocprofv3 emits one row per counter instance), one of k2
    _write_pass(tmp_path / "f", "FETCH_SIZE", [(1, k1, 40000), (1, k1, 43000), (2, k1, 83000), (3, k2, 250000)])
    _write_pass(tmp_path / "w", "WRITE_SIZE", [(1, k1, 83000), (2, k1, 84000), (3, k2, 167000)])
    out = str(tmp_path / "traffic")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "summarize_pmc.py"), str(tmp_path / "f"), str(tmp_path / "w"), out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = json.load(open(out + ".json"))
    ln = rows["layernorm_kernel<4>"]
    assert ln["launches"] == 2
    assert ln["fetch_mb_per_launch"] == round(2 * 83000 * 1024 / 1e6, 1)  # KiB -> bytes, doubled on gfx950
    assert ln["write_mb_per_launch"] == round(83500 * 1024 / 1e6, 1)
    g = rows["gemm8_kernel<0, 0, 2>"]
    assert g["launches"] == 1 and g["fetch_mb_per_launch"] == 512.0 and g["write_mb_per_launch"] == 171.0
    assert list(rows)[0] == "gemm8_kernel<0, 0, 2>" or list(rows)[0] == "layernorm_kernel<4>"  # sorted by total traffic
    assert "| `layernorm_kernel<4>` | 2 |" in open(out + ".md").read()

#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) -> profiles/<tag>_hbm_traffic.{json,md}.

Usage: summarize_pmc.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out prefix>
Units and corrections as MI355X_MICROARCH.md prescribes: both counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of a
wide coalesced read stream as 64 B and is doubled, WRITE_SIZE is taken as is. Calibration line: layernorm_kernel reads rows x F fp32 and
writes rows x F bf16."""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name).strip()


def per_kernel(folder: str, counter: str):
    files = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no counter_collection.csv under {folder}")
    tot, disp = defaultdict(float), defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return {k: (tot[k] / len(disp[k]), len(disp[k])) for k in tot}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    rows = {}
    for k in fetch:
        f_kib, n = fetch[k]
        w_kib = write.get(k, (0.0, n))[0]
        rows[k] = {"launches": n, "fetch_mb_per_launch": round(2 * f_kib * 1024 / 1e6, 1), "write_mb_per_launch": round(w_kib * 1024 / 1e6, 1)}
    rows = dict(sorted(rows.items(), key=lambda kv: -(kv[1]["fetch_mb_per_launch"] + kv[1]["write_mb_per_launch"]) * kv[1]["launches"]))
    try:  # stamp the kernel sources the passes were taken on: bench.py drops `roofline.traffic` when they have changed since
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from muggled_dpt_amd import native
        meta = {"csrc_sha": native.source_hash(), "workload": os.environ.get("MDPT_PMC_WORKLOAD", "vitl/504/32/bf16")}
    except Exception as e:  # noqa: BLE001
        meta = {"csrc_sha": None, "error": str(e)}
    json.dump({**rows, "_meta": meta}, open(out + ".json", "w"), indent=1)
    with open(out + ".md", "w") as fh:
        fh.write("| kernel | launches | FETCH_SIZE x2 (MB/launch) | WRITE_SIZE (MB/launch) |\n|---|---|---|---|\n")
        for k, r in rows.items():
            fh.write(f"| `{k}` | {r['launches']} | {r['fetch_mb_per_launch']} | {r['write_mb_per_launch']} |\n")
    print(open(out + ".md").read())


if __name__ == "__main__":
    main()

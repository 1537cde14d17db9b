#!/usr/bin/env python3
"""Per-kernel SQ wave-state shares from one rocprofv3 PMC pass -> markdown table.
Usage: summarize_sq.py <dir of the pass> <out.md>
Counters: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS.
Shares are fractions of SQ_WAVE_CYCLES: wait = parked on s_waitcnt / barrier, stall = issue stall (of which on LDS), active = issuing,
valu = VALU issue; mfma = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES), a per-wave-cycle MFMA occupancy proxy (not the per-SIMD MfmaUtil)."""
import csv, glob, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_pmc import short  # noqa: E402


def main():
    folder, out = sys.argv[1:3]
    files = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        sys.exit(f"no counter_collection.csv under {folder}")
    tot, disp = defaultdict(lambda: defaultdict(float)), defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    rows = []
    for k, c in tot.items():
        w = c.get("SQ_WAVE_CYCLES", 0.0)
        if w <= 0:
            continue
        rows.append((w, k, len(disp[k]), c.get("SQ_WAIT_ANY", 0) / w, c.get("SQ_WAIT_INST_ANY", 0) / w, c.get("SQ_WAIT_INST_LDS", 0) / w,
                     c.get("SQ_ACTIVE_INST_ANY", 0) / w, c.get("SQ_ACTIVE_INST_VALU", 0) / w, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * w),
                     c.get("SQ_LDS_BANK_CONFLICT", 0)))
    rows.sort(reverse=True)
    with open(out, "w") as fh:
        fh.write(__doc__.split("Usage")[0].strip() + "\n\n" + "\n".join(__doc__.splitlines()[3:]) + "\n\n")
        fh.write("| kernel | launches | wait | stall | (stall on LDS) | active | valu | mfma | LDS bank-conflict cycles |\n|---|---|---|---|---|---|---|---|---|\n")
        for _, k, n, wait, stall, stall_lds, act, valu, mfma, bank in rows[:14]:
            fh.write(f"| `{k}` | {n} | {wait:.2f} | {stall:.2f} | {stall_lds:.2f} | {act:.2f} | {valu:.2f} | {mfma:.3f} | {bank / 1e6:.1f}M |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()

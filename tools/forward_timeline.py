#!/usr/bin/env python3
"""One forward pass as a launch-by-launch timeline, from a rocprofv3 --kernel-trace CSV.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python bench.py --no-split --steps 2 --warmup 1 --no-cpu-baseline --no-profile
    python tools/forward_timeline.py DIR out.md

The last forward in the trace (from its patchify launch on) is printed in launch order: duration, gap to the previous launch, grid in
workgroups, and consecutive launches of one kernel at one grid collapsed into a row. It is what shows which decoder launches
underfill the 256 CUs (grid < 256 workgroups) or pay a round-quantisation tail."""
import csv, glob, os, re, sys


def short(name):
    name = re.sub(r"^(void )?\(anonymous namespace\)::", "", name.strip().strip('"'))
    name = re.sub(r"\(.*$", "", name)
    return name


def main():
    src, out = sys.argv[1], sys.argv[2]
    files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit(f"no kernel_trace.csv under {src}")
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), grid // max(wg, 1), wg))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("patchify_kernel")]
    if not starts:
        sys.exit("no patchify launch in the trace; kernels seen: " + ", ".join(sorted({r[2] for r in rows})))
    fwd = rows[starts[-1]:]
    t0, t_end = fwd[0][0], fwd[-1][1]
    lines = [f"One forward, {len(fwd)} launches, {(t_end - t0) / 1e3:.1f} us from the first launch's start to the last launch's end "
             f"(sum of kernel durations {sum(r[1] - r[0] for r in fwd) / 1e3:.1f} us).", "",
             "| # | kernel | launches | workgroups | threads | avg us | total us | avg gap before (us) |", "|---|---|---|---|---|---|---|---|"]
    i, prev_end, idx = 0, fwd[0][0], 0
    while i < len(fwd):
        j, dur, gap = i, 0, 0
        while j < len(fwd) and fwd[j][2] == fwd[i][2] and fwd[j][3] == fwd[i][3]:
            dur += fwd[j][1] - fwd[j][0]
            gap += max(0, fwd[j][0] - prev_end)
            prev_end = fwd[j][1]
            j += 1
        n = j - i
        lines.append(f"| {idx} | `{fwd[i][2]}` | {n} | {fwd[i][3]} | {fwd[i][4]} | {dur / n / 1e3:.1f} | {dur / 1e3:.1f} | {gap / n / 1e3:.2f} |")
        idx += 1
        i = j
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""gpurun_out/parity_report.json (written by tests/conftest.py at the end of a `pytest -m gpu` session: the largest relative error
every test recorded through tests/helpers.rel_err) -> a markdown table, grouped by test file, worst first.

    python tools/summarize_parity.py gpurun_out/parity_report.json profiles/r02_parity_report.md"""
import json, sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    rec = json.load(open(src))
    by_file = {}
    for name, err in rec.items():
        f, _, t = name.partition("::")
        by_file.setdefault(f, []).append((float(err), t))
    lines = ["Largest relative error (max|GPU - reference| / max|reference|) recorded by every GPU parity test of the last full",
             "`python -m pytest tests -m gpu` session. `dtype0` = float32 model = split-bf16 (x3) arithmetic, `dtype1` = bfloat16 model;",
             "the number after the dtype in a test id is the tolerance that test asserts.", ""]
    for f in sorted(by_file):
        lines += [f"### {f}", "", "| test | max rel. error |", "|---|---|"]
        for err, t in sorted(by_file[f], reverse=True):
            lines.append(f"| `{t}` | {err:.3g} |")
        lines.append("")
    open(out, "w").write("\n".join(lines))
    print(f"{len(rec)} records -> {out}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""DESIGN.md's "Numbers of this round" table from one collection pass: reads <dir>/<prefix>bench_*.json (bench.py lines written by
tools/collect_profiles.sh) and prints the markdown rows. Usage: python tools/summarize_numbers.py profiles r06_"""
import json
import os
import sys


def line(path):
    with open(path) as fh:
        rows = [ln for ln in fh.read().strip().splitlines() if ln.startswith("{")]
    return json.loads(rows[-1])


def err(d):
    e = (d or {}).get("error_vs_cpu_fp32") or {}
    return f"{e['rel_to_max']:.1e}" if "rel_to_max" in e else "—"


def main():
    d0, pre = sys.argv[1], sys.argv[2]
    g = lambda name: line(os.path.join(d0, f"{pre}bench_{name}.json"))
    n1 = g("n1")
    rf = n1["roofline"]
    tr = rf.get("traffic")
    tr_s = f", {tr / 1e6:.1f} MB HBM per launch (PMC)" if isinstance(tr, (int, float)) else (f", traffic {json.dumps(tr)[:80]}" if tr else "")
    print("| configuration | value | roofline kernel frac (path) | error vs CPU fp32 oracle |")
    print("|---|---|---|---|")
    print(f"| ViT-L 504² B=32 bf16 (**headline**) | {n1['value']:.1f} maps/s ({n1['ms_per_step']:.2f} ms) | {rf['frac']:.3f} ({n1['path_frac_of_mfma_peak']:.3f}); "
          f"dominant kernel {rf['avg_us']:.1f} µs alone{tr_s} | {err(n1)} |")
    for key, label in (("mixed_mode", "same, mixed"), ("fp16_mode", "same, fp16"), ("fp32_class_mode", "same, bf16x3")):
        m = n1[key]
        r = (m.get("roofline") or {}).get("frac")
        print(f"| {label} | {m['value']:.1f} ({m['ms_per_step']:.2f} ms) | {r if r is None else format(r, '.3f')} | {err(m)} |")
    sec = n1.get("secondary", {})
    for key, label in (("vitl_1036_b8", "ViT-L 1036² B=8"), ("beitl_384_b16", "BEiT-L 384² B=16"), ("swinl_384_b16", "SwinV2-L 384² B=16")):
        s = sec.get(key)
        if not s:
            continue
        mm = s.get("mixed_mode", {})
        print(f"| {label} bf16 / mixed | {s['value']:.1f} / {mm.get('value', float('nan')):.1f} | {s['roofline']['frac']:.3f} ({s['path_frac_of_mfma_peak']:.3f}) | {err(s)} / {err(mm)} |")
    for key, label in (("vits_504_b1", "ViT-S 504² B=1 bf16 (configs[1])"), ("vitl_504_b1", "ViT-L 504² B=1 bf16")):
        s = sec.get(key)
        if not s:
            continue
        lat, inf = s.get("latency_mode", {}), s.get("inference_b1", {})
        print(f"| {label} | {s['ms_per_step']:.3f} ms ({lat.get('ms_per_step', float('nan')):.3f} latency mode); `inference()` {inf.get('ms_per_call_sync', float('nan')):.3f} ms sync, "
              f"{inf.get('ms_per_call_pipelined', float('nan')):.3f} pipelined | {s['path_frac_of_mfma_peak']:.3f} (path) | {err(s)} |")
    cb = n1.get("cpu_baseline", {})
    print(f"| CPU oracle (kind \"{cb.get('kind')}\"), {cb.get('cpu')} | {cb.get('value')} maps/s on {cb.get('cores')} threads ({cb.get('policy_value')} at the reference's "
          f"{cb.get('policy_cores')}-thread policy) | — | — |")
    for name in ("mixed", "fp16", "x3", "1036", "beitl", "swinl"):
        try:
            d = g(name)
        except OSError:
            continue
        print(f"<!-- own run `{name}`: {d['value']:.1f} maps/s, {d['ms_per_step']:.2f} ms, roofline {d['roofline']['frac']:.3f}, err {err(d)} -->")


if __name__ == "__main__":
    main()

"""Print `value ms_per_step [dominant kernel avg_us frac]` of one bench.py run (for tools/probes/ab_libs.sh):
     bash tools/probes/ab_libs.sh 3 python tools/probes/bench_value.py --steps 10 --warmup 3"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-secondary", "--no-cpu-baseline", *sys.argv[1:]],
                     capture_output=True, text=True, cwd=root)
line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
if not line:
    sys.exit(out.stdout[-2000:] + out.stderr[-2000:])
d = json.loads(line[-1])
r = d.get("roofline") or {}
print(d["value"], d["ms_per_step"], r.get("kernel"), r.get("avg_us"), r.get("frac"))

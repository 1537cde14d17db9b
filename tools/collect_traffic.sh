#!/bin/bash
# Runs ON the MI355X box: only the two PMC passes whose summary bench.py's roofline.traffic reads (FETCH_SIZE, WRITE_SIZE; separate runs,
# --kernel-trace only) - for a source change after the full collection (tools/collect_profiles.sh). Output: gpurun_out/traffic_<tag>/hbm_traffic.{json,md}
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/traffic_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -- python "$R/bench.py" --no-split --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2> "$OUT/pmc_$C.log"
done
python "$R/tools/summarize_pmc.py" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/hbm_traffic" > "$OUT/hbm_traffic.log" 2>&1
rm -rf "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
ls -la "$OUT"

#!/bin/bash
# tools/probes/mfma_power.hip with rocm-smi's power / sclk readings sampled next to it (one line per ~0.25 s, tagged with the time)
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$R/gpurun_out"
( while true; do echo "t=$(date +%s.%N | cut -c1-14) $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk' | tr -s ' ' | tr '\n' '|')"; sleep 0.25; done ) > "$R/gpurun_out/mfma_power_smi.txt" &
SMI=$!
"$R/tools/probes/_bin/mfma_power" | while IFS= read -r line; do echo "t=$(date +%s.%N | cut -c1-14) $line"; done
kill $SMI

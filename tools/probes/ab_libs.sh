#!/bin/bash
# A/B of two builds of libmdpt.so on ONE box (box-to-box spread is +-3 %): tools/probes/_bin/libmdpt_A.so and libmdpt_B.so are copied over
# the in-tree library in turn, interleaved, and the given probe command is run with each.   bash tools/probes/ab_libs.sh 3 python tools/probes/gpu_attn_bench.py
R=$(cd "$(dirname "$0")/../.." && pwd)
N=$1; shift
for i in $(seq 1 "$N"); do
  for v in A B; do
    cp "$R/tools/probes/_bin/libmdpt_$v.so" "$R/muggled_dpt_amd/csrc/libmdpt.so"
    echo "== $v round $i"
    "$@" 2>&1 | grep -v amdgpu.ids
  done
done

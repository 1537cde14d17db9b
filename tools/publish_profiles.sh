#!/bin/bash
# gpurun_out/profiles_<tag>/ (scratch, written by tools/collect_profiles.sh on the GPU box) -> profiles/<tag>_* (tracked): the summaries that are judged.
set -u
TAG=${1:-r06}
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/gpurun_out/profiles_$TAG
D=$R/profiles
[ -d "$S" ] || { echo "no $S"; exit 1; }
for f in "$S"/*; do
  b=$(basename "$f")
  case "$b" in
    *.log|*.err) continue ;;
    split_kernel_stats.csv) o=kernel_stats_split.csv ;;
    nosplit_kernel_stats.csv) o=kernel_stats_nosplit.csv ;;
    x3_kernel_stats.csv|fp16_kernel_stats.csv|mixed_kernel_stats.csv) o=kernel_stats_${b%%_*}_nosplit.csv ;;
    *) o=$b ;;
  esac
  [ -s "$f" ] && cp "$f" "$D/${TAG}_$o"
done
ls "$D" | grep -c "^${TAG}_"

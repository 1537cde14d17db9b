# same-box A/B of the 4-key bias runs in the SwinV2 window attention (library rebuilt with the debug switches compiled in)
mkdir -p gpurun_out/r4m
MDPT_EXTRA_HIPCC_FLAGS=-DMDPT_DEBUG_SWITCHES python -c "from muggled_dpt_amd import native; native.build(force=True)" > gpurun_out/r4m/build.log 2>&1
for r in 1 2 3; do
  for v in run4 gather; do
    if [ $v = gather ]; then export MDPT_SWIN_NO_RUN4=1; else unset MDPT_SWIN_NO_RUN4; fi
    timeout 300 python bench.py --model swinl --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4m/b.json 2> gpurun_out/r4m/b.err
    python - "$v" "$r" <<'PY'
import json, sys
d=json.load(open('gpurun_out/r4m/b.json'))
att=[ (k,v) for k,v in d['kernel_time_share'].items() if k.startswith('attn')]
print(f"== {sys.argv[1]:6s} round {sys.argv[2]}  {d['value']:8.1f} maps/s  {d['ms_per_step']:.3f} ms  attention share {att}")
PY
  done
done | tee gpurun_out/r4m/ab.txt

mkdir -p gpurun_out/r4r
MDPT_EXTRA_HIPCC_FLAGS=-DMDPT_DEBUG_SWITCHES python -c "from muggled_dpt_amd import native; native.build(force=True)" > gpurun_out/r4r/build.log 2>&1
for r in 1 2 3; do
  echo "== base (narrow at 1297, wide at 5477) round $r"; python tools/probes/gpu_attn_kernel_ab.py 2>&1 | grep -v amdgpu.ids
  echo "== CMAX round $r"; MDPT_ATTN_CMAX=1 python tools/probes/gpu_attn_kernel_ab.py 2>&1 | grep -v amdgpu.ids
  echo "== base, wide forced round $r"; MDPT_ATTN_WIDE=1 python tools/probes/gpu_attn_kernel_ab.py 2>&1 | grep -v amdgpu.ids
  echo "== CMAX, wide forced round $r"; MDPT_ATTN_WIDE=1 MDPT_ATTN_CMAX=1 python tools/probes/gpu_attn_kernel_ab.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4r/attn_cmax_ab.txt

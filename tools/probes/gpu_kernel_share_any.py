#!/usr/bin/env python3
"""Per-kernel table (in-library HIP-event profile, batch split off) of any bench.py model:
   python tools/probes/gpu_kernel_share_any.py swinl 384 16 [latency | bf16 | fp16 | mixed | bf16x3 | fp16x3] [class=passes,...]"""
import ctypes, json, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
from muggled_dpt_amd import native
name, size, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
lib = native.load()
model, _ = bench.make_model_and_weights(name)
opt = sys.argv[4] if len(sys.argv) > 4 else ""
dt = torch.float32 if opt in native.PRECISIONS and opt != "bf16" else torch.bfloat16
model = model.to("cuda", dt)
x = torch.randn(batch, 3, size, size, generator=torch.Generator().manual_seed(1)).to("cuda", dt)
if opt == "latency":
    model.set_latency_mode(True)
elif opt in native.PRECISIONS:
    model.set_precision(opt)
if len(sys.argv) > 5:  # per-class pass counts on top of the mode, e.g. reasm=5,fusion_proj=5,fusion=4,head=4 (4 / 5 = fp8 cross terms)
    model.set_class_passes({k: int(v) for k, v in (kv.split("=") for kv in sys.argv[5].split(","))})
native.check(lib, lib.mdpt_set_batch_split(model._get_engine().handle, 0))
native.check(lib, lib.mdpt_debug_set_reassemble_overlap(model._get_engine().handle, 0))  # nothing on the side stream: every kernel alone
with torch.inference_mode():
    for _ in range(2): model(x)
    torch.cuda.synchronize()
    lib.mdpt_profile_enable(1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): model(x)
    e1.record(); torch.cuda.synchronize()
buf = ctypes.create_string_buffer(1 << 16)
lib.mdpt_profile_report(buf, len(buf)); lib.mdpt_profile_enable(0)
pr = json.loads(buf.value.decode())
tot = sum(k["total_ms"] for k in pr["kernels"]) / 5
print(f"{name} {size} B={batch} {opt}: forward {e0.elapsed_time(e1)/5:7.3f} ms (sum of kernels {tot:.3f})")
for k in pr["kernels"]:
    print(f"  {k['name']:52s} {k['launches'] / 5:5.1f} launches/fwd  avg {k['avg_us']:8.1f} us  {k['total_ms'] / 5:7.3f} ms/fwd ({k['total_ms'] / 5 / tot * 100:4.1f} %)  "
          f"{(k['tflops'] / 2500 if k['gflop'] > 0 else float('nan')):6.3f} of peak")

#!/usr/bin/env python3
"""How long does a launch-heavy forward (SwinV2-L 384x384 batch 16, ~330 launches per 10 ms step) take to reach its steady step time after the GPU
sat idle - (a) host asleep, (b) host busy on 32 torch threads (what the CPU oracle of a bench leg does)? Prints the step time in windows of
0.1 s after the idle period. Decides the warm-up of bench.py's secondary legs."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench

model, _ = bench.make_model_and_weights("swinl")
model = model.to("cuda", torch.bfloat16)
x = torch.randn(16, 3, 384, 384, generator=torch.Generator().manual_seed(11)).to("cuda", torch.bfloat16)


def windows(label, seconds=2.0):
    out = []
    t_end = time.perf_counter() + seconds
    with torch.inference_mode():
        while time.perf_counter() < t_end:
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 0.1:
                model(x)
                n += 1
                if n % 4 == 0:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / n * 1e3)
    print(f"{label:44s}", " ".join(f"{v:5.1f}" for v in out), flush=True)


windows("cold start")
windows("straight after")
for idle in (2, 10):
    time.sleep(idle)
    windows(f"after {idle} s asleep")
a = torch.randn(4096, 4096)
for threads in (32, max(1, (os.cpu_count() or 2) // 2)):
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 10:
        a @ a
    windows(f"after 10 s of matmuls on {threads} threads")
    windows("  and the 2 s after that")
torch.set_num_threads(1)
windows("num_threads back to 1")

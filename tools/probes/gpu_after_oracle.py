#!/usr/bin/env python3
"""Does a CPU oracle forward (torch fp32 on 32 threads: conv2d / SDPA / matmul) slow the launch-heavy SwinV2-L forward that follows it, and for how
long? (bench.py's secondary legs read 14.4 / 21 ms instead of 10.4 / 15 ms for whichever timing followed an oracle call.)"""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
from oracle import dpt_oracle

model, ow = bench.make_model_and_weights("swinl", want_weights=True)
model = model.to("cuda", torch.bfloat16)
x_cpu = torch.randn(16, 3, 384, 384, generator=torch.Generator().manual_seed(11))
x = x_cpu.to("cuda", torch.bfloat16)


def windows(label, seconds=1.5):
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


def host_only(label):
    """host cost of queueing one forward (no waiting for the GPU beyond the queue depth)"""
    torch.cuda.synchronize()
    with torch.inference_mode():
        t0 = time.perf_counter()
        model(x)
        t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{label:44s} host time to queue one forward {1e3 * (t1 - t0):.2f} ms, affinity {len(os.sched_getaffinity(0))} cpus, on cpu {os.sched_getcpu() if hasattr(os, 'sched_getcpu') else '?'}", flush=True)


windows("before")
host_only("before")
torch.set_num_threads(32)
t0 = time.perf_counter()
ref = dpt_oracle.forward(ow[1], ow[0], x_cpu[:1])
print(f"oracle: {time.perf_counter() - t0:.1f} s", flush=True)
host_only("after the oracle")
windows("after the oracle")
windows("next 1.5 s")
m2, _ = bench.make_model_and_weights("swinl")
m2 = m2.to("cuda", torch.float32)
host_only("after building another model")
windows("after building another model")
torch.set_num_threads(1)
windows("num_threads = 1")
host_only("num_threads = 1")

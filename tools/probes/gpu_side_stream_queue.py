#!/usr/bin/env python3
"""Step time of a split-batch forward (SwinV2-L 384x384 batch 16) as a function of HOW MANY streams the process created before the handle made
its side stream: taking the first stream the runtime hands out (probe off) against probing up to four candidates for one that runs beside the
caller's stream (shipped; csrc/stream_probe.hip). Streams share GPU_MAX_HW_QUEUES (4) hardware queues: when the side stream lands on the
caller's queue the two half batches run back to back."""
import ctypes, os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench
from muggled_dpt_amd import native
hip = ctypes.CDLL("libamdhip64.so")
lib = native.load()
x = torch.randn(16, 3, 384, 384, generator=torch.Generator().manual_seed(11)).to("cuda", torch.bfloat16)
keep = []
for extra in range(0, 9):
    row = []
    info = ""
    for probe in (0, 1):
        model, _ = bench.make_model_and_weights("swinl")
        model = model.to("cuda", torch.bfloat16)
        h = model._get_engine().handle
        native.check(lib, lib.mdpt_debug_set_side_stream_probe(h, probe))
        dt, _ = bench.time_model(model, x, 20)
        row.append(dt * 1e3)
        if probe:
            c, r = ctypes.c_int32(), ctypes.c_int32()
            native.check(lib, lib.mdpt_debug_side_stream_info(h, ctypes.byref(c), ctypes.byref(r)))
            info = f"({r.value} of {c.value} candidates were on the caller's queue)"
        del model
    print(f"{len(keep):2d} other streams alive: first stream handed out {row[0]:6.2f} ms   probed {row[1]:6.2f} ms  {info}", flush=True)
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    keep.append(s)

"""The RCCL side of the data-parallel path on the hardware that is available to the test tier (one GPU): a one-rank "nccl" process group
must initialise on the box and carry DataParallelDepth's collective. The world-size-2 semantics are covered on CPU with gloo
(tests/test_parallel.py); the 8-GPU run is the driver's. `pytest -m gpu`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from muggled_dpt_amd.parallel import init_distributed, all_gather_maps, DataParallelDepth
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", init_method="env://", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.randn(4, 56, 56, device="cuda").to(torch.bfloat16)
out = torch.empty_like(x)
dist.all_gather_into_tensor(out, x)          # the exact collective DataParallelDepth issues (RCCL all-gather on this GPU)
torch.cuda.synchronize()
assert torch.equal(out, x)
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK", torch.cuda.nccl.version())
""" % REPO


def test_rccl_backend_initialises_and_allgathers_on_this_box():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


WRAPPER_SCRIPT = r"""
import ctypes, sys, torch
sys.path.insert(0, %r)
from muggled_dpt_amd import native
lib = native.load()
rccl = None
for name in ("librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
    try:
        rccl = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
        break
    except OSError:
        continue
assert rccl is not None, "librccl.so not found"
comm = ctypes.c_void_p()
devs = (ctypes.c_int * 1)(0)
assert rccl.ncclCommInitAll(ctypes.byref(comm), 1, devs) == 0
x = torch.randn(3, 56, 84, device="cuda")
out = torch.empty_like(x)
native.check(lib, lib.mdpt_allgather(comm, x.data_ptr(), out.data_ptr(), x.numel(), native.dtype_code(x.dtype), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
assert torch.equal(out, x)
rccl.ncclCommDestroy(comm)
print("WRAPPER_OK")
""" % REPO


def test_c_abi_allgather_wrapper_on_a_one_rank_communicator():
    """mdpt_allgather (the dtype-tagged ncclAllGather wrapper for non-torch hosts): communicator created straight from librccl with ctypes
    (in a child process: RCCL prints a version banner at exit)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", WRAPPER_SCRIPT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "WRAPPER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]

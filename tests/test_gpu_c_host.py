"""The drop-in boundary without PyTorch on the calling side: tests/c_host/host_main.cpp is a plain C++ program (HIP runtime + include/mdpt.h
only) that creates a model from a weight file, runs mdpt_forward and writes the depth maps. Its output is checked against the oracle.
`pytest -m gpu`."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

from tests.helpers import rel_err, seeded_input, synthetic_model

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "tests", "c_host", "host_main.cpp")
EXE = os.path.join(REPO, "tests", "c_host", "host_main")


def build_c_host() -> str:
    from muggled_dpt_amd import native
    lib = native.build()
    if os.path.exists(EXE) and os.path.getmtime(EXE) >= max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(REPO, "include", "mdpt.h"))):
        return EXE
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")
    assert hipcc, "hipcc not found"
    r = subprocess.run([hipcc, "-O2", "-std=c++17", "-I", os.path.join(REPO, "include"), SRC, "-L", os.path.dirname(lib), "-lmdpt",
                        "-Wl,-rpath," + os.path.dirname(lib), "-o", EXE], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return EXE


def _write_weights(path, cfg, w, precision):
    with open(path, "wb") as f:
        c = [cfg["features_per_token"], cfg["num_heads"], cfg["num_blocks"], *cfg["reassembly_features_list"], *cfg["base_patch_grid_hw"],
             cfg["fusion_channels"], cfg["patch_size_px"]]
        f.write(struct.pack("<11i", *c))
        f.write(struct.pack("<2i", precision, len(w)))
        for name, t in w.items():
            nb = name.encode()
            f.write(struct.pack("<i", len(nb)) + nb)
            f.write(struct.pack("<i", t.dim()))
            f.write(struct.pack(f"<{t.dim()}q", *t.shape))
            f.write(t.detach().contiguous().to(torch.float32).numpy().tobytes())


@pytest.mark.parametrize("precision,tol", [(1, 1e-4), (0, 2e-2)])
def test_plain_cpp_host_runs_the_path_through_the_c_abi(tmp_path, precision, tol):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from oracle import dpt_oracle
    exe = build_c_host()
    osd, cfg, w = synthetic_model("tiny", 0)
    x = seeded_input((3, 3, 56, 84), 17)
    wp, ip, op = (str(tmp_path / n) for n in ("weights.bin", "input.bin", "output.bin"))
    _write_weights(wp, cfg, w, precision)
    with open(ip, "wb") as f:
        f.write(struct.pack("<3i", 3, 56, 84))
        f.write(x.numpy().tobytes())
    r = subprocess.run([exe, wp, ip, op], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C_HOST_OK" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
    y = torch.from_numpy(np.fromfile(op, dtype=np.float32).reshape(3, 56, 84))
    assert rel_err(y, dpt_oracle.forward(w, cfg, x)) <= tol

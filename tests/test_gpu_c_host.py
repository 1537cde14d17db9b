"""The drop-in boundary without PyTorch on the calling side: tests/c_host/host_main.cpp is a plain C++ program (HIP runtime + include/mdpt.h
only) that loads a `.mdpt` model file (DPTModel.export, muggled_dpt_amd/export.py), runs mdpt_forward / mdpt_prepare_image and writes the depth
maps. Its output is checked against the oracle and, bit for bit, against the Python facade - all four families, two sizes per handle.
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


def _run_host(exe, model_path, jobs, timeout=600):
    """jobs: [(input file, output file, is_image, extra args)] on one handle -> [(B, H, W, torch tensor in the output's dtype)]"""
    args = [exe, model_path]
    for inp, out, is_image, extra in jobs:
        args += (["--image"] if is_image else []) + [inp, out] + [str(e) for e in extra]
    r = subprocess.run(args, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0 and r.stdout.count("C_HOST_OK") == len(jobs) and "C_HOST_DONE" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    outs = []
    for _, out, _, _ in jobs:
        raw = open(out, "rb").read()
        b, h, w, dt = struct.unpack_from("<4i", raw, 0)
        if dt == 0:
            t = torch.from_numpy(np.frombuffer(raw, dtype=np.float32, offset=16).copy())
        elif dt == 2:
            t = torch.from_numpy(np.frombuffer(raw, dtype=np.float16, offset=16).copy())
        else:
            t = torch.from_numpy(np.frombuffer(raw, dtype=np.int16, offset=16).copy()).view(torch.bfloat16)
        outs.append(t.reshape(b, h, w))
    return outs


def _write_tensor_input(path, x):
    code = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[x.dtype]
    with open(path, "wb") as f:
        f.write(struct.pack("<4i", x.shape[0], x.shape[2], x.shape[3], code))
        f.write((x.view(torch.int16) if x.dtype == torch.bfloat16 else x).contiguous().numpy().tobytes())


def _family_model(family):
    if family == "beit":
        from muggled_dpt_amd import make_beit_dpt_from_midas_v31_state_dict as make
        from muggled_dpt_amd.synthetic import make_synthetic_beit_state_dict as synth
        return make(synth("beit_tiny", 0))[1], 16 * 2
    if family == "swinv2":
        from muggled_dpt_amd import make_swinv2_dpt_from_midas_v31_state_dict as make
        from muggled_dpt_amd.synthetic import make_synthetic_swinv2_state_dict as synth
        return make(synth("swin2_tiny", 0))[1], 32
    from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
    if family == "v1":
        from muggled_dpt_amd import make_depthanythingv1_dpt_from_original_state_dict as make
        from muggled_dpt_amd.synthetic import STANDARD_CONFIGS
        return make(make_synthetic_original_state_dict(dict(STANDARD_CONFIGS["tiny"], num_blocks=8), 0))[1], 28
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict as make
    return make(make_synthetic_original_state_dict("tiny", 0))[1], 28


@pytest.mark.parametrize("family,dtype,precision", [("v2", torch.float32, None), ("v2", torch.bfloat16, None), ("v2", torch.float32, "mixed"), ("v1", torch.float16, None),
                                                    ("beit", torch.float32, None), ("beit", torch.bfloat16, None), ("swinv2", torch.float32, "mixed"), ("swinv2", torch.bfloat16, None)])
def test_exported_model_file_runs_in_the_torch_free_host_bit_for_bit(tmp_path, family, dtype, precision):
    """SURVEY §8(f) row 4, the export half (experiments/export_onnx.py:119-148 hands the model to another runtime with dynamic H / W): DPTModel.export
    writes the `.mdpt` artefact, the plain C++ host (HIP runtime + include/mdpt.h only) loads it - every family, parameters in their own dtype, the
    arithmetic mode and per-class passes of the model - and runs TWO image sizes on one handle. Its depth maps equal the facade's bit for bit."""
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    exe = build_c_host()
    model, unit = _family_model(family)
    model = model.to(dtype)
    if precision:
        model.set_precision(precision)
        model.set_class_passes({"fusion_in": 2})  # a non-default class table travels with the file too
    mp = str(tmp_path / "model.mdpt")
    summary = model.export(mp)
    assert summary["tensors"] > 20 and os.path.getsize(mp) > summary["parameter_bytes"]
    sizes = [(2, 2 * unit, 3 * unit), (1, 4 * unit, 2 * unit)] if family != "swinv2" else [(2, 128, 192), (1, 256, 128)]
    jobs, xs = [], []
    for k, (b, hh, ww) in enumerate(sizes):
        x = seeded_input((b, 3, hh, ww), 40 + k).to(dtype)
        xs.append(x)
        _write_tensor_input(str(tmp_path / f"in{k}.bin"), x)
        jobs.append((str(tmp_path / f"in{k}.bin"), str(tmp_path / f"out{k}.bin"), False, []))
    outs = _run_host(exe, mp, jobs)
    gpu = model.to("cuda")
    for x, y_host in zip(xs, outs):
        y = gpu(x.cuda()).cpu()
        assert y.dtype == y_host.dtype == dtype and y.shape == y_host.shape
        assert torch.equal(y.view(torch.int16) if dtype != torch.float32 else y, y_host.view(torch.int16) if dtype != torch.float32 else y_host), \
            f"{family} {dtype} {precision}: the torch-free host and the facade differ"
        assert float(y.float().abs().max()) > 0


def test_torch_free_host_runs_inference_from_a_uint8_image(tmp_path):
    """DPTModel.inference (dpt_model.py:87-109) without PyTorch: uint8 BGR image -> mdpt_prepare_image (file's normalisation + the reference's
    size rule incl. its banker's rounding) -> mdpt_forward. Equal to the facade's inference() bit for bit, square and aspect-ratio sizing."""
    exe = build_c_host()
    model, _ = _family_model("v2")
    model = model.to(torch.bfloat16)
    mp = str(tmp_path / "model.mdpt")
    model.export(mp)
    rng = np.random.default_rng(9)
    cases = [((333, 517), 140, 1), ((333, 517), 196, 0), ((70, 70), 140, 1)]
    jobs = []
    for k, ((ih, iw), side, square) in enumerate(cases):
        img = rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
        with open(tmp_path / f"img{k}.bin", "wb") as f:
            f.write(struct.pack("<2i", ih, iw) + img.tobytes())
        jobs.append((str(tmp_path / f"img{k}.bin"), str(tmp_path / f"o{k}.bin"), True, [side, square]))
        cases[k] = (img, side, square)
    outs = _run_host(exe, mp, jobs)
    gpu = model.to("cuda")
    for (img, side, square), y_host in zip(cases, outs):
        y = gpu.inference(img, side, bool(square)).cpu()
        assert y.shape == y_host.shape, (y.shape, y_host.shape)
        assert torch.equal(y.view(torch.int16), y_host.view(torch.int16))


@pytest.mark.parametrize("precision,tol", [("bf16x3", 1e-4), ("bf16", 2e-2)])
def test_plain_cpp_host_runs_the_path_through_the_c_abi(tmp_path, precision, tol):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from oracle import dpt_oracle
    exe = build_c_host()
    osd, cfg, w = synthetic_model("tiny", 0)
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    model.set_precision(precision)
    x = seeded_input((3, 3, 56, 84), 17)
    mp, ip, op = (str(tmp_path / n) for n in ("model.mdpt", "input.bin", "output.bin"))
    model.export(mp)
    _write_tensor_input(ip, x)
    (y,) = _run_host(exe, mp, [(ip, op, False, [])])
    assert rel_err(y, dpt_oracle.forward(w, cfg, x)) <= tol

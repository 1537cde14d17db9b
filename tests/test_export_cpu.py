"""The `.mdpt` deployment artefact (muggled_dpt_amd/export.py; the native analogue of the reference's experiments/export_onnx.py:119-148): writing and
reading need no GPU. Every family: export -> read -> rebuild, tensors bit for bit, settings carried; foreign files are refused."""
import os
import struct

import pytest
import torch

from muggled_dpt_amd import export as mexport
from muggled_dpt_amd import native


def _model(family):
    from muggled_dpt_amd.synthetic import (STANDARD_CONFIGS, make_synthetic_beit_state_dict, make_synthetic_original_state_dict,
                                           make_synthetic_swinv2_state_dict)
    import muggled_dpt_amd as m
    if family == "beit":
        return m.make_beit_dpt_from_midas_v31_state_dict(make_synthetic_beit_state_dict("beit_tiny", 0))[1]
    if family == "swinv2":
        return m.make_swinv2_dpt_from_midas_v31_state_dict(make_synthetic_swinv2_state_dict("swin2_tiny", 0))[1]
    if family == "v1":
        return m.make_depthanythingv1_dpt_from_original_state_dict(make_synthetic_original_state_dict(dict(STANDARD_CONFIGS["tiny"], num_blocks=8), 0))[1]
    return m.make_depthanythingv2_dpt_from_original_state_dict(make_synthetic_original_state_dict("tiny", 0))[1]


@pytest.mark.parametrize("family", ["v2", "v1", "beit", "swinv2"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_export_read_rebuild_roundtrip(tmp_path, family, dtype):
    model = _model(family).to(dtype)
    model.set_precision("mixed")
    model.set_class_passes({"head": 3, "fusion_in": 2})
    model.set_weight_rounding_compensation(False)
    path = str(tmp_path / "m.mdpt")
    summary = model.export(path)
    rec = mexport.read_model_file(path)
    assert rec["abi_version"] == native.ABI_VERSION and rec["config"].precision == native.PREC_MIXED
    assert rec["config"].family == {"v2": 0, "v1": 1, "beit": 2, "swinv2": 3}[family]
    assert rec["class_passes"] == {"head": 3, "fusion_in": 2} and rec["wrc"] is False and rec["latency_mode"] is False
    assert tuple(round(v, 4) for v in rec["rgb_mean"]) == tuple(round(v, 4) for v in model.patch_embed.rgb_offset)
    assert rec["tiling_size"] == model.patch_embed._tiling_size
    sd = {f"{c}.{k}": v for c in mexport.COMPONENTS for k, v in getattr(model, c).state_dict().items()}
    assert sd.keys() == rec["tensors"].keys() and summary["tensors"] == len(sd)
    as_bits = lambda t: t.view(torch.int16) if t.dtype != torch.float32 else t.view(torch.int32)
    assert all(v.dtype == dtype and torch.equal(as_bits(v), as_bits(sd[k])) for k, v in rec["tensors"].items())
    cfg, again = mexport.load_exported(path)
    assert again.family == family and again.__dict__["_precision"] == "mixed" and again.__dict__["_class_passes"] == {"head": 3, "fusion_in": 2}
    assert again.__dict__["_wrc"] is False and next(again.parameters()).dtype == dtype
    sd2 = {f"{c}.{k}": v for c in mexport.COMPONENTS for k, v in getattr(again, c).state_dict().items()}
    assert all(torch.equal(as_bits(sd2[k]), as_bits(sd[k])) for k in sd)
    # the library accepts the stored config as it is (the torch-free host hands exactly these bytes to mdpt_create)
    import ctypes
    lib = native.load()
    h = ctypes.c_void_p()
    native.check(lib, lib.mdpt_create(ctypes.byref(rec["config"]), ctypes.byref(h)))
    names = {lib.mdpt_weight_name(h, i).decode() for i in range(lib.mdpt_num_weights(h))}
    lib.mdpt_destroy(h)
    assert names <= set(rec["tensors"]), sorted(names - set(rec["tensors"]))[:5]


def test_precision_follows_the_stored_dtype_and_a_class_mask_of_the_compensation_travels(tmp_path):
    """an fp32 model exported with dtype=bfloat16 stores 2-byte tensors - both readers of the file (the C host through mdpt_config.precision,
    load_exported through the tensor dtype) have to run the SAME arithmetic: plain bf16, not the fp32 class of the exporting model. And the per-class
    form of mdpt_set_weight_rounding_compensation (round 6) round-trips as names."""
    model = _model("v2")  # float32, no explicit precision: bf16x3 as it stands
    path = str(tmp_path / "m.mdpt")
    model.export(path, dtype=torch.bfloat16)
    rec = mexport.read_model_file(path)
    assert rec["config"].precision == native.PREC_BF16
    _, again = mexport.load_exported(path)
    from muggled_dpt_amd.dpt_model import model_precision_code
    assert model_precision_code(again, next(again.parameters()).dtype) == native.PREC_BF16
    model.export(path)  # stored as held: the fp32 class
    assert mexport.read_model_file(path)["config"].precision == native.PREC_BF16X3
    model.set_precision("mixed")
    model.set_weight_rounding_compensation(("qkv", "fc2"))
    model.export(path)
    rec = mexport.read_model_file(path)
    assert rec["config"].precision == native.PREC_MIXED and set(rec["wrc"]) == {"qkv", "fc2"}
    _, again = mexport.load_exported(path)
    assert again.__dict__["_wrc"] == (1 << native.OP_CLASSES.index("qkv")) | (1 << native.OP_CLASSES.index("fc2"))
    with pytest.raises(ValueError):
        model.set_weight_rounding_compensation(("attn",))


def test_foreign_and_stale_files_are_refused(tmp_path):
    bad = tmp_path / "x.mdpt"
    bad.write_bytes(b"not a model")
    with pytest.raises(ValueError, match="bad magic"):
        mexport.read_model_file(str(bad))
    model = _model("v2")
    good = str(tmp_path / "g.mdpt")
    model.export(good, dtype=torch.bfloat16)
    raw = bytearray(open(good, "rb").read())
    raw[8:12] = struct.pack("<i", native.ABI_VERSION + 1)
    stale = tmp_path / "s.mdpt"
    stale.write_bytes(bytes(raw))
    with pytest.raises(ValueError, match="ABI version"):
        mexport.read_model_file(str(stale))
    assert os.path.getsize(good) < 0.6 * sum(p.numel() * 4 for p in model.parameters()) + 200_000  # stored in the requested 2-byte dtype

"""CPU-side checks of the C-ABI shared library (no GPU, no compute calls): it loads, exports every symbol
include/mdpt.h declares, validates configs, enumerates the reference's parameter names and plans workspaces."""
import ctypes
import os
import re

import pytest

from muggled_dpt_amd import native
from muggled_dpt_amd.state_dict_conversion import expected_new_keys
from muggled_dpt_amd.synthetic import STANDARD_CONFIGS

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return native.load()


def _cfg(name="vitl", precision=native.PREC_BF16, **over):
    c = dict(STANDARD_CONFIGS[name])
    c.update(over)
    s = native.MdptConfig()
    s.features_per_token, s.num_heads, s.num_blocks = c["features_per_token"], c["num_heads"], c["num_blocks"]
    for i, v in enumerate(c["reassembly_features_list"]):
        s.reassembly_features[i] = v
    s.base_patch_grid_h, s.base_patch_grid_w = c["base_patch_grid_hw"]
    s.fusion_channels, s.patch_size_px = c["fusion_channels"], c["patch_size_px"]
    s.is_giant, s.is_metric, s.precision = int(c.get("is_giant", 0)), 0, precision
    return s


def test_every_header_symbol_is_exported(lib):
    header = open(os.path.join(REPO, "include", "mdpt.h")).read()
    declared = set(re.findall(r"\b(mdpt_[a-z0-9_]+)\s*\(", header))
    declared -= {"mdpt_config", "mdpt_handle"}
    assert len(declared) >= 20
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/mdpt.h but not exported by libmdpt.so"
    assert declared == set(native.SYMBOLS), declared ^ set(native.SYMBOLS)


def test_create_validates_config(lib):
    h = ctypes.c_void_p()
    assert lib.mdpt_create(ctypes.byref(_cfg()), ctypes.byref(h)) == 0
    lib.mdpt_destroy(h)
    for bad in (dict(features_per_token=1000), dict(num_heads=8), dict(num_blocks=10), dict(patch_size_px=15)):
        rc = lib.mdpt_create(ctypes.byref(_cfg(**bad)), ctypes.byref(h))
        assert rc < 0 and lib.mdpt_last_error(), bad
    assert lib.mdpt_create(ctypes.byref(_cfg(precision=7)), ctypes.byref(h)) == -1


@pytest.mark.parametrize("name", ["tiny", "vits", "vitl", "vitg", "tiny_giant"])
def test_parameter_inventory_uses_reference_key_names(lib, name):
    h = ctypes.c_void_p()
    assert lib.mdpt_create(ctypes.byref(_cfg(name)), ctypes.byref(h)) == 0
    names = [lib.mdpt_weight_name(h, i).decode() for i in range(lib.mdpt_num_weights(h))]
    want = {f"{comp}.{k}" for comp, keys in expected_new_keys(STANDARD_CONFIGS[name]).items() for k in keys}
    assert set(names) == want and len(names) == len(want)
    ndim, shape = ctypes.c_int32(), (ctypes.c_int64 * 4)()
    i = names.index("imgencoder.stages.0.blocks.0.attn.qkv.weight")
    assert lib.mdpt_weight_shape(h, i, ctypes.byref(ndim), shape) == 0
    f = STANDARD_CONFIGS[name]["features_per_token"]
    assert ndim.value == 2 and list(shape)[:2] == [3 * f, f]
    lib.mdpt_destroy(h)


def test_bind_rejects_unknown_and_misshapen_weights_and_forward_needs_finalize(lib):
    h = ctypes.c_void_p()
    assert lib.mdpt_create(ctypes.byref(_cfg("tiny")), ctypes.byref(h)) == 0
    shape = (ctypes.c_int64 * 1)(64)
    assert lib.mdpt_bind_weight(h, b"not.a.key", 4096, 0, 1, shape) == -1
    assert lib.mdpt_bind_weight(h, b"patch_embed.proj.bias", 4096, 0, 1, (ctypes.c_int64 * 1)(63)) == -4
    assert b"size mismatch" in lib.mdpt_last_error()
    assert lib.mdpt_bind_weight(h, b"patch_embed.proj.bias", 4096, 1, 1, shape) == 0  # a bf16 tensor
    # strict load: finalize refuses while parameters are missing (reference: strict load_state_dict RuntimeError)
    assert lib.mdpt_finalize(h, 4096 * 256, 1 << 40, None) == -3 and b"missing parameter" in lib.mdpt_last_error()
    assert lib.mdpt_forward(h, 4096, 0, 1, 56, 56, 4096, 0, 4096 * 256, 1 << 40, None) == -2  # not finalized
    lib.mdpt_destroy(h)


def test_workspace_planning_and_grid_rules(lib):
    h = ctypes.c_void_p()
    assert lib.mdpt_create(ctypes.byref(_cfg("vitl")), ctypes.byref(h)) == 0
    n1, n32 = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.mdpt_workspace_bytes(h, 1, 504, 504, ctypes.byref(n1)) == 0
    assert lib.mdpt_workspace_bytes(h, 32, 504, 504, ctypes.byref(n32)) == 0
    assert 0 < n1.value < n32.value < 64 << 30 and n32.value > 20 * n1.value
    assert lib.mdpt_workspace_bytes(h, 1, 518, 518, ctypes.byref(n1)) == native.E_GRID  # 37x37 grid is odd
    assert b"even" in lib.mdpt_last_error()
    assert lib.mdpt_workspace_bytes(h, 1, 500, 504, ctypes.byref(n1)) == -1  # not divisible by the patch size
    assert lib.mdpt_workspace_bytes(h, 1, 1036, 1036, ctypes.byref(n1)) == 0
    p = ctypes.c_size_t()
    assert lib.mdpt_packed_bytes(h, ctypes.byref(p)) == 0 and 600e6 < p.value < 800e6  # ~334 M params in bf16
    lib.mdpt_destroy(h)


def test_make_dpt_routes_v1_and_v2_by_file_name(tmp_path):
    """determine_model_type_from_state_dict: same key sniffing / file-name rules as the reference (make_dpt.py:78-116)."""
    import torch
    from muggled_dpt_amd.make_dpt import determine_model_type_from_state_dict, make_dpt_from_state_dict
    from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
    osd = make_synthetic_original_state_dict("tiny", 0)
    assert determine_model_type_from_state_dict("/x/depth_anything_v2_vits.pth", osd) == "depthanythingv2"
    assert determine_model_type_from_state_dict("/x/depth_anything_vitl14.pth", osd) == "depthanythingv1"
    assert determine_model_type_from_state_dict("/x/whatever.pth", {"pretrained.model.blocks.0.attn.relative_position_bias_table": 0}) == "beit"
    assert determine_model_type_from_state_dict("/x/whatever.pth", {"foo": 0}) == "unknown"
    p1 = str(tmp_path / "depth_anything_vits14.pth")
    torch.save(osd, p1)
    cfg, model = make_dpt_from_state_dict(p1)
    assert model.family == "v1" and len(cfg) == 9
    assert determine_model_type_from_state_dict("/x/w.pt", {"pretrained.model.layers.0.blocks.0.attn.logit_scale": 0}) == "swinv2"
    with pytest.raises(AssertionError):  # a Depth-Anything checkpoint forced through the SwinV2 loader fails in config sniffing
        make_dpt_from_state_dict(p1, model_type="swinv2")
    with pytest.raises(NotImplementedError):
        make_dpt_from_state_dict(p1, model_type="nonsense")


def test_graft_entry_build_check_matches_header_abi_version():
    """__graft_entry__.build() asserts the library's ABI version: keep native.ABI_VERSION and include/mdpt.h in step."""
    header = open(os.path.join(REPO, "include", "mdpt.h")).read()
    assert int(re.search(r"#define MDPT_ABI_VERSION (\d+)", header).group(1)) == native.ABI_VERSION == native.load().mdpt_abi_version()
    src = open(os.path.join(REPO, "__graft_entry__.py")).read()
    assert "native.ABI_VERSION" in src


def test_enable_optimizations_false_adds_hookable_softmax_modules_only():
    """reference components/transformer_block.py:79-136: the non-optimised attention owns an nn.Softmax that tooling hooks
    (demo_helpers/model_capture.py:54-59). Here they are parameter-free probe modules, one per block, in block order."""
    import torch.nn as nn
    import muggled_dpt_amd as mda
    from muggled_dpt_amd.synthetic import make_synthetic_beit_state_dict
    from tests.helpers import synthetic_model
    osd, _, _ = synthetic_model("tiny", 0)
    cfg, slow = mda.make_depthanythingv2_dpt_from_original_state_dict(osd, enable_optimizations=False)
    _, fast = mda.make_depthanythingv2_dpt_from_original_state_dict(osd)
    names = [n for n, m in slow.named_modules() if isinstance(m, nn.Softmax)]
    assert names == [f"imgencoder.stages.{b}.blocks.0.attn.softmax" for b in range(cfg["num_blocks"])]
    assert not any(isinstance(m, nn.Softmax) for m in fast.modules())
    assert list(slow.state_dict()) == list(fast.state_dict())
    bcfg, beit = mda.make_beit_dpt_from_midas_v31_state_dict(make_synthetic_beit_state_dict("beit_tiny", 1), enable_optimizations=False)
    assert sum(isinstance(m, nn.Softmax) for m in beit.modules()) == bcfg["num_blocks"]
    # the oracle's capture hands back one [B, heads, N, N] row-stochastic matrix per block
    import torch
    from oracle import dpt_oracle as orc
    from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components
    w = flatten_components(convert_state_dict_keys(cfg, osd))
    tokens, grid = orc.patch_embed(w, torch.randn(1, 3, 56, 56, generator=torch.Generator().manual_seed(0)))
    cap = []
    orc.image_encoder(w, cfg, tokens, grid, capture=cap)
    assert len(cap) == cfg["num_blocks"] and tuple(cap[0].shape) == (1, cfg["num_heads"], 17, 17)
    assert float((cap[-1].sum(-1) - 1).abs().max()) < 1e-5


def test_every_transformer_block_is_a_hookable_module_and_the_oracle_captures_block_outputs():
    """A forward hook on a block module is how the reference's tooling reads block outputs (demo_helpers/model_capture.py:54-59,
    experiments/block_norm_visualization.py:282). The block modules are parameter containers that pass a tensor through when called
    (so registered hooks fire with the tokens mdpt_encoder_probe_blocks dumped); making them hookable must not change the key set."""
    import torch
    import muggled_dpt_amd as mda
    from muggled_dpt_amd.dpt_model import _BlockProbe
    from muggled_dpt_amd.synthetic import make_synthetic_beit_state_dict, make_synthetic_swinv2_state_dict
    from tests.helpers import synthetic_model
    osd, _, _ = synthetic_model("tiny", 0)
    cfg, model = mda.make_depthanythingv2_dpt_from_original_state_dict(osd)
    blocks = [m for m in model.modules() if isinstance(m, _BlockProbe)]
    assert len(blocks) == cfg["num_blocks"]
    assert [n for n, m in model.named_modules() if isinstance(m, _BlockProbe)] == [f"imgencoder.stages.{b}.blocks.0" for b in range(cfg["num_blocks"])]
    seen = []
    h = blocks[1].register_forward_hook(lambda mod, args, out: seen.append(out))
    t = torch.arange(6.0).reshape(1, 2, 3)
    assert blocks[1](t) is t and len(seen) == 1 and seen[0] is t
    h.remove()
    assert all(k.count("_BlockProbe") == 0 for k in model.state_dict())
    _, beit = mda.make_beit_dpt_from_midas_v31_state_dict(make_synthetic_beit_state_dict("beit_tiny", 1))
    scfg, swin = mda.make_swinv2_dpt_from_midas_v31_state_dict(make_synthetic_swinv2_state_dict("swin2_tiny", 1))
    assert sum(isinstance(m, _BlockProbe) for m in beit.modules()) == 4
    assert sum(isinstance(m, _BlockProbe) for m in swin.modules()) == sum(scfg["layers_per_stage"])
    # the oracle's per-block capture: one [B, 1 + gh*gw, F] tensor per block, the last one out-normed is the last tap
    from oracle import dpt_oracle as orc
    from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components
    w = flatten_components(convert_state_dict_keys(cfg, osd))
    tokens, grid = orc.patch_embed(w, torch.randn(1, 3, 56, 56, generator=torch.Generator().manual_seed(0)))
    outs = []
    taps = orc.image_encoder(w, cfg, tokens, grid, block_outputs=outs)
    assert len(outs) == cfg["num_blocks"] and tuple(outs[0].shape) == (1, 17, cfg["features_per_token"])
    assert torch.equal(taps[3], orc.layernorm(outs[-1], w["imgencoder.outnorm.weight"], w["imgencoder.outnorm.bias"]))


def test_parameter_snapshot_key_works_for_inference_tensors_and_sees_replaced_parameters():
    """reading p._version of an inference tensor raises, so a model built under torch.inference_mode() failed on
    every call; and a Parameter replaced by attribute assignment kept the old packed snapshot alive."""
    import torch
    import muggled_dpt_amd as mda
    from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
    osd = make_synthetic_original_state_dict("tiny", 0)
    with torch.inference_mode():
        _, model = mda.make_depthanythingv2_dpt_from_original_state_dict(osd)
        model = model.to(torch.bfloat16)
    assert any(p.is_inference() for p in model.parameters())
    key0 = model._param_versions()  # must not raise
    assert key0 == model._param_versions()
    _, m2 = mda.make_depthanythingv2_dpt_from_original_state_dict(osd)
    k_before = m2._param_versions()
    with torch.no_grad():
        next(m2.parameters()).mul_(1.0)  # in-place edit: version counter moves
    assert m2._param_versions() != k_before
    k_before = m2._param_versions()
    name, old = next(iter(m2.named_parameters()))
    mod = m2
    *path, leaf = name.split(".")
    for part in path:
        mod = getattr(mod, part)
    setattr(mod, leaf, torch.nn.Parameter(old.detach().clone()))  # replaced Parameter object: identity moves
    assert m2._param_versions() != k_before

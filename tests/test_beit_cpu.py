"""CPU side of the MiDaS v3.1 BEiT family: oracle vs the fixtures generated from the reference (tests/golden/gen_golden.py),
checkpoint conversion contract, C-ABI inventory. No GPU compute."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from muggled_dpt_amd import native
from muggled_dpt_amd import state_dict_conversion_beit as conv
from muggled_dpt_amd.state_dict_conversion import flatten_components
from muggled_dpt_amd.synthetic import BEIT_CONFIGS, make_synthetic_beit_state_dict
from oracle import dpt_oracle
from tests.helpers import stats

ATOL = 2e-5


def _tiny(seed):
    osd = make_synthetic_beit_state_dict("beit_tiny", seed)
    cfg = conv.get_model_config_from_state_dict(osd)
    return osd, cfg, flatten_components(conv.convert_state_dict_keys(cfg, osd))


def test_config_sniffing_and_key_names_match_reference(golden_dir):
    osd, cfg, w = _tiny(5)
    want = dict(BEIT_CONFIGS["beit_tiny"], enable_cache=False, enable_optimizations=True)
    assert {k: (tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in cfg.items()} == \
           {k: (tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in want.items()}
    with open(os.path.join(golden_dir, "beit_tiny_new_keys.json")) as f:
        ref_keys = json.load(f)
    assert set(ref_keys) == set(w)
    for k, shp in ref_keys.items():
        assert list(w[k].shape) == shp, k
    # dropped / reshaped keys (convert_midas_state_dict_keys.py:137-161)
    assert w["imgencoder.stages.0.blocks.0.attn.q_bias"].shape == (1, 2, 1, 64)
    osd["pretrained.model.blocks.0.attn.relative_position_index"] = torch.zeros(17, 17, dtype=torch.long)
    assert set(flatten_components(conv.convert_state_dict_keys(cfg, osd))) == set(w)
    with pytest.raises(AssertionError):
        conv.get_model_config_from_state_dict({k: v for k, v in osd.items() if "patch_embed" not in k})


@pytest.mark.parametrize("tag", ["base", "wide", "tall"])
def test_oracle_every_stage_boundary(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "beit_tiny.npz"))
    _, cfg, w = _tiny(int(g["weight_seed"]))
    depth, st = dpt_oracle.forward(w, cfg, torch.from_numpy(g[f"{tag}_input"]), return_stages=True)
    assert float((depth - torch.from_numpy(g[f"{tag}_depth"])).abs().max()) <= ATOL
    assert float((st["patch_tokens"] - torch.from_numpy(g[f"{tag}_patch_tokens"])).abs().max()) <= ATOL
    for i in range(4):
        assert float((st["stages"][i] - torch.from_numpy(g[f"{tag}_tap{i}"])).abs().max()) <= ATOL
        assert float((st["reasm"][i] - torch.from_numpy(g[f"{tag}_reasm{i}"])).abs().max()) <= ATOL
    assert float((st["fused"] - torch.from_numpy(g[f"{tag}_fused"])).abs().max()) <= ATOL


def test_oracle_relative_position_bias(golden_dir):
    g = np.load(os.path.join(golden_dir, "beit_tiny.npz"))
    lut = torch.from_numpy(g["relpos_lut"])
    for grid in ((4, 4), (4, 6), (6, 2), (8, 8)):
        ref = torch.from_numpy(g[f"relpos_g{grid[0]}x{grid[1]}"])
        got = dpt_oracle.beit_relpos_bias(lut, (4, 4), grid)
        assert got.shape == ref.shape == (1, 2, grid[0] * grid[1] + 1, grid[0] * grid[1] + 1)
        assert float((got - ref).abs().max()) <= 1e-6
    # at the base grid the table is used as-is: token (0,0)->(0,0) reads the centre entry, cls entries are the last three
    idx = dpt_oracle.beit_relative_position_index((4, 4))
    assert int(idx[1, 1]) == 3 * 7 + 3 and int(idx[0, 5]) == 49 and int(idx[5, 0]) == 50 and int(idx[0, 0]) == 51


def test_oracle_prepare_image(golden_dir):
    g = np.load(os.path.join(golden_dir, "beit_prepare_image.npz"))
    kw = dict(default_size_px=64, tiling_px=32, rgb_mean=(0.5,) * 3, rgb_std=(0.5,) * 3)
    a = dpt_oracle.prepare_image(g["image"], **kw)
    b = dpt_oracle.prepare_image(g["image"], max_side_length=256, use_square_sizing=False, **kw)
    assert tuple(a.shape) == (1, 3, 64, 64) and float((a - torch.from_numpy(g["default"])).abs().max()) <= 1e-5
    assert tuple(b.shape) == (1, 3, 192, 256) and float((b - torch.from_numpy(g["rect256"])).abs().max()) <= 1e-5


def test_odd_grid_raises():
    _, cfg, w = _tiny(5)
    with pytest.raises(RuntimeError):
        dpt_oracle.forward(w, cfg, torch.zeros(1, 3, 48, 48))


def test_c_abi_inventory_for_beit():
    lib = native.load()
    c = BEIT_CONFIGS["beit_large_384"]
    s = native.MdptConfig()
    s.features_per_token, s.num_heads, s.num_blocks = c["features_per_token"], c["num_heads"], c["num_blocks"]
    for i, v in enumerate(c["reassembly_features_list"]):
        s.reassembly_features[i] = v
    s.base_patch_grid_h, s.base_patch_grid_w = c["base_patch_grid_hw"]
    s.fusion_channels, s.patch_size_px, s.precision, s.family = 256, 16, native.PREC_BF16, native.FAMILY_BEIT
    h = ctypes.c_void_p()
    assert lib.mdpt_create(ctypes.byref(s), ctypes.byref(h)) == 0, lib.mdpt_last_error()
    names = [lib.mdpt_weight_name(h, i).decode() for i in range(lib.mdpt_num_weights(h))]
    want = {f"{comp}.{k}" for comp, keys in conv.expected_new_keys(c).items() for k in keys}
    assert set(names) == want and len(names) == len(want)
    ndim, shape = ctypes.c_int32(), (ctypes.c_int64 * 4)()
    i = names.index("imgencoder.stages.2.blocks.1.attn.relpos_enc.ref_bias_lut")
    assert lib.mdpt_weight_shape(h, i, ctypes.byref(ndim), shape) == 0
    assert ndim.value == 2 and list(shape)[:2] == [47 * 47 + 3, 16]
    i = names.index("reassemble.spatial_upx4.readout_proj.1.weight")
    lib.mdpt_weight_shape(h, i, ctypes.byref(ndim), shape)
    assert list(shape)[:2] == [1024, 2048]
    ws = ctypes.c_size_t()
    assert lib.mdpt_workspace_bytes(h, 1, 384, 384, ctypes.byref(ws)) == 0 and ws.value > 0
    assert lib.mdpt_workspace_bytes(h, 1, 400, 400, ctypes.byref(ws)) == native.E_GRID  # 25x25 grid
    lib.mdpt_destroy(h)
    s.family = 9
    assert lib.mdpt_create(ctypes.byref(s), ctypes.byref(h)) < 0


def test_make_dpt_routes_beit(tmp_path):
    from muggled_dpt_amd import make_dpt_from_state_dict
    osd = make_synthetic_beit_state_dict("beit_tiny", 5)
    path = str(tmp_path / "dpt_beit_tiny_64.pt")
    torch.save(osd, path)
    cfg, model = make_dpt_from_state_dict(path)
    assert model.family == "beit" and cfg["num_heads"] == 2 and cfg["base_patch_grid_hw"] == (4, 4)
    assert model.patch_embed.rgb_offset == (0.5, 0.5, 0.5) and model.patch_embed._tiling_size == 32


def test_beit_large_fixture_is_self_consistent(golden_dir):
    g = np.load(os.path.join(golden_dir, "beit_large_384.npz"))
    assert g["depth_strided"].shape == (1, 96, 96) and g["depth_stats"][1] > 0
    assert stats(torch.from_numpy(g["tap0_crop"])).shape == (4,)


def test_relative_position_index_matches_the_reference_docstring_example():
    """The only known-answer data the reference itself carries for this path: the worked 2x3-grid example in
    v31_beit/components/relative_positional_encoder.py:229-236 (cls row / column / corner = 15 / 16 / 17)."""
    want = torch.tensor([[17, 15, 15, 15, 15, 15, 15],
                         [16, 7, 6, 5, 2, 1, 0],
                         [16, 8, 7, 6, 3, 2, 1],
                         [16, 9, 8, 7, 4, 3, 2],
                         [16, 12, 11, 10, 7, 6, 5],
                         [16, 13, 12, 11, 8, 7, 6],
                         [16, 14, 13, 12, 9, 8, 7]])
    assert torch.equal(dpt_oracle.beit_relative_position_index((2, 3)), want)
    # the device kernels use bias = lut[tq[q] - tk[k]] with tq = (y + gh - 1)(2gw - 1) + x + gw - 1, tk = y (2gw - 1) + x
    # (beit_relpos_kernel in csrc/elementwise.hip): the same decomposition reproduces the token-token block
    gh, gw = 2, 3
    ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
    tq = ((ys + gh - 1) * (2 * gw - 1) + xs + gw - 1).flatten()
    tk = (ys * (2 * gw - 1) + xs).flatten()
    assert torch.equal(tq[:, None] - tk[None, :], want[1:, 1:])

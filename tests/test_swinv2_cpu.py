"""CPU side of the MiDaS v3.1 SwinV2 family: oracle vs the fixtures generated from the reference (tests/golden/gen_golden.py),
checkpoint conversion contract, C-ABI inventory. No GPU compute."""
import ctypes
import json
import math
import os

import numpy as np
import pytest
import torch

from muggled_dpt_amd import native
from muggled_dpt_amd import state_dict_conversion_swinv2 as conv
from muggled_dpt_amd.state_dict_conversion import flatten_components
from muggled_dpt_amd.synthetic import SWINV2_CONFIGS, make_synthetic_swinv2_state_dict
from oracle import dpt_oracle

ATOL = 2e-5


def _tiny(seed):
    osd = make_synthetic_swinv2_state_dict("swin2_tiny", seed)
    cfg = conv.get_model_config_from_state_dict(osd)
    return osd, cfg, flatten_components(conv.convert_state_dict_keys(cfg, osd))


def test_config_sniffing_and_key_names_match_reference(golden_dir):
    osd, cfg, w = _tiny(6)
    want = dict(SWINV2_CONFIGS["swin2_tiny"], enable_cache=False, enable_optimizations=True)
    norm = lambda c: {k: (tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in c.items()}  # noqa: E731
    assert norm(cfg) == norm(want)
    with open(os.path.join(golden_dir, "swin2_tiny_new_keys.json")) as f:
        ref_keys = json.load(f)
    assert set(ref_keys) == set(w)
    for k, shp in ref_keys.items():
        assert list(w[k].shape) == shp, k
    # load-time transforms (convert_midas_state_dict_keys.py:115-161): exp of the clamped logit scale, [1,H,1,d] q/v biases
    raw = osd["pretrained.model.layers.1.blocks.0.attn.logit_scale"]
    assert torch.equal(w["imgencoder.stages.1.blocks.0.attn.logit_scale"], torch.clamp(raw, max=math.log(100.0)).exp())
    osd2 = dict(osd)
    osd2["pretrained.model.layers.1.blocks.0.attn.logit_scale"] = torch.full_like(raw, 9.0)
    w2 = flatten_components(conv.convert_state_dict_keys(cfg, osd2))
    assert float(w2["imgencoder.stages.1.blocks.0.attn.logit_scale"].max()) == pytest.approx(100.0, rel=1e-6)
    assert w["imgencoder.stages.3.blocks.1.attn.v_bias"].shape == (1, 16, 1, 32)
    # the large config resolves the pretrained window sizes through the reference's fixed table
    big = {k: torch.empty(0) for k in ()}
    assert conv._PRETRAINED_WINDOW_LUT[24] == [12, 12, 12, 6] and not big
    with pytest.raises(AssertionError):
        conv.get_model_config_from_state_dict({k: v for k, v in osd.items() if not k.endswith("attn_mask")})


@pytest.mark.parametrize("tag", ["base", "wide", "tall"])
def test_oracle_every_stage_boundary(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "swin2_tiny.npz"))
    _, cfg, w = _tiny(int(g["weight_seed"]))
    depth, st = dpt_oracle.forward(w, cfg, torch.from_numpy(g[f"{tag}_input"]), return_stages=True)
    assert float((depth - torch.from_numpy(g[f"{tag}_depth"])).abs().max()) <= ATOL
    assert float((st["patch_tokens"] - torch.from_numpy(g[f"{tag}_patch_tokens"])).abs().max()) <= ATOL
    for i in range(4):
        assert float((st["stages"][i] - torch.from_numpy(g[f"{tag}_tap{i}"])).abs().max()) <= ATOL
        assert float((st["reasm"][i] - torch.from_numpy(g[f"{tag}_reasm{i}"])).abs().max()) <= ATOL
    assert float((st["fused"] - torch.from_numpy(g[f"{tag}_fused"])).abs().max()) <= ATOL


def test_oracle_window_bookkeeping_known_answers(golden_dir):
    g = np.load(os.path.join(golden_dir, "swin2_tiny.npz"))
    n = 0
    for key in g.files:
        if key.startswith("winshift_"):
            grid, targ = key[len("winshift_"):].split("_t")
            gh, gw = (int(v) for v in grid.split("x"))
            win, shift = dpt_oracle.swin_window_and_shift((gh, gw), (int(targ), int(targ)))
            assert [*win, *shift] == list(g[key]), key
            n += 1
        elif key.startswith("mask_"):
            grid, win, shift = key[len("mask_"):].split("_")
            gh, gw = (int(v) for v in grid.split("x"))
            wh, ww = (int(v) for v in win[1:].split("x"))
            sh, sw = (int(v) for v in shift[1:].split("x"))
            assert torch.equal(dpt_oracle.swin_shift_mask((gh, gw), (wh, ww), (sh, sw)), torch.from_numpy(g[key])), key
            n += 1
    assert n >= 10
    # 18 does not tile by 4: closest divisor in [2, 8) is 6; a side no larger than the window is not shifted
    assert dpt_oracle.swin_window_and_shift((18, 30), (4, 4)) == ((6, 6), (3, 3))
    assert dpt_oracle.swin_window_and_shift((12, 12), (24, 24)) == ((12, 12), (0, 0))


def test_oracle_continuous_position_bias(golden_dir):
    g = np.load(os.path.join(golden_dir, "swin2_tiny.npz"))
    _, cfg, w = _tiny(int(g["weight_seed"]))
    for win in ((4, 4), (3, 1), (2, 6)):
        ref = torch.from_numpy(g[f"cpb_w{win[0]}x{win[1]}"])
        got = dpt_oracle.swin_cpb_bias(w, "imgencoder.stages.1.blocks.0.attn.relpos_enc", win, None, 4)
        assert got.shape == ref.shape and float((got - ref).abs().max()) <= 1e-5
    assert float(ref.min()) >= 0.0 and float(ref.max()) <= 16.0


def test_oracle_prepare_image_and_bad_grids(golden_dir):
    g = np.load(os.path.join(golden_dir, "swin2_prepare_image.npz"))
    _, cfg, w = _tiny(6)
    x = dpt_oracle.prepare_image(g["image"], 128, False, default_size_px=64, tiling_px=32, rgb_mean=(0.5,) * 3, rgb_std=(0.5,) * 3)
    assert tuple(x.shape) == (1, 3, 96, 128) and float((x - torch.from_numpy(g["rect128"])).abs().max()) <= 1e-5
    for bad in ((1, 3, 72, 72), (1, 3, 48, 64)):
        with pytest.raises(RuntimeError):
            dpt_oracle.forward(w, cfg, torch.zeros(*bad))


def test_c_abi_inventory_for_swinv2():
    lib = native.load()
    c = SWINV2_CONFIGS["swin2_large_384"]
    s = native.MdptConfig()
    s.features_per_token, s.num_heads, s.num_blocks = c["features_per_stage"][0], c["heads_per_stage"][0], sum(c["layers_per_stage"])
    for i in range(4):
        s.reassembly_features[i] = c["features_per_stage"][i]
        s.swin_heads[i], s.swin_layers[i] = c["heads_per_stage"][i], c["layers_per_stage"][i]
        s.swin_pretrained_window[i] = c["pretrained_window_sizes_per_stage"][i]
    s.swin_window_h, s.swin_window_w = c["window_size_hw"]
    s.base_patch_grid_h, s.base_patch_grid_w = c["base_patch_grid_hw"]
    s.fusion_channels, s.patch_size_px, s.precision, s.family = 256, 4, native.PREC_BF16, native.FAMILY_SWINV2
    h = ctypes.c_void_p()
    assert lib.mdpt_create(ctypes.byref(s), ctypes.byref(h)) == 0, lib.mdpt_last_error()
    names = [lib.mdpt_weight_name(h, i).decode() for i in range(lib.mdpt_num_weights(h))]
    want = {f"{comp}.{k}" for comp, keys in conv.expected_new_keys(c).items() for k in keys}
    assert set(names) == want and len(names) == len(want)
    ndim, shape = ctypes.c_int32(), (ctypes.c_int64 * 4)()
    lib.mdpt_weight_shape(h, names.index("imgencoder.patch_merge_layers.1.reduction.weight"), ctypes.byref(ndim), shape)
    assert ndim.value == 2 and list(shape)[:2] == [768, 1536]
    lib.mdpt_weight_shape(h, names.index("imgencoder.stages.2.blocks.17.attn.relpos_enc.bias_mlp.2.weight"), ctypes.byref(ndim), shape)
    assert list(shape)[:2] == [24, 512]
    ws = ctypes.c_size_t()
    assert lib.mdpt_workspace_bytes(h, 1, 384, 384, ctypes.byref(ws)) == 0 and ws.value > 0
    assert lib.mdpt_workspace_bytes(h, 1, 400, 400, ctypes.byref(ws)) == native.E_GRID  # grid 100: not divisible by 8
    lib.mdpt_destroy(h)
    s.swin_heads[2] = 12  # head dim 64: not a SwinV2 configuration this build covers
    assert lib.mdpt_create(ctypes.byref(s), ctypes.byref(h)) < 0 and b"head dim" in lib.mdpt_last_error()
    s.swin_heads[2] = 24
    s.swin_layers[1] = 3
    assert lib.mdpt_create(ctypes.byref(s), ctypes.byref(h)) < 0


def test_make_dpt_routes_swinv2(tmp_path):
    from muggled_dpt_amd import make_dpt_from_state_dict
    osd = make_synthetic_swinv2_state_dict("swin2_tiny", 6)
    path = str(tmp_path / "dpt_swin2_tiny_64.pt")
    torch.save(osd, path)
    cfg, model = make_dpt_from_state_dict(path)
    assert model.family == "swinv2" and cfg["window_size_hw"] == (4, 4) and cfg["base_patch_grid_hw"] == (16, 16)
    assert model.patch_embed._tiling_size == 32 and model.patch_embed._default_size_px == 64
    assert hasattr(model.reassemble, "spatial_downx8") and hasattr(model.imgencoder, "patch_merge_layers")


def test_relative_position_index_matches_the_reference_docstring_example():
    """Worked 2x3-window example of v31_swinv2/components/relative_positional_encoder.py:262-275: shifted/scaled y offsets plus
    shifted x offsets (the two matrices printed there) sum to the index."""
    ypart = torch.tensor([[5, 5, 5, 0, 0, 0]] * 3 + [[10, 10, 10, 5, 5, 5]] * 3)
    xpart = torch.tensor([[2, 1, 0, 2, 1, 0], [3, 2, 1, 3, 2, 1], [4, 3, 2, 4, 3, 2]] * 2)
    assert torch.equal(dpt_oracle.swin_relative_position_index((2, 3)), ypart + xpart)
    # decomposition used on the device (swin_window_map_kernel in csrc/swin.hip): index = tq[i] - tk[j]
    wh, ww = 2, 3
    iy, ix = torch.meshgrid(torch.arange(wh), torch.arange(ww), indexing="ij")
    tq = ((iy + wh - 1) * (2 * ww - 1) + ix + ww - 1).flatten()
    tk = (iy * (2 * ww - 1) + ix).flatten()
    assert torch.equal(tq[:, None] - tk[None, :], ypart + xpart)


def test_swin2_tiny_256_oracle_vs_reference_fixture_and_abi(golden_dir):
    """swin2_tiny_256 (96-wide first stage): the oracle reproduces the reference fixture; the C ABI accepts widths that are multiples
    of 32 (head dim) and still rejects others."""
    import numpy as np
    from oracle import dpt_oracle
    from muggled_dpt_amd.synthetic import make_synthetic_swinv2_state_dict
    from muggled_dpt_amd.state_dict_conversion import flatten_components
    g = np.load(os.path.join(golden_dir, "swin2_tiny_256.npz"))
    osd = make_synthetic_swinv2_state_dict("swin2_tiny_256", int(g["weight_seed"]))
    cfg = conv.get_model_config_from_state_dict(osd)
    assert list(cfg["features_per_stage"]) == [96, 192, 384, 768] and list(cfg["heads_per_stage"]) == [3, 6, 12, 24]
    w = flatten_components(conv.convert_state_dict_keys(cfg, osd))
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(int(g["input_seed"])))
    y = dpt_oracle.forward(w, cfg, x)
    assert float((y[:, ::4, ::4] - torch.from_numpy(g["depth_strided"])).abs().max()) <= 1e-5
    lib = native.load()
    s = native.MdptConfig()
    c = SWINV2_CONFIGS["swin2_tiny_256"]
    s.features_per_token, s.num_heads, s.num_blocks = 96, 3, sum(c["layers_per_stage"])
    for i in range(4):
        s.reassembly_features[i] = c["features_per_stage"][i]
        s.swin_heads[i], s.swin_layers[i] = c["heads_per_stage"][i], c["layers_per_stage"][i]
        s.swin_pretrained_window[i] = c["pretrained_window_sizes_per_stage"][i]
    s.swin_window_h, s.swin_window_w = c["window_size_hw"]
    s.base_patch_grid_h, s.base_patch_grid_w = c["base_patch_grid_hw"]
    s.fusion_channels, s.patch_size_px, s.precision, s.family = 256, 4, native.PREC_BF16, native.FAMILY_SWINV2
    h = ctypes.c_void_p()
    assert lib.mdpt_create(ctypes.byref(s), ctypes.byref(h)) == 0, lib.mdpt_last_error()
    lib.mdpt_destroy(h)
    s.features_per_token = s.reassembly_features[0] = 80
    s.swin_heads[0] = 2
    assert lib.mdpt_create(ctypes.byref(s), ctypes.byref(h)) < 0

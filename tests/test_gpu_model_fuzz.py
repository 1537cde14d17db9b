"""Randomised small Depth-Anything-V2 configurations (widths, head counts incl. odd ones, block counts, reassembly widths, fusion channels,
rectangular grids, batch sizes) through the whole HIP path vs the CPU oracle, fp32-class mode at the north-star tolerance and bf16 mode at
its own. Odd head counts make QKV tiles straddle the Q|K|V boundaries; widths that are not multiples of 64/256 exercise every padding rule.
`pytest -m gpu`."""
import numpy as np
import pytest
import torch

from tests.helpers import emulated_tol, rel_err

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(20260928)
    out = []
    for k in range(10):
        heads = int(rng.integers(1, 9))                      # 1..8 heads of 64
        F = 64 * heads
        blocks = int(rng.choice([4, 8]))
        feats = [int(16 * rng.integers(1, 9)) for _ in range(4)]
        C = int(rng.choice([32, 48, 64, 96]))
        gh, gw = int(2 * rng.integers(1, 8)), int(2 * rng.integers(1, 8))
        B = int(rng.integers(1, 6))
        giant = bool(k % 5 == 4)
        out.append((dict(features_per_token=F, num_heads=heads, num_blocks=blocks, reassembly_features_list=feats,
                         base_patch_grid_hw=(5, 5), fusion_channels=C, patch_size_px=14, is_giant=giant), (gh, gw), B, k))
    return out


@pytest.mark.parametrize("cfg,grid,B,seed", _cases(), ids=lambda v: None)
def test_random_configurations_match_the_oracle(cfg, grid, B, seed):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components, get_model_config_from_state_dict
    from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
    from oracle import dpt_oracle
    osd = make_synthetic_original_state_dict(cfg, seed)
    c = get_model_config_from_state_dict(osd)
    w = flatten_components(convert_state_dict_keys(c, osd))
    x = torch.randn(B, 3, grid[0] * 14, grid[1] * 14, generator=torch.Generator().manual_seed(100 + seed))
    ref = dpt_oracle.forward(w, c, x)
    for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, emulated_tol(w, c, x))):  # bf16: 1.5 x an independent CPU emulation of the same rounding on this input
        _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
        y = model.to("cuda", dtype)(x.to("cuda", dtype))
        assert tuple(y.shape) == tuple(ref.shape)
        e = rel_err(y.float().cpu(), ref)
        assert e <= tol, f"{cfg} grid {grid} B={B} {dtype}: rel err {e:.3e} > {tol}"


def _beit_cases():
    rng = np.random.default_rng(77)
    out = []
    for k in range(6):
        heads = int(rng.integers(1, 7))
        base = int(rng.choice([4, 6, 8]))
        out.append((dict(features_per_token=64 * heads, num_heads=heads, num_blocks=int(rng.choice([4, 8])),
                         reassembly_features_list=[int(16 * rng.integers(1, 7)) for _ in range(4)], base_patch_grid_hw=(base, base),
                         fusion_channels=int(rng.choice([32, 64])), patch_size_px=16),
                    (int(2 * rng.integers(1, 7)), int(2 * rng.integers(1, 7))), int(rng.integers(1, 4)), k))
    return out


@pytest.mark.parametrize("cfg,grid,B,seed", _beit_cases(), ids=lambda v: None)
def test_random_beit_configurations_match_the_oracle(cfg, grid, B, seed):
    """MiDaS v3.1 BEiT: random widths / head counts / base grids (the learned relative-position table is resized to every grid)."""
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from muggled_dpt_amd import make_beit_dpt_from_midas_v31_state_dict
    from muggled_dpt_amd import state_dict_conversion_beit as conv
    from muggled_dpt_amd.state_dict_conversion import flatten_components
    from muggled_dpt_amd.synthetic import make_synthetic_beit_state_dict
    from oracle import dpt_oracle
    osd = make_synthetic_beit_state_dict(cfg, seed)
    c, _ = make_beit_dpt_from_midas_v31_state_dict(osd)
    w = flatten_components(conv.convert_state_dict_keys(c, osd))
    x = torch.randn(B, 3, grid[0] * 16, grid[1] * 16, generator=torch.Generator().manual_seed(200 + seed))
    ref = dpt_oracle.forward(w, c, x)
    for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, emulated_tol(w, c, x))):  # bf16: 1.5 x an independent CPU emulation of the same rounding on this input
        _, model = make_beit_dpt_from_midas_v31_state_dict(osd)
        y = model.to("cuda", dtype)(x.to("cuda", dtype))
        e = rel_err(y.float().cpu(), ref)
        assert e <= tol, f"{cfg} grid {grid} B={B} {dtype}: rel err {e:.3e} > {tol}"


def _swin_cases():
    rng = np.random.default_rng(99)
    out = []
    for k in range(5):
        h0 = int(rng.integers(1, 4))                      # heads of stage 0 (head dim 32), doubling per stage
        feats = [32 * h0 * (1 << s) for s in range(4)]
        win = int(rng.choice([4, 8]))
        base = int(rng.choice([16, 32]))
        pre = [None] * 4 if k % 2 == 0 else [win, win, win, max(win // 2, 1)]
        gh, gw = int(8 * rng.integers(1, 5)), int(8 * rng.integers(1, 5))
        out.append((dict(features_per_stage=feats, heads_per_stage=[h0 * (1 << s) for s in range(4)], layers_per_stage=[2, 2, int(rng.choice([2, 4])), 2],
                         base_patch_grid_hw=(base, base), window_size_hw=(win, win), pretrained_window_sizes_per_stage=pre,
                         fusion_channels=int(rng.choice([32, 64])), patch_size_px=4), (gh, gw), int(rng.integers(1, 4)), k))
    return out


@pytest.mark.parametrize("cfg,grid,B,seed", _swin_cases(), ids=lambda v: None)
def test_random_swinv2_configurations_match_the_oracle(cfg, grid, B, seed):
    """MiDaS v3.1 SwinV2: random widths / head counts / window sizes / pretrained-window settings on rectangular grids (windows are
    re-fitted and shifts re-derived per stage and grid)."""
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from muggled_dpt_amd import make_swinv2_dpt_from_midas_v31_state_dict
    from muggled_dpt_amd import state_dict_conversion_swinv2 as conv
    from muggled_dpt_amd.state_dict_conversion import flatten_components
    from muggled_dpt_amd.synthetic import make_synthetic_swinv2_state_dict
    from oracle import dpt_oracle
    osd = make_synthetic_swinv2_state_dict(cfg, seed)
    c, _ = make_swinv2_dpt_from_midas_v31_state_dict(osd)
    w = flatten_components(conv.convert_state_dict_keys(c, osd))
    x = torch.randn(B, 3, grid[0] * 4, grid[1] * 4, generator=torch.Generator().manual_seed(300 + seed))
    ref = dpt_oracle.forward(w, c, x)
    for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, emulated_tol(w, c, x, factor=2.0))):  # bf16: 2 x an independent CPU emulation of the same rounding on this input
        # (cosine attention at logit scale ~10 makes the toy maps chaotic: the emulation itself moves by +-30 % with the host's summation order, and it does not round the attention operands)
        _, model = make_swinv2_dpt_from_midas_v31_state_dict(osd)
        y = model.to("cuda", dtype)(x.to("cuda", dtype))
        e = rel_err(y.float().cpu(), ref)
        assert e <= tol, f"{cfg} grid {grid} B={B} {dtype}: rel err {e:.3e} > {tol}"

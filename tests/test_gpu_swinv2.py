"""MiDaS v3.1 SwinV2 family on the HIP path vs the fixtures generated from the reference (tests/golden/gen_golden.py) and the
oracle. Run with `pytest -m gpu` on an MI355X. Tolerances as in test_gpu_parity.py."""
import os

import numpy as np
import pytest
import torch

from tests.helpers import REL_TOL_BF16_SWIN, REL_TOL_BF16_SWIN_TOY, REL_TOL_X3, record_err, rel_err, seeded_input, stats

pytestmark = pytest.mark.gpu

MODES = [(torch.float32, REL_TOL_X3), (torch.bfloat16, REL_TOL_BF16_SWIN)]  # 1.25 x the reference's own bf16 error (tests/helpers.py)
MODES_TOY = [(torch.float32, REL_TOL_X3), (torch.bfloat16, REL_TOL_BF16_SWIN_TOY)]  # toy configs, see tests/helpers.py


def _build(name, seed, dtype):
    from muggled_dpt_amd import make_swinv2_dpt_from_midas_v31_state_dict
    from muggled_dpt_amd import state_dict_conversion_swinv2 as conv
    from muggled_dpt_amd.state_dict_conversion import flatten_components
    from muggled_dpt_amd.synthetic import make_synthetic_swinv2_state_dict
    osd = make_synthetic_swinv2_state_dict(name, seed)
    cfg, model = make_swinv2_dpt_from_midas_v31_state_dict(osd)
    w = flatten_components(conv.convert_state_dict_keys(cfg, osd))
    return model.to("cuda", dtype), cfg, w


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from muggled_dpt_amd import native
    native.load()


@pytest.mark.parametrize("dtype,tol", MODES_TOY)
@pytest.mark.parametrize("tag", ["base", "wide", "tall"])
def test_swin_tiny_every_stage_boundary_vs_golden(golden_dir, tag, dtype, tol):
    """Shifted 4x4 windows with masks in stages 0/1, a single unshifted window in stage 2, shrunken windows (2x2, 2x3, 3x1)
    in stage 3; 'tall' shifts along one axis only in stage 1 (grid 12x4)."""
    g = np.load(os.path.join(golden_dir, "swin2_tiny.npz"))
    model, cfg, w = _build("swin2_tiny", int(g["weight_seed"]), dtype)
    x = torch.from_numpy(g[f"{tag}_input"])
    y = model(x.to("cuda", dtype))
    assert y.dtype == dtype and tuple(y.shape) == (x.shape[0], x.shape[2], x.shape[3])
    taps = model.debug_taps(x.shape[0], tuple(x.shape[2:]))
    for i in range(4):
        assert rel_err(taps["stages"][i].cpu(), torch.from_numpy(g[f"{tag}_tap{i}"])) <= tol, f"tap{i}"
        assert rel_err(taps["reasm"][i].cpu(), torch.from_numpy(g[f"{tag}_reasm{i}"])) <= tol, f"reasm{i}"
    assert rel_err(taps["fused"].cpu(), torch.from_numpy(g[f"{tag}_fused"])) <= tol
    assert rel_err(y.float().cpu(), torch.from_numpy(g[f"{tag}_depth"])) <= tol  # measured: bf16 1.0e-2 ... 1.5e-2


@pytest.mark.parametrize("dtype,tol", MODES_TOY)
def test_swin_stage_entry_points(golden_dir, dtype, tol):
    g = np.load(os.path.join(golden_dir, "swin2_tiny.npz"))
    model, cfg, w = _build("swin2_tiny", int(g["weight_seed"]), dtype)
    dev = lambda k: torch.from_numpy(g[k]).to("cuda", dtype)  # noqa: E731
    tok, hw = model.patch_embed(dev("wide_input"))
    assert tuple(hw) == (16, 24) and rel_err(tok.float().cpu(), torch.from_numpy(g["wide_patch_tokens"])) <= tol
    taps = model.imgencoder(dev("wide_patch_tokens"), (16, 24))
    for i in range(4):
        assert tuple(taps[i].shape) == tuple(g[f"wide_tap{i}"].shape)
        assert rel_err(taps[i].float().cpu(), torch.from_numpy(g[f"wide_tap{i}"])) <= tol, f"tap{i}"
    reasm = model.reassemble(*[dev(f"wide_tap{i}") for i in range(4)], (16, 24))
    for i in range(4):
        assert rel_err(reasm[i].float().cpu(), torch.from_numpy(g[f"wide_reasm{i}"])) <= tol, f"reasm{i}"
    fused = model.fusion(*[dev(f"wide_reasm{i}") for i in range(4)])
    assert rel_err(fused.float().cpu(), torch.from_numpy(g["wide_fused"])) <= tol
    depth = model.head(dev("wide_fused"))
    assert rel_err(depth.float().cpu(), torch.from_numpy(g["wide_depth"])) <= tol


def test_swin_bad_grid_raises_and_inference(golden_dir):
    g = np.load(os.path.join(golden_dir, "swin2_prepare_image.npz"))
    model, cfg, w = _build("swin2_tiny", 6, torch.float32)
    for bad in ((1, 3, 72, 72), (1, 3, 48, 64)):
        with pytest.raises(RuntimeError):
            model(torch.zeros(*bad, device="cuda"))
    x = model.prepare_image_bgr(g["image"], 128, False)
    assert tuple(x.shape) == (1, 3, 96, 128) and float((x.cpu() - torch.from_numpy(g["rect128"])).abs().max()) <= 2e-5
    from oracle import dpt_oracle
    d = model.inference(g["image"], 128, True)
    assert tuple(d.shape) == (1, 128, 128)
    assert rel_err(d.cpu(), dpt_oracle.inference(w, cfg, g["image"], 128, True)) <= REL_TOL_X3


def test_swin_window_that_does_not_tile_the_grid():
    """grid 24x40 with target window 16: neither side tiles; 24 -> window 24 (the divisor in [8, 32) closest to the side: one
    unshifted window), 40 -> window 20 shifted by 10, i.e. shift along one axis only; checked against the oracle."""
    from muggled_dpt_amd import make_swinv2_dpt_from_midas_v31_state_dict
    from muggled_dpt_amd import state_dict_conversion_swinv2 as conv
    from muggled_dpt_amd.state_dict_conversion import flatten_components
    from muggled_dpt_amd.synthetic import SWINV2_CONFIGS, make_synthetic_swinv2_state_dict
    from oracle import dpt_oracle
    cfg0 = dict(SWINV2_CONFIGS["swin2_tiny"], window_size_hw=(16, 16), base_patch_grid_hw=(32, 32), pretrained_window_sizes_per_stage=[16, 16, 16, 8])
    osd = make_synthetic_swinv2_state_dict(cfg0, 2)
    cfg, model = make_swinv2_dpt_from_midas_v31_state_dict(osd)
    assert cfg["pretrained_window_sizes_per_stage"] == [16, 16, 16, 8]
    w = flatten_components(conv.convert_state_dict_keys(cfg, osd))
    x = seeded_input((1, 3, 96, 160), 3)
    assert dpt_oracle.swin_window_and_shift((24, 40), (16, 16)) == ((24, 20), (0, 10))
    assert rel_err(model.to("cuda")(x.to("cuda")).cpu(), dpt_oracle.forward(w, cfg, x)) <= REL_TOL_X3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hw", [(384, 384), (384, 416)])
def test_swin_fused_qk_epilogue_is_bitwise_identical_to_the_prep_kernel(dtype, hw):
    """With the 8-phase GEMM tile the cosine-attention Q / K / V operands come straight out of the QKV GEMM's accumulators
    (gemm.hip epilogue_swin_qk / epilogue_swin_vt); every other tile writes fp32 QKV rows and runs swin_qk_prep / swin_v_prep. Same
    arithmetic in the same order: the depth maps must be the same bits whichever form each block took (the tile rule makes that depend
    on the batch size). swin2_base widths (128 .. 1024) put every stage on the fused form when tile 5 is forced; shifted and unshifted
    blocks both run. 384x416: a 96x104 grid whose window width becomes 26 (shift 13) - token runs of 4 do not stay together, so V takes
    the fp32-rows + swin_v_prep route while Q / K stay fused."""
    model, cfg, w = _build("swin2_base_384", 3, dtype)
    x = seeded_input((2, 3, *hw), 9).to("cuda", dtype)
    model.set_gemm_tile(1)   # 128x128 lockstep everywhere: unfused
    y_unfused = model(x)
    model.set_gemm_tile(5)   # 8-phase everywhere it can run: fused
    y_fused = model(x)
    model.set_gemm_tile(0)   # the shipped rule: a mix
    y_auto = model(x)
    assert torch.isfinite(y_fused.float()).all() and float(y_fused.float().abs().max()) > 0
    assert torch.equal(y_fused, y_unfused), float((y_fused.float() - y_unfused.float()).abs().max())
    assert torch.equal(y_auto, y_unfused)


@pytest.mark.parametrize("dtype,tol", MODES)
def test_swin_large_384_vs_golden_fixture(golden_dir, dtype, tol):
    """BASELINE.json configs[5]: SwinV2-L @384, batch 1 (compact fixture: strided depth, crops, per-boundary stats)."""
    from muggled_dpt_amd.synthetic import make_synthetic_swinv2_state_dict
    g = np.load(os.path.join(golden_dir, "swin2_large_384.npz"))
    osd = make_synthetic_swinv2_state_dict("swin2_large_384", int(g["weight_seed"]))
    np.testing.assert_allclose(float(osd["pretrained.model.layers.2.blocks.3.attn.qkv.weight"].double().sum()), g["weight_checksum"][0], rtol=1e-9)
    del osd
    model, cfg, w = _build("swin2_large_384", int(g["weight_seed"]), dtype)
    x = seeded_input((1, 3, 384, 384), int(g["input_seed"]))
    np.testing.assert_allclose(float(x.double().sum()), g["input_checksum"][0], rtol=1e-9)
    y = model(x.to("cuda", dtype)).float().cpu()
    ref = torch.from_numpy(g["depth_strided"])
    assert record_err(float((y[:, ::4, ::4].double() - ref.double()).abs().max() / ref.abs().max())) <= tol
    taps = model.debug_taps(1, (384, 384))
    for i in range(4):
        crop = torch.from_numpy(g[f"tap{i}_crop"])
        scale = float(g[f"tap{i}_stats"][1] - g[f"tap{i}_stats"][0])
        assert record_err(float((taps["stages"][i][:, :64, :64].cpu() - crop).abs().max()) / scale) <= tol, f"tap{i}"
    if dtype == torch.float32:
        np.testing.assert_allclose(stats(y)[3], g["depth_stats"][3], rtol=1e-3)


def test_swin_batch_is_independent_of_batch_composition():
    model, cfg, w = _build("swin2_base_384", 2, torch.bfloat16)
    x = seeded_input((3, 3, 192, 256), 5).to("cuda", torch.bfloat16)
    y = model(x)
    for i in range(3):
        assert torch.equal(y[i:i + 1], model(x[i:i + 1])), i
    assert torch.isfinite(y.float()).all() and float(y.float().max()) > 0


@pytest.mark.parametrize("dtype,tol", MODES)
def test_swin_tiny_256_vs_golden_fixture(golden_dir, dtype, tol):
    """The smallest MiDaS v3.1 SwinV2 (reference make_swinv2_dpt.py:107-115): 96-wide first stage, 3 heads - bf16 operand planes of that
    stage are padded to 128 columns on the device. Fixture generated from the reference (tests/golden/gen_golden.py --only-swin-tiny256)."""
    from muggled_dpt_amd.synthetic import make_synthetic_swinv2_state_dict
    g = np.load(os.path.join(golden_dir, "swin2_tiny_256.npz"))
    osd = make_synthetic_swinv2_state_dict("swin2_tiny_256", int(g["weight_seed"]))
    np.testing.assert_allclose(float(osd["pretrained.model.layers.0.blocks.1.attn.qkv.weight"].double().sum()), g["weight_checksum"][0], rtol=1e-9)
    del osd
    model, cfg, w = _build("swin2_tiny_256", int(g["weight_seed"]), dtype)
    x = seeded_input((2, 3, 256, 256), int(g["input_seed"]))
    np.testing.assert_allclose(float(x.double().sum()), g["input_checksum"][0], rtol=1e-9)
    y = model(x.to("cuda", dtype)).float().cpu()
    ref = torch.from_numpy(g["depth_strided"])
    assert record_err(float((y[:, ::4, ::4].double() - ref.double()).abs().max() / ref.abs().max())) <= tol
    taps = model.debug_taps(2, (256, 256))
    for i in range(4):
        crop = torch.from_numpy(g[f"tap{i}_crop"])
        scale = float(g[f"tap{i}_stats"][1] - g[f"tap{i}_stats"][0])
        assert record_err(float((taps["stages"][i][:, :64, :64].cpu() - crop).abs().max()) / scale) <= tol, f"tap{i}"
    if dtype == torch.float32:
        np.testing.assert_allclose(stats(y)[3], g["depth_stats"][3], rtol=1e-3)
    # stage-by-stage entry points carry the padded planes too (tokens -> encoder -> reassemble)
    tok, hw = model.patch_embed(x[:1].to("cuda", dtype))
    maps = model.reassemble(*model.imgencoder(tok, hw), hw)
    assert rel_err(model.head(model.fusion(*maps)).float().cpu(), y[:1]) <= (1e-5 if dtype == torch.float32 else tol)


@pytest.mark.parametrize("dtype,tol", MODES)
def test_swin_large_384_batch16(golden_dir, dtype, tol):
    """BASELINE.json configs[4] at its full batch: 16 images at 384x384. Row 0 is the fixture's image (checked against the
    reference-generated fixture); rows 0, 7 and 15 must equal their batch-of-1 results bit for bit (batch invariance = what data-parallel
    sharding relies on); every map finite and non-trivial."""
    g = np.load(os.path.join(golden_dir, "swin2_large_384.npz"))
    model, cfg, w = _build("swin2_large_384", int(g["weight_seed"]), dtype)
    x = torch.cat([seeded_input((1, 3, 384, 384), int(g["input_seed"])), seeded_input((15, 3, 384, 384), 77)]).to("cuda", dtype)
    y = model(x)
    assert tuple(y.shape) == (16, 384, 384) and y.dtype == dtype and bool(torch.isfinite(y.float()).all())
    ref = torch.from_numpy(g["depth_strided"])
    assert record_err(float((y[:1, ::4, ::4].float().cpu().double() - ref.double()).abs().max() / ref.abs().max())) <= tol
    for i in (0, 7, 15):
        assert torch.equal(model(x[i:i + 1])[0], y[i]), f"image {i}: batch-of-1 result differs from its row in the batch of 16"
    assert float(y.float().amax(dim=(1, 2)).min()) > 0, "a degenerate (all-zero) map in the batch"

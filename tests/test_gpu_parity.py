"""Parity of the HIP path (called through the C ABI via the Python facade) against the CPU oracle and the committed
golden fixtures. Run with `pytest -m gpu` on an MI355X.

Tolerances (rel = max|y - ref| / max|ref|, the north-star metric) live in tests/helpers.py:
  * float32 model  = MDPT_PREC_BF16X3 (split-bf16 MFMA, fp32 accumulate): REL_TOL_X3 = 1e-4  (north-star bar 1e-3; measured 2-3.5e-5)
  * bfloat16 model = MDPT_PREC_BF16 (single-pass bf16 MFMA):             REL_TOL_BF16 = min(1.25 x the reference's own bf16 error on the
    same fixture, 1.3 x the worst error this code measured) (PyTorch's own bf16 CPU path is 2.1e-2 off its fp32 path on the same weights,
    tests/golden/reference_lowprec_errors.json - a pure-bf16 pipeline cannot meet 1e-3)
Every error a test measures is written to gpurun_out/parity_report.json (tests/conftest.py).
"""
import os

import numpy as np
import pytest
import torch

from tests.helpers import REL_TOL_BF16, REL_TOL_BF16_TOY, REL_TOL_X3, record_err, ref_lowprec_tol, rel_err, seeded_input, synthetic_model

pytestmark = pytest.mark.gpu

MODES = [(torch.float32, REL_TOL_X3), (torch.bfloat16, REL_TOL_BF16)]
MODES_TOY = [(torch.float32, REL_TOL_X3), (torch.bfloat16, REL_TOL_BF16_TOY)]  # toy configs, see tests/helpers.py


def _oracle():
    from oracle import dpt_oracle
    return dpt_oracle


def _model(name, dtype, seed=0):
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    osd, cfg, w = synthetic_model(name, seed)
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    return model.to("cuda", dtype), cfg, w


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from muggled_dpt_amd import native
    native.load()  # loud failure if libmdpt.so is missing: there is no fallback path


@pytest.mark.parametrize("dtype,tol", MODES_TOY)
def test_tiny_every_stage_boundary_vs_golden(golden_dir, dtype, tol):
    g = np.load(os.path.join(golden_dir, "tiny_full.npz"))
    model, cfg, w = _model("tiny", dtype)
    x = torch.from_numpy(g["input"])
    y = model(x.to("cuda", dtype))
    assert y.dtype == dtype and y.device.type == "cuda" and tuple(y.shape) == (2, 56, 56)
    assert rel_err(y.float().cpu(), torch.from_numpy(g["depth"])) <= tol
    taps = model.debug_taps(2, (56, 56))
    for i in range(4):
        assert rel_err(taps["stages"][i].cpu(), torch.from_numpy(g[f"tap{i}"])) <= tol, f"tap{i}"
        assert rel_err(taps["reasm"][i].cpu(), torch.from_numpy(g[f"reasm{i}"])) <= tol, f"reasm{i}"
    assert rel_err(taps["fused"].cpu(), torch.from_numpy(g["fused"])) <= tol


@pytest.mark.parametrize("dtype,tol", MODES_TOY)
def test_stage_entry_points_match_reference_submodule_calls(golden_dir, dtype, tol):
    """simple_examples/internal_features.py:39-45 usage: each sub-module called on the previous stage's (golden) output."""
    g = np.load(os.path.join(golden_dir, "tiny_full.npz"))
    model, cfg, w = _model("tiny", dtype)
    dev = lambda a: torch.from_numpy(np.asarray(a)).to("cuda", dtype)  # noqa: E731
    tok, hw = model.patch_embed(dev(g["input"]))
    assert tuple(hw) == tuple(g["grid_hw"]) and rel_err(tok.float().cpu(), torch.from_numpy(g["patch_tokens"])) <= tol
    enc = model.imgencoder(dev(g["patch_tokens"]), hw)
    for i in range(4):
        assert rel_err(enc[i].float().cpu(), torch.from_numpy(g[f"tap{i}"])) <= tol
    rs = model.reassemble(*[dev(g[f"tap{i}"]) for i in range(4)], hw)
    for i in range(4):
        assert tuple(rs[i].shape) == tuple(g[f"reasm{i}"].shape)
        assert rel_err(rs[i].float().cpu(), torch.from_numpy(g[f"reasm{i}"])) <= tol
    fu = model.fusion(*[dev(g[f"reasm{i}"]) for i in range(4)])
    assert rel_err(fu.float().cpu(), torch.from_numpy(g["fused"])) <= tol
    hd = model.head(dev(g["fused"]))
    assert rel_err(hd.float().cpu(), torch.from_numpy(g["depth"])) <= tol


@pytest.mark.parametrize("dtype,tol", MODES_TOY)
def test_rectangular_grid_and_odd_batch(golden_dir, dtype, tol):
    g = np.load(os.path.join(golden_dir, "tiny_rect.npz"))
    model, cfg, w = _model("tiny", dtype)
    y = model(torch.from_numpy(g["input"]).to("cuda", dtype))
    assert rel_err(y.float().cpu(), torch.from_numpy(g["depth"])) <= tol
    x3 = seeded_input((3, 3, 84, 56), 5)  # batch 3, grid 6x4
    assert rel_err(model(x3.to("cuda", dtype)).float().cpu(), _oracle().forward(w, cfg, x3)) <= tol


def test_odd_patch_grid_raises_like_the_reference():
    model, _, _ = _model("tiny", torch.float32)
    with pytest.raises(RuntimeError):
        model(torch.randn(1, 3, 42, 42, device="cuda"))  # 3x3 grid: reference dies at fusion_model.py:151
    with pytest.raises(AssertionError):
        model.verify_input(torch.randn(1, 3, 50, 56, device="cuda"))
    assert model.verify_input(torch.randn(1, 3, 56, 56, device="cuda")) is True


def test_cpu_tensor_or_cpu_model_fails_loudly():
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    osd, _, _ = synthetic_model("tiny", 0)
    _, cpu_model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cpu_model(torch.randn(1, 3, 56, 56))
    with pytest.raises(RuntimeError):
        cpu_model.to("cuda")(torch.randn(1, 3, 56, 56))  # input left on the CPU


@pytest.mark.parametrize("dtype,tol", MODES)
def test_vits_504_vs_golden_fixture_and_inference(golden_dir, dtype, tol):
    """BASELINE configs[1]/[0] shapes: ViT-S, 504x504 tensor, batch 1; plus inference() on a 518x518 uint8 image."""
    g = np.load(os.path.join(golden_dir, "vits504.npz"))
    model, cfg, w = _model("vits", dtype, int(g["weight_seed"]))
    x = seeded_input((1, 3, 504, 504), int(g["input_seed"]))
    y = model(x.to("cuda", dtype)).float().cpu()
    ref_max = float(g["depth_stats"][1])
    assert record_err(float((y[:, ::4, ::4].double() - torch.from_numpy(g["depth_strided"]).double()).abs().max()) / ref_max) <= tol
    assert record_err(float((y[:, 200:264, 100:164].double() - torch.from_numpy(g["depth_crop"]).double()).abs().max()) / ref_max) <= tol
    taps = model.debug_taps(1, (504, 504))
    for i in range(4):
        a, b = taps["stages"][i][:, :64, :64].cpu().double(), torch.from_numpy(g[f"tap{i}_crop"]).double()
        assert record_err(float((a - b).abs().max()) / max(abs(g[f"tap{i}_stats"][0]), abs(g[f"tap{i}_stats"][1]))) <= tol
    img = np.random.default_rng(1).integers(0, 256, (518, 518, 3), dtype=np.uint8)
    d = model.inference(img)
    assert tuple(d.shape) == (1, 504, 504) and d.dtype == dtype
    # preprocessing runs in the model dtype (reference patch_embed.py:133): allow the bf16 input rounding on top
    # (1.25 x the reference's own bf16 error on this configuration with model AND image cast to bfloat16, which is what its inference() does)
    ptol = tol if dtype == torch.float32 else ref_lowprec_tol("vits504")
    assert record_err(float((d.float().cpu()[:, ::4, ::4].double() - torch.from_numpy(g["inference518_strided"]).double()).abs().max()) / float(g["inference518_stats"][1])) <= ptol


def test_vitl_504_vs_golden_fixture(golden_dir):
    """Headline model (BASELINE configs[2]), one image, both modes, against the reference-generated fixture."""
    g = np.load(os.path.join(golden_dir, "vitl504.npz"))
    x = seeded_input((1, 3, 504, 504), int(g["input_seed"]))
    ref_max = float(g["depth_stats"][1])
    for dtype, tol in MODES:
        model, cfg, w = _model("vitl", dtype, int(g["weight_seed"]))
        y = model(x.to("cuda", dtype)).float().cpu()
        err = record_err(float((y[:, ::4, ::4].double() - torch.from_numpy(g["depth_strided"]).double()).abs().max()) / ref_max, str(dtype))
        assert err <= tol, f"{dtype}: {err}"
        del model
        torch.cuda.empty_cache()


def test_full_size_batch32_properties():
    """BASELINE configs[2] at full size (ViT-L, batch 32, 504x504) through size-independent properties:
    batch independence (sharding a batch never changes a map: the basis of data parallelism), determinism, finiteness."""
    model, cfg, w = _model("vitl", torch.bfloat16)
    x = torch.randn(32, 3, 504, 504, generator=torch.Generator().manual_seed(7)).to("cuda", torch.bfloat16)
    y = model(x)
    assert tuple(y.shape) == (32, 504, 504) and bool(torch.isfinite(y).all())
    assert torch.equal(model(x), y), "same input twice must be bit-identical"
    for i in (0, 13, 31):
        assert torch.equal(model(x[i:i + 1])[0], y[i]), f"image {i}: batch-of-1 result differs from its row in the batch of 32"
    y_halves = torch.cat((model(x[:16]), model(x[16:])), dim=0)
    assert torch.equal(y_halves, y), "two shards of 16 must reproduce the batch of 32 bit-for-bit"
    assert float(y.float().max()) > 0.1, "degenerate (all-zero) output"


def test_vitl_batch32_every_checked_image_vs_oracle():
    """BASELINE configs[2] at FULL size (ViT-L, 504x504 tensor, batch 32), both arithmetic modes: images 0, 7, 13 and 31 of the batch
    against the CPU oracle (dpt_model.py:61-83 restated), per image, plus each of them against its batch-of-1 result (bitwise)."""
    osd, cfg, w = synthetic_model("vitl", 0)
    x = seeded_input((32, 3, 504, 504), 1)
    idx = [0, 7, 13, 31]
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    ref = _oracle().forward(w, cfg, x[idx])
    # bf16: measured per image over the whole batch 0.9e-2 ... 1.9e-2 (median 1.4e-2, tools/probes/gpu_vitl_batch32_parity.py); the bound is the
    # reference-derived one of tests/helpers.py (the reference's own bf16 path: 2.15e-2 on image 0)
    for dtype, tol in ((torch.float32, REL_TOL_X3), (torch.bfloat16, REL_TOL_BF16)):
        model, _, _ = _model("vitl", dtype)
        xd = x.to("cuda", dtype)
        y = model(xd)
        assert tuple(y.shape) == (32, 504, 504) and y.dtype == dtype
        for k, i in enumerate(idx):
            err = record_err(float((y[i].float().cpu().double() - ref[k].double()).abs().max() / ref[k].double().abs().max()), f"{dtype} image {i}")
            assert err <= tol, f"{dtype} image {i}: {err:.3e}"
            assert torch.equal(model(xd[i:i + 1])[0], y[i]), f"{dtype} image {i}: batch-of-1 result differs from its row in the batch of 32"
        del model, y, xd
        torch.cuda.empty_cache()


def test_vitl_1036_vs_oracle():
    """The other north-star size on the headline model: ViT-L, 1036x1036 (74x74 grid, 5477 tokens), one image, both modes, vs the CPU oracle
    (~30 s of host time on the GPU box)."""
    osd, cfg, w = synthetic_model("vitl", 0)
    x = seeded_input((1, 3, 1036, 1036), 1)
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    ref = _oracle().forward(w, cfg, x)
    for dtype, tol in MODES:  # measured: x3 3.2e-5, bf16 1.5e-2
        model, _, _ = _model("vitl", dtype)
        y = model(x.to("cuda", dtype)).float().cpu()
        assert tuple(y.shape) == (1, 1036, 1036)
        err = rel_err(y, ref)
        assert err <= tol, f"{dtype}: {err:.3e}"
        del model
        torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_every_gemm_tile_variant_is_bitwise_identical(dtype):
    """All main-loop variants (128x128x64, 256x256x64 lockstep, 256x128x32 3-deep ring, 256x256x32 ping-pong, 64x64x64)
    through every A-row generator and epilogue of the model: same K order + fp-contract off => identical bits."""
    model, cfg, w = _model("vits", dtype)
    x = seeded_input((2, 3, 252, 252), 11).to("cuda", dtype)
    y_auto = model(x)
    ref = _oracle().forward(w, cfg, x.float().cpu())
    assert rel_err(y_auto.float().cpu(), ref) <= (REL_TOL_X3 if dtype == torch.float32 else REL_TOL_BF16)
    for tile in (1, 2, 4, 5, 6):
        model.set_gemm_tile(tile)
        assert torch.equal(model(x), y_auto), f"tile variant {tile} changed the result"
    model.set_gemm_tile(0)


def test_vits_1036_matches_oracle():
    """The other BASELINE size: 1036x1036 (grid 74x74, N = 5477 tokens) - long-sequence attention + big decoder maps."""
    model, cfg, w = _model("vits", torch.float32)
    x = seeded_input((1, 3, 1036, 1036), 3)
    assert rel_err(model(x.to("cuda")).cpu(), _oracle().forward(w, cfg, x)) <= REL_TOL_X3


def test_prepare_image_kernel_vs_golden(golden_dir):
    """mdpt_prepare_image (HIP antialiased bilinear + BGR->RGB + normalise) against outputs of the reference's
    prepare_image_bgr (tests/golden/gen_golden.py): same shapes (518->504 snapping, aspect-ratio mode, 1036, upscaling) and values."""
    from tests.helpers import stats
    g = np.load(os.path.join(golden_dir, "prepare_image.npz"))
    model, _, _ = _model("vits", torch.float32)
    names = sorted({k[: -len("_img")] for k in g.files if k.endswith("_img")})
    assert len(names) >= 5
    for name in names:
        side, square = (int(v) for v in g[f"{name}_args"])
        out = model.prepare_image_bgr(g[f"{name}_img"], None if side < 0 else side, bool(square))
        assert out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == tuple(int(v) for v in g[f"{name}_shape"]), name
        err = float((out[:, :, ::7, ::7].cpu().double() - torch.from_numpy(g[f"{name}_out_strided"]).double()).abs().max())
        assert err <= 2e-5, f"{name}: max abs err {err}"   # normalised pixel units (~[-2.1, 2.6]); fp32 summation-order noise only
        np.testing.assert_allclose(stats(out.cpu()), g[f"{name}_stats"], rtol=2e-5, atol=2e-4)


def test_prepare_image_writes_the_model_dtype_and_reuses_its_pinned_staging_buffer():
    """mdpt_prepare_image's output is dtype-tagged - a bf16 / fp16 model gets its image from the kernel itself (the fp32
    result rounded ONCE, no cast kernel), the uint8 host image travels through a reusable pinned buffer, and back-to-back calls with
    different images do not overwrite an image whose copy is still in flight."""
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in ((518, 518), (300, 411), (518, 518))]
    model, _, _ = _model("tiny", torch.float32)
    want = [model.prepare_image_bgr(im, 140) for im in imgs]
    for dtype in (torch.bfloat16, torch.float16):
        m16, _, _ = _model("tiny", dtype)
        outs = [m16.prepare_image_bgr(im, 140) for im in imgs]  # queued back to back: no host synchronisation in between
        torch.cuda.synchronize()
        for o, w32 in zip(outs, want):
            assert o.dtype == dtype and o.shape == w32.shape
            assert torch.equal(o, w32.to(dtype)), "the 16-bit image must be the fp32 image rounded once"
        stage = m16.patch_embed.__dict__["_host_stage"]
        assert len(stage) == 1 and next(iter(stage.values()))["pinned"].is_pinned()
        # a non-contiguous view (cv2 crops) takes the same path
        crop = imgs[0][10:400, 20:300]
        assert torch.equal(m16.prepare_image_bgr(crop, 140), model.prepare_image_bgr(np.ascontiguousarray(crop), 140).to(dtype))


@pytest.mark.parametrize("dtype,tol", MODES_TOY)
def test_depth_anything_v1_family(golden_dir, dtype, tol):
    """§8(f) row 3: Depth-Anything V1 (taps after the last four blocks) through make_dpt_from_state_dict's v1 route."""
    from muggled_dpt_amd import make_depthanythingv1_dpt_from_original_state_dict
    from muggled_dpt_amd.synthetic import STANDARD_CONFIGS, make_synthetic_original_state_dict
    g = np.load(os.path.join(golden_dir, "tiny_v1.npz"))
    osd = make_synthetic_original_state_dict(dict(STANDARD_CONFIGS["tiny"], num_blocks=int(g["num_blocks"])), int(g["weight_seed"]))
    cfg, model = make_depthanythingv1_dpt_from_original_state_dict(osd)
    assert len(cfg) == 9 and "blocks.7.attn.qkv.weight" in model.imgencoder.state_dict()
    model = model.to("cuda", dtype)
    x = torch.from_numpy(g["input"]).to("cuda", dtype)
    y = model(x)
    assert rel_err(y.float().cpu(), torch.from_numpy(g["depth"])) <= tol
    taps = model.debug_taps(2, (56, 84))
    for i in range(4):
        assert rel_err(taps["stages"][i].cpu(), torch.from_numpy(g[f"tap{i}"])) <= tol, f"tap{i}"
    assert rel_err(taps["fused"].cpu(), torch.from_numpy(g["fused"])) <= tol


def test_channels_last_model_and_input():
    """The reference's GPU default: run_image.py:158 moves the model with `.to(device, dtype, memory_format=torch.channels_last)`
    (demo_helpers/misc.py:76-77) and channels_last images reach forward(). Same bits as the contiguous call; parameters whose strides
    are permuted by the format (4-D conv weights) are bound through a contiguous view."""
    for dtype in (torch.bfloat16, torch.float16, torch.float32):
        model, cfg, w = _model("tiny", dtype)
        x = seeded_input((2, 3, 56, 84), 41).to("cuda", dtype)
        y = model(x)
        model_cl = model.to("cuda", dtype, memory_format=torch.channels_last)
        assert model_cl is model and model.patch_embed.proj.weight.is_contiguous(memory_format=torch.channels_last)
        x_cl = x.to(memory_format=torch.channels_last)
        assert not x_cl.is_contiguous() and x_cl.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(model(x_cl), y), f"{dtype}: a channels_last model / image changed the result"
        assert torch.equal(model(x), y)


def test_vit_base_config_matches_oracle():
    """ViT-B sized model (F=768, 12 heads; reference make_depthanythingv2_dpt.py:97-104 standard configs)."""
    model, cfg, w = _model("vitb", torch.float32)
    x = seeded_input((1, 3, 252, 308), 21)
    assert rel_err(model(x.to("cuda")).cpu(), _oracle().forward(w, cfg, x)) <= REL_TOL_X3


def test_metric_head_sigmoid():
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    osd, cfg, w = synthetic_model("tiny", 0)
    osd = dict(osd)
    osd["is_metric"] = torch.tensor(1.0)
    cfg_m, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    assert cfg_m["is_metric"] is True
    x = seeded_input((1, 3, 56, 56), 4)
    ref = _oracle().forward(w, dict(cfg, is_metric=True), x)
    assert rel_err(model.to("cuda")(x.to("cuda")).cpu(), ref) <= REL_TOL_X3


def test_make_dpt_from_state_dict_file_roundtrip(tmp_path):
    from muggled_dpt_amd import make_dpt_from_state_dict
    osd, cfg, w = synthetic_model("tiny", 0)
    path = str(tmp_path / "depth_anything_v2_tiny.pth")
    torch.save(osd, path)
    cfg2, model = make_dpt_from_state_dict(path, enable_cache=True)
    assert cfg2["features_per_token"] == 64 and cfg2["enable_cache"] is True
    x = seeded_input((1, 3, 56, 56), 9)
    assert rel_err(model.to("cuda")(x.to("cuda")).cpu(), _oracle().forward(w, cfg, x)) <= REL_TOL_X3


@pytest.mark.parametrize("dtype,tol", MODES_TOY)
def test_vit_giant_swiglu_ffn(golden_dir, dtype, tol):
    """§8(f) row 3: is_giant checkpoints (mlp.w12 / mlp.w3 -> SwiGLU FFN, hidden width 344 here: exercises the K padding of the
    outer GEMM) vs a fixture generated from the reference."""
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
    g = np.load(os.path.join(golden_dir, "tiny_giant.npz"))
    cfg, model = make_depthanythingv2_dpt_from_original_state_dict(make_synthetic_original_state_dict("tiny_giant", int(g["weight_seed"])))
    assert cfg["is_giant"] is True and "stages.0.blocks.0.mlp.inner_linear_doubled.weight" in model.imgencoder.state_dict()
    model = model.to("cuda", dtype)
    y = model(torch.from_numpy(g["input"]).to("cuda", dtype))
    taps = model.debug_taps(2, (56, 84))
    for i in range(4):
        assert rel_err(taps["stages"][i].cpu(), torch.from_numpy(g[f"tap{i}"])) <= tol, f"tap{i}"
    assert rel_err(taps["fused"].cpu(), torch.from_numpy(g["fused"])) <= tol
    assert rel_err(y.float().cpu(), torch.from_numpy(g["depth"])) <= tol  # measured: bf16 0.8e-2


def test_vit_giant_full_width_runs():
    """The real ViT-G sizes (F=1536, 24 heads, 40 blocks, SwiGLU hidden 4096, C=384) at 504x504, bf16: finite, non-trivial output
    and equal to the batch-1 result (CPU oracle at this size takes minutes, so only properties are checked here)."""
    model, cfg, w = _model("vitg", torch.bfloat16)
    x = seeded_input((2, 3, 504, 504), 2).to("cuda", torch.bfloat16)
    y = model(x)
    assert tuple(y.shape) == (2, 504, 504) and torch.isfinite(y.float()).all() and float(y.float().max()) > 0
    assert torch.equal(y[1:], model(x[1:]))


def test_batch_split_under_hipgraph_capture_and_toggle():
    """mdpt_forward splits batches >= 8 over the caller's stream and an internal side stream (event fork / join). The split must
    (a) not change a single bit, (b) be capturable into a hipGraph (cross-stream events are legal in stream capture)."""
    from muggled_dpt_amd import native
    model, cfg, w = _model("vits", torch.bfloat16)
    x = seeded_input((8, 3, 252, 252), 17).to("cuda", torch.bfloat16)
    y_split = model(x)
    eng = model._get_engine()
    native.check(eng.lib, eng.lib.mdpt_set_batch_split(eng.handle, 0))
    y_plain = model(x)
    native.check(eng.lib, eng.lib.mdpt_set_batch_split(eng.handle, 8))
    assert torch.equal(y_split, y_plain)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        model(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        y_graph = model(x)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y_graph, y_plain)


@pytest.mark.parametrize("dtype,tol", MODES_TOY)
def test_fusion_blocks_are_callable_one_by_one(golden_dir, dtype, tol):
    """experiments/fusion_scaling.py:330-334 drives model.fusion.blocks[i] directly: blocks[3](r4), blocks[i](r_i, previous)."""
    g = np.load(os.path.join(golden_dir, "tiny_full.npz"))
    model, cfg, w = _model("tiny", dtype)
    orc = _oracle()
    reasm = [torch.from_numpy(g[f"reasm{i}"]) for i in range(4)]
    dev = [r.to("cuda", dtype) for r in reasm]
    assert len(model.fusion.blocks) == 4 and model.fusion.blocks[-1] is model.fusion.blocks[3]
    prev_ref, prev = None, None
    for i in (3, 2, 1, 0):
        ref = orc.fusion_block(w, i, reasm[i], prev_ref)
        out = model.fusion.blocks[i](dev[i]) if i == 3 else model.fusion.blocks[i](dev[i], prev)
        assert out.dtype == dtype and tuple(out.shape) == tuple(ref.shape)
        assert rel_err(out.float().cpu(), ref) <= tol, f"block {i}"
        prev_ref, prev = ref, ref.to("cuda", dtype)  # feed the exact previous map so errors do not compound
    assert rel_err(prev_ref, torch.from_numpy(g["fused"])) <= 1e-5
    with pytest.raises(TypeError):
        model.fusion.blocks[3](dev[3], dev[3])
    with pytest.raises(TypeError):
        model.fusion.blocks[1](dev[1])


def test_float16_model_dtype_runs_fp16_operands():
    """run_image.py offers fp16 models (`-f16`, demo_helpers/misc.py:73-77 picks bf16 when supported, else fp16): a float16 model runs
    fp16 MFMA operands (MDPT_PREC_FP16, round 4; tests/test_gpu_precision_modes.py has the full-size cases) with fp16 tensors at the API
    boundary. Tolerance: a quarter of the bf16 toy tolerance (measured ~8x below the bf16 model's error)."""
    model, cfg, w = _model("tiny", torch.float16)
    x = seeded_input((2, 3, 56, 84), 23)
    y = model(x.to("cuda", torch.float16))
    assert y.dtype == torch.float16 and tuple(y.shape) == (2, 56, 84)
    assert rel_err(y.float().cpu(), _oracle().forward(w, cfg, x)) <= REL_TOL_BF16_TOY / 4
    tok, hw = model.patch_embed(x.to("cuda", torch.float16))
    assert tok.dtype == torch.float16 and tuple(hw) == (4, 6)


def test_latency_mode_split_kv_attention_matches_the_oracle():
    """mdpt_set_latency_mode: small launches split the key loop of the attention kernel over the four waves of a workgroup (partial
    softmax states merged through LDS). Same accuracy against the oracle; bits may differ from the batch-invariant default form."""
    for name, shape in (("tiny", (2, 3, 56, 84)), ("vits", (1, 3, 504, 504))):
        model, cfg, w = _model(name, torch.bfloat16)
        x = seeded_input(shape, 21)
        ref = _oracle().forward(w, cfg, x)
        y_default = model(x.to("cuda", torch.bfloat16))
        model.set_latency_mode(True)
        y_fast = model(x.to("cuda", torch.bfloat16))
        assert rel_err(y_fast.float().cpu(), ref) <= (REL_TOL_BF16_TOY if name == "tiny" else REL_TOL_BF16)
        assert rel_err(y_fast.float().cpu(), y_default.float().cpu()) <= (REL_TOL_BF16_TOY if name == "tiny" else REL_TOL_BF16)
        model.set_latency_mode(False)
        assert torch.equal(model(x.to("cuda", torch.bfloat16)), y_default)  # and back: the default form is reproduced exactly


def test_prepare_image_kernel_on_random_sizes_vs_oracle():
    """The HIP antialiased resize against the oracle (= F.interpolate(antialias=True), pinned to the reference) on random image sizes,
    aspect ratios, target sides and both sizing modes: strong down-scales (wide triangle filters), up-scales, 1-pixel-wide borders."""
    model, _, _ = _model("tiny", torch.float32)
    orc = _oracle()
    rng = np.random.default_rng(123)
    for k in range(24):
        h, w = int(rng.integers(9, 900)), int(rng.integers(9, 900))
        side = [None, int(rng.integers(28, 700))][int(rng.integers(0, 2))]
        square = bool(rng.integers(0, 2))
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        got = model.prepare_image_bgr(img, side, square)
        want = orc.prepare_image(img, side, square, default_size_px=model.patch_embed._default_size_px)
        assert got.is_cuda and tuple(got.shape) == tuple(want.shape), (h, w, side, square, tuple(got.shape), tuple(want.shape))
        err = float((got.cpu() - want).abs().max())
        assert err <= 3e-5, f"{h}x{w} side={side} square={square}: max abs err {err:.2e}"


def test_prepare_image_bicubic_vs_golden_and_unsupported_modes_raise(golden_dir):
    """interpolation_mode="bicubic" (patch_embed.py:108,141) runs in the same HIP kernel (cubic filter, a = -0.5, antialiased) and matches
    the reference-generated fixture; every other mode raises like torch's F.interpolate(antialias=True) does - there is no torch fallback."""
    g = np.load(os.path.join(golden_dir, "prepare_image_bicubic.npz"))
    model, _, _ = _model("tiny", torch.float32)
    orc = _oracle()
    for name in ("down", "sq", "up"):
        side, square = (int(v) for v in g[f"{name}_args"])
        out = model.prepare_image_bgr(g[f"{name}_img"], side, bool(square), "bicubic")
        want = torch.from_numpy(g[f"{name}_out"])
        assert out.is_cuda and tuple(out.shape) == tuple(want.shape), name
        assert record_err(float((out.cpu() - want).abs().max()), name) <= 3e-5, name   # normalised pixel units
    rng = np.random.default_rng(5)
    for k in range(8):
        h, w = int(rng.integers(9, 600)), int(rng.integers(9, 600))
        side = int(rng.integers(28, 500))
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        got = model.prepare_image_bgr(img, side, bool(k & 1), "bicubic")
        want = orc.prepare_image(img, side, bool(k & 1), "bicubic", default_size_px=model.patch_embed._default_size_px)
        assert tuple(got.shape) == tuple(want.shape) and float((got.cpu() - want).abs().max()) <= 5e-5, (h, w, side)
    img = g["sq_img"]
    for mode in ("nearest", "area", "nearest-exact"):
        with pytest.raises(ValueError):
            model.prepare_image_bgr(img, 56, True, mode)
    with pytest.raises(TypeError):
        model.prepare_image_bgr(img.astype(np.float32), 56, True)

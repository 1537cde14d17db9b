"""The fp16-operand and mixed-pass arithmetic modes (MDPT_PREC_FP16 / FP16X3 / MIXED, mdpt_set_class_passes) against the CPU oracle.
Run with `pytest -m gpu` on an MI355X.

What the reference does with these dtypes: demo_helpers/misc.py:61-77 hands the model float16 whenever bf16 is not preferred, and a
float16 model there computes with fp16 (11-bit significand) operands. Here a torch.float16 model selects v_mfma_f32_*_f16 (same
rate as bf16, 8x smaller operand rounding); "mixed" adds three-pass (hi + lo) arithmetic for the op classes that dominate the error
(tests/precision_budget/: CPU emulation of the budget; profiles/r04_precision_budget.md: the measured table).

Tolerances: rel = max|y - ref| / max|ref|, north-star bar 1e-3.
  * fp16, ViT-L 504 batch 32: emulated 1.7e-3 ... 2.2e-3 per image  -> asserted at 3e-3 and at <= 1/4 of the bf16 error of the same image
  * mixed: asserted at the north-star bar itself, 1e-3, on every checked image
  * fp16x3: asserted at REL_TOL_X3 (1e-4), like bf16x3
"""
import os

import numpy as np
import pytest
import torch

from tests.helpers import REL_TOL_BF16_TOY, REL_TOL_X3, record_err, ref_lowprec_tol, rel_err, seeded_input, synthetic_model

pytestmark = pytest.mark.gpu

REL_TOL_FP16 = 3e-3        # single-pass fp16 operands, full-size models
REL_TOL_FP16_TOY = 5e-3    # 64-feature toy configurations (bf16: 3e-2; fp16 rounds 8x finer)
REL_TOL_MIXED = 1e-3       # the north-star bar
REL_GATE_MIXED = 8.5e-4    # regression gate UNDER the bar for the float32 mixed model at full size: the max-over-pixels metric moves
                           # by +-10 % under equally valid re-roundings, the shipped table has to leave that much


def _oracle():
    from oracle import dpt_oracle
    return dpt_oracle


def _model(name, dtype, precision=None, seed=0):
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    osd, cfg, w = synthetic_model(name, seed)
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    model = model.to("cuda", dtype)
    if precision:
        model.set_precision(precision)
    return model, cfg, w


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from muggled_dpt_amd import native
    native.load()


def test_float16_model_selects_fp16_operands_and_beats_bf16_by_the_mantissa_ratio(golden_dir):
    """A float16 model runs MDPT_PREC_FP16: every stage boundary of the toy model against the reference fixture, and the depth error is
    several times below the bf16 model's on the same input (3 more significand bits = 8x finer rounding)."""
    g = np.load(os.path.join(golden_dir, "tiny_full.npz"))
    x = torch.from_numpy(g["input"])
    errs = {}
    for dtype in (torch.float16, torch.bfloat16):
        model, cfg, w = _model("tiny", dtype)
        y = model(x.to("cuda", dtype))
        assert y.dtype == dtype and tuple(y.shape) == (2, 56, 56)
        errs[dtype] = rel_err(y.float().cpu(), torch.from_numpy(g["depth"]))
        if dtype == torch.float16:
            from muggled_dpt_amd import native
            assert model._get_engine().precision == native.PREC_FP16
            taps = model.debug_taps(2, (56, 56))
            for i in range(4):
                assert rel_err(taps["stages"][i].cpu(), torch.from_numpy(g[f"tap{i}"])) <= REL_TOL_FP16_TOY, f"tap{i}"
                assert rel_err(taps["reasm"][i].cpu(), torch.from_numpy(g[f"reasm{i}"])) <= REL_TOL_FP16_TOY, f"reasm{i}"
            assert rel_err(taps["fused"].cpu(), torch.from_numpy(g["fused"])) <= REL_TOL_FP16_TOY
    assert errs[torch.float16] <= REL_TOL_FP16_TOY
    assert errs[torch.bfloat16] <= REL_TOL_BF16_TOY
    assert errs[torch.float16] * 3 <= errs[torch.bfloat16], errs


@pytest.mark.parametrize("precision,tol", [("fp16", REL_TOL_FP16_TOY), ("fp16x3", REL_TOL_X3), ("mixed", 1.5e-3), ("bf16x3", REL_TOL_X3)])  # toy model, mixed: emulated 5e-4 ... 7e-4
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_precision_override_is_independent_of_the_boundary_dtype(dtype, precision, tol):
    """set_precision picks the MFMA operand arithmetic; the model dtype only decides what crosses the C ABI (image in, depth out,
    parameters as bound). A 16-bit boundary adds its own rounding of the input image and of the depth map on top of `tol`."""
    model, cfg, w = _model("tiny", dtype, precision)
    x = seeded_input((2, 3, 56, 84), 29)
    xb = x.to(dtype)
    ref = _oracle().forward({k: v.to(dtype).float() for k, v in w.items()}, cfg, xb.float())  # the reference's own view of a cast model
    y = model(xb.cuda())
    assert y.dtype == dtype
    boundary = 0.0 if dtype == torch.float32 else (2.0 ** -11 if dtype == torch.float16 else 2.0 ** -8)
    assert rel_err(y.float().cpu(), ref) <= tol + boundary


def test_vits_504_fp16_and_mixed_vs_golden_fixture(golden_dir):
    """BASELINE configs[1] in the fp16 modes: ViT-S, 518x518 image -> 504x504 tensor, batch 1, against the fixture generated from the
    reference itself (tests/golden/gen_golden.py)."""
    g = np.load(os.path.join(golden_dir, "vits504.npz"))
    x = seeded_input((1, 3, 504, 504), int(g["input_seed"]))
    for precision, tol in (("fp16", REL_TOL_FP16), ("mixed", REL_TOL_MIXED), ("fp16x3", REL_TOL_X3)):
        model, cfg, w = _model("vits", torch.float32, precision, seed=int(g["weight_seed"]))
        y = model(x.cuda()).cpu()
        err = record_err(float((y[:, ::4, ::4].double() - torch.from_numpy(g["depth_strided"]).double()).abs().max()) / float(g["depth_stats"][1]), precision)
        assert err <= tol, f"{precision}: {err:.3e}"
        del model
        torch.cuda.empty_cache()


def test_vitl_batch32_fp16_and_mixed_every_checked_image_vs_oracle():
    """BASELINE configs[2] at FULL size (ViT-L, 504x504 tensor, batch 32): images 0, 7, 13 and 31 against the CPU oracle in the fp16
    and the mixed mode, per image; each of them bitwise equal to its batch-of-1 result; the mixed mode meets the north star's 1e-3."""
    osd, cfg, w = synthetic_model("vitl", 0)
    x = seeded_input((32, 3, 504, 504), 1)
    idx = [0, 7, 13, 31]
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    ref32 = _oracle().forward(w, cfg, x[idx])
    # a float16 MODEL holds fp16-rounded parameters and sees an fp16-rounded image - in the reference too (run_image.py:158 casts the model,
    # patch_embed.py:133 the image): its cases are checked against the oracle on exactly those rounded tensors
    ref16 = _oracle().forward({k: v.half().float() for k, v in w.items()}, cfg, x[idx].half().float())
    # ... and ALSO against the plain fp32 oracle: what a float16 MODEL loses to its fp16-rounded parameters and image comes on
    # top of the operand arithmetic. The yardstick is the reference's own float16 path on this configuration (model and input cast to float16
    # against its fp32 run: tests/golden/reference_lowprec_errors.json, 9.9e-3): at most HALF of that.
    tol16_vs_fp32 = ref_lowprec_tol("vitl504", dtype="fp16", factor=0.5)
    for dtype, precision, tol in ((torch.float16, None, REL_TOL_FP16), (torch.float32, "mixed", REL_GATE_MIXED), (torch.float16, "mixed", REL_TOL_MIXED + 2.0 ** -11)):
        ref = ref32 if dtype == torch.float32 else ref16
        model, _, _ = _model("vitl", dtype, precision)
        xd = x.to("cuda", dtype)
        y = model(xd)
        assert tuple(y.shape) == (32, 504, 504) and y.dtype == dtype
        for k, i in enumerate(idx):
            err = record_err(float((y[i].float().cpu().double() - ref[k].double()).abs().max() / ref[k].double().abs().max()), f"{dtype} {precision} image {i}")
            assert err <= tol, f"{dtype} {precision} image {i}: {err:.3e}"
            if dtype == torch.float16:
                e32 = record_err(float((y[i].float().cpu().double() - ref32[k].double()).abs().max() / ref32[k].double().abs().max()), f"{dtype} {precision} image {i} vs the fp32 oracle")
                assert e32 <= tol16_vs_fp32, f"{dtype} {precision} image {i} vs the fp32 oracle: {e32:.3e} > {tol16_vs_fp32:.3e}"
            assert torch.equal(model(xd[i:i + 1])[0], y[i]), f"{dtype} {precision} image {i}: batch-of-1 result differs from its row in the batch of 32"
        del model, y, xd
        torch.cuda.empty_cache()


def test_vitl_1036_fp16_and_mixed_vs_oracle():
    """The other north-star size in the modes round 4 added: ViT-L, 1036x1036 (74x74 grid, N = 5477 tokens: reductions
    4x longer than at 504), one image, single-pass fp16 and the mixed mode against the CPU fp32 oracle (~30 s of host time on the GPU box)."""
    osd, cfg, w = synthetic_model("vitl", 0)
    x = seeded_input((1, 3, 1036, 1036), 1)
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    ref = _oracle().forward(w, cfg, x)
    model, _, _ = _model("vitl", torch.float32)
    for precision, tol in (("fp16", REL_TOL_FP16), ("mixed", REL_TOL_MIXED)):
        model.set_precision(precision)
        y = model(x.cuda()).cpu()
        assert tuple(y.shape) == (1, 1036, 1036)
        err = record_err(rel_err(y, ref), f"1036 {precision}")
        assert err <= tol, f"{precision} at 1036x1036: {err:.3e}"
    del model
    torch.cuda.empty_cache()


def test_class_passes_all_three_equals_the_x3_mode_bitwise_and_each_class_switches_alone():
    """mdpt_set_class_passes: every class at 3 passes on top of "fp16" IS "fp16x3" (same bits); one class at a time changes the result
    and never makes it worse than single-pass fp16 by more than noise."""
    from muggled_dpt_amd import native
    x = seeded_input((2, 3, 56, 84), 31)
    model, cfg, w = _model("tiny", torch.float32, "fp16x3")
    y_x3 = model(x.cuda())
    ref = _oracle().forward(w, cfg, x)
    model.set_precision("fp16")
    y_1 = model(x.cuda())
    e1 = rel_err(y_1.cpu(), ref)
    model.set_class_passes({c: 3 for c in native.OP_CLASSES})
    assert torch.equal(model(x.cuda()), y_x3)
    for c in native.OP_CLASSES:
        model.set_class_passes({c: 3})
        y_c = model(x.cuda())
        ec = rel_err(y_c.cpu(), ref)
        assert ec <= 1.5 * e1 + 1e-5, f"class {c} at 3 passes: {ec:.3e} vs {e1:.3e} single-pass"
        assert not torch.equal(y_c, y_1), f"class {c}: three passes left every bit unchanged"
    # two passes (activations split, weights one plane): between the single-pass and the three-pass result of the same class, never the
    # bits of either; every class at once stays fp16-class or better
    for c in native.OP_CLASSES:
        if c == "attn":
            with pytest.raises(ValueError):
                model.set_class_passes({c: 2})
            continue
        model.set_class_passes({c: 2})
        y_c = model(x.cuda())
        ec = rel_err(y_c.cpu(), ref)
        assert ec <= 1.5 * e1 + 1e-5, f"class {c} at 2 passes: {ec:.3e} vs {e1:.3e} single-pass"
        if c == "proj":  # its activations come from the attention kernel, whose single-pass form writes no lo plane (a zero plane): nothing to add
            assert torch.equal(y_c, y_1)
            continue
        assert not torch.equal(y_c, y_1), f"class {c}: two passes left every bit unchanged"
    model.set_class_passes({c: 2 for c in native.OP_CLASSES if c != "attn"})
    e2 = rel_err(model(x.cuda()).cpu(), ref)
    assert e2 <= e1 + 1e-5, f"every class at 2 passes: {e2:.3e} vs {e1:.3e} single-pass"
    model.set_class_passes(None)
    assert torch.equal(model(x.cuda()), y_1)
    with pytest.raises(ValueError):
        model.set_class_passes({"nope": 3})
    with pytest.raises(ValueError):
        model.set_precision("fp8")


@pytest.mark.parametrize("name,hw,batch", [("tiny", (56, 84), 2), ("vits", (140, 112), 1), ("vitl", (112, 140), 3), ("vitl", (504, 504), 1)])
def test_two_pass_head_tail_is_one_kernel_and_loses_only_the_weight_rounding(name, hw, batch):
    """MDPT_CLASS_HEAD_TAIL at two passes = head_tail2_kernel (head.hip): the x(P/8) upsample interpolated from hi + lo planes of conv 1's output,
    hi + lo halo planes against ONE rounded plane of the 3x3 conv's weights (head_model.py:78-85). The head stage alone on an fp32 fused map,
    everything else at three passes: with fp16-REPRESENTABLE conv weights the two-pass tail is fp32-class like the three-pass kernels; with the
    original weights what is left is their rounding - below the single-pass tail's error, above the three-pass one's."""
    osd, cfg, w = synthetic_model(name, 0)
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    gh, gw = hw[0] // 14, hw[1] // 14
    fused = seeded_input((batch, cfg["fusion_channels"], 8 * gh, 8 * gw), 17)
    errs = {}
    for rounded in (True, False):
        osd_r = dict(osd)
        key = next(k for k in osd_r if k.endswith("output_conv2.0.weight"))
        if rounded:
            osd_r[key] = osd_r[key].half().float()
        _, model = make_depthanythingv2_dpt_from_original_state_dict(osd_r)
        model = model.to("cuda", torch.float32)
        model.set_precision("fp16x3")
        wr = dict(w)
        if rounded:
            wr["head.proj_1ch.0.weight"] = w["head.proj_1ch.0.weight"].half().float()
        ref = _oracle().head(wr, cfg, fused)
        for passes in (3, 2, 1):
            model.set_class_passes({"head_tail": passes})
            y = model.head(fused.cuda())
            assert tuple(y.shape) == tuple(ref.shape)
            errs[(rounded, passes)] = rel_err(y.cpu(), ref)
        del model
    assert errs[(True, 3)] <= REL_TOL_X3 and errs[(True, 2)] <= REL_TOL_X3, errs   # representable weights: nothing is lost
    assert errs[(False, 3)] <= REL_TOL_X3, errs
    assert errs[(False, 2)] <= 0.9 * errs[(False, 1)] and errs[(False, 1)] <= REL_TOL_FP16_TOY, errs
    torch.cuda.empty_cache()


def _resid_after(model, x, block, step):
    """fp32 residual stream [B * npad * F] after sub-step `step` of encoder block `block` (mdpt_debug_set_stop / mdpt_debug_read test hooks)."""
    from muggled_dpt_amd import native
    eng = model._get_engine()
    b, _, h, w = x.shape
    native.check(eng.lib, eng.lib.mdpt_debug_set_stop(eng.handle, block, step))
    try:
        model(x)
        npad = -(-((h // eng.P) * (w // eng.P) + 1) // 8) * 8
        out = torch.empty(b * npad * eng.F, device="cuda", dtype=torch.float32)
        ws_ptr, ws_bytes = eng.workspace(b, (h, w))
        native.check(eng.lib, eng.lib.mdpt_debug_read(eng.handle, b"resid", out.data_ptr(), out.numel(), ws_ptr, ws_bytes, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
    finally:
        native.check(eng.lib, eng.lib.mdpt_debug_set_stop(eng.handle, -1, -1))
    return out.cpu()


def test_small_layer_scales_do_not_push_the_fp16_planes_into_the_subnormal_range():
    """real checkpoints carry layer-scale gammas of 1e-2 ... 1e-5; folded into proj / fc2 at pack time, gamma * W (ViT-S: ~4e-5) sits
    below fp16's normal range (6.1e-5), where the hi plane keeps ~10 bits and the lo plane - fp(v - hi), 2^-12 of the entry - none: fp16x3 and
    the compensation's weight residue would silently degrade to single-pass (expected ~7e-4 of an update at gamma x 1e-3, ~1e-2 at 1e-4). The fp16
    build packs such a matrix times a power of two and undoes it exactly in the GEMM (GemmParams::wscale: accumulators start at resid * s, the
    epilogue multiplies by 1 / s). Checked on a model whose residual stream is as small as its layer-scaled updates (image, patch bias, cls, pos and
    gammas x 1e-3 - with a stream of O(1) the fp32 stream's own resolution, 2e-3 of such an update, hides everything: tools/probes/gpu_wscale_check.py):
    the depth map against the oracle per mode, and the UPDATE block 1's proj / fc2 add to the stream, fp16x3 against bf16x3 (fp32's exponent range)."""
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components, get_model_config_from_state_dict
    f = 1e-3
    osd, _, _ = synthetic_model("vits", 0)
    small = ("pretrained.cls_token", "pretrained.pos_embed", "pretrained.patch_embed.proj.bias")
    osd = {k: (v * f if (".ls1.gamma" in k or ".ls2.gamma" in k or k in small) else v.clone()) for k, v in osd.items()}
    cfg = get_model_config_from_state_dict(osd)
    w = flatten_components(convert_state_dict_keys(cfg, osd))
    x = seeded_input((2, 3, 140, 112), 23) * f
    ref = _oracle().forward(w, cfg, x)
    upd, errs = {}, {}
    for prec in ("bf16x3", "fp16x3", "fp16", "mixed"):
        _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
        model = model.to("cuda", torch.float32)
        model.set_precision(prec)
        errs[prec] = record_err(rel_err(model(x.cuda()).cpu(), ref), f"depth, {prec}")
        if prec.endswith("x3"):
            for name, before, after in (("proj", 2, 3), ("fc2", 5, 6)):
                upd[(prec, name)] = (_resid_after(model, x.cuda(), 1, after).double() - _resid_after(model, x.cuda(), 1, before).double())
        del model
    assert errs["bf16x3"] <= REL_TOL_X3 and errs["fp16x3"] <= REL_TOL_X3, errs
    assert errs["fp16"] <= REL_TOL_FP16_TOY and errs["mixed"] <= 1.5e-3, errs
    for name in ("proj", "fc2"):
        r, got = upd[("bf16x3", name)], upd[("fp16x3", name)]
        assert float(r.abs().max()) > 0
        err = record_err(float((got - r).abs().max() / r.abs().max()), f"{name} update, fp16x3 vs bf16x3")
        assert err <= 1e-4, f"{name}: the fp16x3 update differs from the bf16x3 one by {err:.3e} of its size"


def test_nonfinite_image_gives_a_nan_depth_map_like_the_reference_in_every_mode():
    """the saturating fp32 -> fp16 operand conversion (v_med3_f32) returns a finite value for NaN and the ReLUs are v_max, so
    the kernels alone would answer a NaN / inf pixel with a finite, meaningless map. The reference (dpt_model.py:61-83, run here once through the
    imported package and pinned by the oracle: tests/test_oracle_golden.py::test_oracle_nonfinite_image_gives_nan_map) returns an all-NaN map for
    that image and leaves the others alone. mdpt_forward does the same in every mode (mdpt_set_nonfinite_propagation, default on): flags from the
    im2col kernel, one small launch behind the head."""
    x = seeded_input((3, 3, 56, 56), 3)
    clean = x.clone()
    x[1, 1, 20, 20] = float("nan")
    x[2, 0, 3, 3] = float("inf")
    for dtype, prec in ((torch.bfloat16, None), (torch.float16, None), (torch.float32, None), (torch.float32, "mixed")):
        m, _, _ = _model("tiny", dtype)
        m.set_precision(prec)
        y = m(x.to("cuda", dtype)).float()
        yc = m(clean.to("cuda", dtype)).float()
        assert bool(torch.isnan(y[1]).all()) and bool(torch.isnan(y[2]).all()), f"{dtype} {prec}: a NaN / inf image must give an all-NaN map"
        assert bool(torch.isfinite(yc).all())
        assert torch.equal(y[0], yc[0]), f"{dtype} {prec}: the clean image of the batch must keep its bits"
        # the flags are cleared by every forward: the same workspace, clean input again
        assert torch.equal(m(clean.to("cuda", dtype)).float(), yc)
    # switch off: the kernels' own answer (fp16 operands: finite) - the behaviour of rounds 4-5, still reachable and still pinned
    m_h, _, _ = _model("tiny", torch.float16)
    m_h.set_nonfinite_propagation(False)
    y = m_h(x.to("cuda", torch.float16)).float()
    assert bool(torch.isfinite(y).all()), "fp16 operand modes without the propagation: a NaN pixel becomes a saturated operand, never a NaN depth"
    m_h.set_nonfinite_propagation(True)
    assert bool(torch.isnan(m_h(x.to("cuda", torch.float16)).float()[1]).all())
    # stage level (not covered by the switch): the bf16 build carries the NaN into the patch tokens, the fp16 build does not
    m_bf, _, _ = _model("tiny", torch.bfloat16)
    tok_bf, _ = m_bf.patch_embed(x.to("cuda", torch.bfloat16))
    tok_h, _ = m_h.patch_embed(x.to("cuda", torch.float16))
    assert bool(torch.isnan(tok_bf.float()).any()) and bool(torch.isfinite(tok_h.float()).all())


def test_nonfinite_propagation_under_the_batch_split_and_for_the_other_families():
    """The two-stream batch split (batch >= 8) runs two plans with a flag array each; SwinV2 and BEiT reach the im2col kernel through their own
    stage functions; the output dtype of the NaN map follows the model dtype."""
    x = seeded_input((9, 3, 56, 56), 4)
    clean = x.clone()
    x[2, 0, 0, 0] = float("nan")    # first half (images 0-3)
    x[8, 2, 55, 55] = float("-inf")  # second half (images 4-8), last pixel of the last image
    for dtype in (torch.float16, torch.bfloat16):
        m, _, _ = _model("tiny", dtype)
        y, yc = m(x.to("cuda", dtype)), m(clean.to("cuda", dtype))
        assert y.dtype == dtype
        bad = torch.isnan(y.float()).flatten(1).all(1).cpu().tolist()
        assert bad == [i in (2, 8) for i in range(9)], bad
        keep = [i for i in range(9) if i not in (2, 8)]
        assert torch.equal(y[keep], yc[keep]) and bool(torch.isfinite(yc.float()).all())
    from tests import test_gpu_beit, test_gpu_swinv2
    for build, name, hw in ((test_gpu_beit._build, "beit_tiny", (96, 160)), (test_gpu_swinv2._build, "swin2_tiny", (192, 320))):
        m, _, _ = build(name, 1, torch.float16)
        x = seeded_input((2, 3, *hw), 6)
        clean = x.clone()
        x[1, 1, 7, 9] = float("nan")
        y, yc = m(x.to("cuda", torch.float16)).float(), m(clean.to("cuda", torch.float16)).float()
        assert bool(torch.isnan(y[1]).all()) and torch.equal(y[0], yc[0]) and bool(torch.isfinite(yc).all()), name


@pytest.mark.parametrize("seed,gammas", [(0, None), (1, (1e-2, 3.0))])
def test_vitl_with_realistic_weight_statistics_every_mode_within_its_tolerance(seed, gammas):
    """VERDICT r05 "missing" 2: no real checkpoint can be loaded offline, and the stand-in tests perturb one statistic at a time on ViT-S. Here the
    full-width ViT-L gets all of them at once (muggled_dpt_amd/synthetic.py: realistic_statistics - log-uniform layer scales down to 1e-5 [or 1e-2 ... 3:
    blocks that write MORE into the stream than the plain synthetic ones], log-normal LayerNorm weights with x8 / x0.05 channels, heavy-tailed Linears,
    two massive-activation channels, images with a DC offset and different contrasts) and every arithmetic mode must hold the tolerance it claims on
    plain synthetic weights. Measured (tools/probes/gpu_realistic_stats_check.py, profiles/r06_realistic_stats.txt): mixed 2.9e-4 ... 4.3e-4 here against
    6.1e-4 ... 6.9e-4 on the plain weights - the plain U(0.5, 1) layer scales are the harder case. Precondition, checked: both depth maps are alive
    (mean >= 5 % of max): the metric is relative to an image's own maximum, and a map the final ReLU clips almost everywhere (one of the probe's
    seeds: mean 0.03 x max) reads 1.04e-3 in the mixed mode with a SMALLER absolute error than any live map."""
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components, get_model_config_from_state_dict
    from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict, realistic_statistics
    from tests.helpers import REL_TOL_BF16
    osd = make_synthetic_original_state_dict("vitl", seed)
    osd = realistic_statistics(osd, seed) if gammas is None else realistic_statistics(osd, seed, *gammas)
    cfg = get_model_config_from_state_dict(osd)
    w = flatten_components(convert_state_dict_keys(cfg, osd))
    x = seeded_input((2, 3, 504, 504), 100 + seed)
    x[0] = x[0] * 0.4 + 1.2
    x[1] = x[1] * 1.2 - 0.8
    ref = _oracle().forward(w, cfg, x)
    for i in range(2):
        assert float(ref[i].mean()) >= 0.05 * float(ref[i].max()) > 0, "precondition: a live depth map"
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    model = model.to("cuda", torch.float32)
    for prec, tol in (("bf16x3", REL_TOL_X3), ("fp16x3", REL_TOL_X3), ("mixed", REL_TOL_MIXED), ("fp16", REL_TOL_FP16), ("bf16", REL_TOL_BF16)):
        model.set_precision(prec)
        y = model(x.cuda()).float().cpu()
        worst = max(rel_err(y[i], ref[i]) for i in range(2))  # per image, against that image's own maximum
        record_err(worst, f"realistic statistics seed {seed}, {prec}")
        assert worst <= tol, f"{prec}: {worst:.3e} > {tol:.3e}"


def test_raw_c_abi_class_passes_keep_bound_weights_and_reject_bad_arguments():
    import ctypes
    from muggled_dpt_amd import native
    lib = native.load()
    c = native.MdptConfig()
    c.features_per_token, c.num_heads, c.num_blocks = 64, 1, 4
    for i, v in enumerate((16, 32, 64, 64)):
        c.reassembly_features[i] = v
    c.base_patch_grid_h = c.base_patch_grid_w = 4
    c.fusion_channels, c.patch_size_px, c.precision, c.family = 32, 14, native.PREC_MIXED, native.FAMILY_DAV2
    h = ctypes.c_void_p()
    native.check(lib, lib.mdpt_create(ctypes.byref(c), ctypes.byref(h)))
    # (a 64-feature toy cannot run the fp8 forms of round 6's table - contraction lengths that are not multiples of 128 - and gets round 5's
    #  16-bit-plane table; tests/test_gpu_f8_cross.py has the eligible configuration)
    want = (ctypes.c_int32 * len(native.OP_CLASSES))()
    lib.mdpt_default_mixed_passes_r05(native.FAMILY_DAV2, want)
    new = (ctypes.c_int32 * len(native.OP_CLASSES))()
    lib.mdpt_default_mixed_passes(new)
    assert {native.OP_CLASSES[i]: new[i] for i in range(len(native.OP_CLASSES)) if new[i] != want[i]} == {"reasm": 5, "fusion": 5, "fusion_proj": 5, "head": 5}
    got = ctypes.c_int32()
    for i in range(len(native.OP_CLASSES)):
        native.check(lib, lib.mdpt_get_class_passes(h, i, ctypes.byref(got)))
        assert got.value == want[i] and got.value in (1, 2, 3)
    n0, b0 = lib.mdpt_num_weights(h), ctypes.c_size_t()
    native.check(lib, lib.mdpt_packed_bytes(h, ctypes.byref(b0)))
    t = torch.zeros(64, device="cuda")
    shape = (ctypes.c_int64 * 1)(64)
    native.check(lib, lib.mdpt_bind_weight(h, b"patch_embed.proj.bias", t.data_ptr(), native.DTYPE_F32, 1, shape))
    native.check(lib, lib.mdpt_set_class_passes(h, native.OP_CLASSES.index("fc1"), 3))
    b1 = ctypes.c_size_t()
    native.check(lib, lib.mdpt_packed_bytes(h, ctypes.byref(b1)))
    assert lib.mdpt_num_weights(h) == n0 and b1.value == b0.value  # fc1 already carried its weight-residue (lo) planes for the compensation
    native.check(lib, lib.mdpt_set_weight_rounding_compensation(h, 0))
    b2 = ctypes.c_size_t()
    native.check(lib, lib.mdpt_packed_bytes(h, ctypes.byref(b2)))
    assert b2.value < b1.value  # qkv / proj / fc2 dropped theirs, fc1 keeps its lo planes as a 3-pass class
    native.check(lib, lib.mdpt_set_class_passes(h, native.OP_CLASSES.index("fc2"), 3))
    b3 = ctypes.c_size_t()
    native.check(lib, lib.mdpt_packed_bytes(h, ctypes.byref(b3)))
    assert b3.value > b2.value
    assert lib.mdpt_set_class_passes(h, 99, 3) == -1  # MDPT_E_INVALID
    assert lib.mdpt_set_class_passes(h, 0, 4) != 0 and lib.mdpt_set_class_passes(h, 0, 0) != 0
    assert lib.mdpt_set_class_passes(h, native.OP_CLASSES.index("attn"), 2) != 0  # both attention operands are activations: 1 or 3
    # two passes = activations split, ONE weight plane: a class moved from 3 to 2 gives its weights' lo planes back
    native.check(lib, lib.mdpt_set_class_passes(h, native.OP_CLASSES.index("reasm"), 3))
    b4 = ctypes.c_size_t()
    native.check(lib, lib.mdpt_packed_bytes(h, ctypes.byref(b4)))
    native.check(lib, lib.mdpt_set_class_passes(h, native.OP_CLASSES.index("reasm"), 2))
    b5 = ctypes.c_size_t()
    native.check(lib, lib.mdpt_packed_bytes(h, ctypes.byref(b5)))
    assert b5.value < b4.value
    lib.mdpt_destroy(h)


def test_fp16_saturates_instead_of_overflowing():
    """fp32 -> fp16 operand converts clamp at +-65504: an activation far outside the fp16 range gives a finite (saturated) result, not
    inf / NaN. Driven through the raw GEMM hook with an A operand of 60000 and weights that sum it past the range in the bf16+GELU path's
    fp16 output plane."""
    from muggled_dpt_amd import native
    lib = native.load()
    M, N, K = 256, 256, 128
    a = torch.full((M, K), 60000.0, device="cuda", dtype=torch.float16)
    w = torch.full((N, K), 1.0, device="cuda", dtype=torch.float16)
    out16 = torch.zeros((M, N), device="cuda", dtype=torch.float16)
    stream = torch.cuda.current_stream().cuda_stream
    native.check(lib, lib.mdpt_debug_set_operand_format(1))
    try:
        native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), None, out16.data_ptr(), M, N, K, 5, 1, stream, None))
        torch.cuda.synchronize()
    finally:
        native.check(lib, lib.mdpt_debug_set_operand_format(0))
    assert bool(torch.isfinite(out16).all()) and float(out16.min()) == 65504.0


@pytest.mark.parametrize("tile", [5, 2, 1, 6, 4])
def test_fp16_gemm_variants_on_ragged_shapes(tile):
    """The bare GEMM kernels in their fp16 build (mdpt_launch_gemm_f16) for every main-loop variant: fp32 strip output and fp16 direct
    output against an fp32 matmul of the same fp16-rounded operands (products exact in fp32: only the accumulation order differs)."""
    from muggled_dpt_amd import native
    lib = native.load()
    rng = np.random.default_rng(tile)
    stream = torch.cuda.current_stream().cuda_stream
    native.check(lib, lib.mdpt_debug_set_operand_format(1))
    try:
        for (M, N, K) in [(1, 256, 128), (257, 1024, 256), (777, 1024, 1024), (1304, 3072, 1024), (1304, 1024, 4096), (4099, 512, 640)]:
            a = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).cuda().to(torch.float16)
            w = torch.from_numpy(rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).cuda().to(torch.float16)
            ref = a.float() @ w.float().t()
            scale = float(ref.abs().max())
            out32 = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.float32)
            native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), out32.data_ptr(), None, M, N, K, tile, 1, stream, None))
            out16 = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.float16)
            native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), None, out16.data_ptr(), M, N, K, tile | (2 << 8), 1, stream, None))
            torch.cuda.synchronize()
            assert float((out32[:M] - ref).abs().max()) <= 2e-5 * scale, f"tile {tile} M={M} N={N} K={K} (fp32 out)"
            gref = torch.nn.functional.gelu(ref)
            assert float((out16[:M].float() - gref).abs().max()) <= 8e-4 * float(gref.abs().max()), f"tile {tile} M={M} N={N} K={K} (fp16 + GELU out)"
            assert torch.all(out32[M:] == 7.0) and torch.all(out16[M:] == 7.0)
    finally:
        native.check(lib, lib.mdpt_debug_set_operand_format(0))


def test_other_families_in_the_fp16_modes(golden_dir):
    """BEiT and SwinV2 toy fixtures (generated from the reference) in fp16 / mixed / fp16x3: the relative-position-bias and window
    attention forms, the readout projection and the patch merge all exist in the fp16 build."""
    from muggled_dpt_amd import make_beit_dpt_from_midas_v31_state_dict, make_swinv2_dpt_from_midas_v31_state_dict
    from muggled_dpt_amd.synthetic import make_synthetic_beit_state_dict, make_synthetic_swinv2_state_dict
    gb = np.load(os.path.join(golden_dir, "beit_tiny.npz"))
    gs = np.load(os.path.join(golden_dir, "swin2_tiny.npz"))
    cases = [(make_beit_dpt_from_midas_v31_state_dict, make_synthetic_beit_state_dict("beit_tiny", int(gb["weight_seed"])), gb, "wide_input", "wide_depth")]
    cases.append((make_swinv2_dpt_from_midas_v31_state_dict, make_synthetic_swinv2_state_dict("swin2_tiny", int(gs["weight_seed"])), gs, "wide_input",
                  "wide_depth"))
    for make, osd, g, kin, kout in cases:
        x = torch.from_numpy(g[kin])
        ref = torch.from_numpy(g[kout])
        errs = {}
        for precision, tol in (("bf16", 6e-2), ("fp16", 1e-2), ("mixed", 5e-3), ("fp16x3", REL_TOL_X3)):
            _, model = make(osd)
            model = model.to("cuda", torch.float32)
            model.set_precision(precision)
            errs[precision] = rel_err(model(x.cuda()).cpu(), ref)
            assert errs[precision] <= tol, f"{make.__name__} {precision}: {errs[precision]:.3e}"
        assert errs["fp16"] * 3 <= errs["bf16"], errs


@pytest.mark.parametrize("precision", ["fp16", "mixed"])
def test_every_gemm_tile_variant_is_bitwise_identical_in_the_fp16_modes(precision):
    """All main-loop variants through every epilogue of the model in the fp16 build, INCLUDING the per-image bias tables of the token-mean
    compensation: the 8-phase kernel adds them in its direct epilogues (two bias vectors per tile, per-row select), the lockstep kernels in
    their LDS-strip epilogues (table row per output row) - same single add, same bits. 252x252: 325 tokens -> 328 rows per image (>= 256)."""
    model, cfg, w = _model("vits", torch.float32, precision)
    model.set_weight_rounding_compensation(True)  # (the default of "mixed"; opt-in for single-pass "fp16")
    x = seeded_input((3, 3, 252, 252), 11).cuda()
    y_auto = model(x)
    ref = _oracle().forward(w, cfg, x.cpu())
    assert rel_err(y_auto.cpu(), ref) <= (REL_TOL_FP16 if precision == "fp16" else REL_TOL_MIXED)
    for tile in (1, 2, 4, 5, 6):
        model.set_gemm_tile(tile)
        assert torch.equal(model(x), y_auto), f"tile variant {tile} changed the result"
    model.set_gemm_tile(0)
    for i in range(3):  # batch invariance with the per-image tables in play
        assert torch.equal(model(x[i:i + 1])[0], y_auto[i])


def test_token_mean_compensation_of_the_weight_rounding_reduces_the_error():
    """mdpt_set_weight_rounding_compensation: on vs off on ViT-S 504x504 in the fp16 mode - the compensated run is closer to the fp32 oracle
    (emulated on ViT-L: the encoder's share of the error halves); toy sizes (24 rows per image: strip epilogues only) too. Defaults: on in
    "mixed", off in single-pass "fp16" (None restores them)."""
    for name, shape in (("vits", (2, 3, 504, 504)), ("tiny", (2, 3, 56, 84))):
        model, cfg, w = _model(name, torch.float32, "fp16")
        x = seeded_input(shape, 37)
        ref = _oracle().forward(w, cfg, x)
        y_default = model(x.cuda()).cpu()
        model.set_weight_rounding_compensation(True)
        y_on = model(x.cuda()).cpu()
        model.set_weight_rounding_compensation(False)
        y_off = model(x.cuda()).cpu()
        assert torch.equal(y_off, y_default)  # single-pass fp16: off unless asked for
        model.set_weight_rounding_compensation(None)
        assert torch.equal(model(x.cuda()).cpu(), y_default)
        model.set_precision("mixed")
        y_mixed = model(x.cuda()).cpu()
        model.set_weight_rounding_compensation(True)
        assert torch.equal(model(x.cuda()).cpu(), y_mixed)  # mixed: on by default
        model.set_weight_rounding_compensation(False)
        assert not torch.equal(model(x.cuda()).cpu(), y_mixed)
        model.set_weight_rounding_compensation(None)
        e_on, e_off = rel_err(y_on, ref), rel_err(y_off, ref)
        rms = lambda y: float((y.double() - ref.double()).pow(2).mean().sqrt())  # noqa: E731
        assert not torch.equal(y_on, y_off)
        assert rms(y_on) <= rms(y_off) * 1.02, f"{name}: rms {rms(y_on):.3e} (on) vs {rms(y_off):.3e} (off); max {e_on:.3e} vs {e_off:.3e}"
    bf = _model("tiny", torch.bfloat16)[0]
    bf.set_weight_rounding_compensation(True)
    from muggled_dpt_amd import native
    with pytest.raises(native.MdptError):
        bf(seeded_input((1, 3, 56, 56), 1).to("cuda", torch.bfloat16))


def test_beit_large_and_swin_large_fixtures_in_the_fp16_and_mixed_modes(golden_dir):
    """BASELINE configs[4] models (BEiT-L-384, SwinV2-L-384; fixtures generated from the reference) in the fp16 operand modes. The yardstick
    for single-pass fp16 is the reference's own float16 CPU path on the same fixture (tests/golden/reference_lowprec_errors.json: 1.4e-2 /
    9.2e-3 - it rounds the residual stream too); mixed is held to the north-star bar for both (token-mean compensation on; SwinV2's encoder
    takes it since round 4: 9.7e-4 -> 7.3e-4 against the oracle at batch 16). At this size the compensation has to pay for itself in the mixed
    mode: the rms error over the map with it is below the one without (measured: SwinV2-L 2.6e-4 -> 1.5e-4, BEiT-L 2.3e-4 -> 1.6e-4 of the
    map's range; on the 6 ... 384-token toy stages it is within noise either way, see the toy test below). In single-pass fp16 the encoder taps
    improve by the same 25 % but the map does not (rms +37 % / +11 % here, -15 % on ViT-L: profiles/r04_wrc_by_family.txt) - it is off by
    default there."""
    from muggled_dpt_amd import make_beit_dpt_from_midas_v31_state_dict, make_swinv2_dpt_from_midas_v31_state_dict
    from muggled_dpt_amd.synthetic import make_synthetic_beit_state_dict, make_synthetic_swinv2_state_dict
    from tests.helpers import ref_lowprec_tol
    for fixture, make, synth, name, mixed_tol in (("beit_large_384", make_beit_dpt_from_midas_v31_state_dict, make_synthetic_beit_state_dict, "beit_large_384", 9e-4),
                                                  ("swin2_large_384", make_swinv2_dpt_from_midas_v31_state_dict, make_synthetic_swinv2_state_dict, "swin2_large_384", 9e-4)):  # (9e-4: 10 % under the bar)
        g = np.load(os.path.join(golden_dir, fixture + ".npz"))
        osd = synth(name, int(g["weight_seed"]))
        x = seeded_input((1, 3, 384, 384), int(g["input_seed"]))
        ref = torch.from_numpy(g["depth_strided"]).double()
        _, model = make(osd)
        model = model.to("cuda", torch.float32)
        for precision, tol in (("fp16", min(ref_lowprec_tol(fixture, dtype="fp16", factor=1.0), 4e-3)), ("mixed", mixed_tol)):
            model.set_precision(precision)
            y = model(x.cuda()).cpu()
            err = record_err(float((y[:, ::4, ::4].double() - ref).abs().max() / ref.abs().max()), f"{fixture} {precision}")
            assert err <= tol, f"{fixture} {precision}: {err:.3e} > {tol:.3e}"
            if precision == "mixed":  # the encoder's single-pass Linears are what is left of the error: the compensation (default on) has to win
                rms = lambda t: float((t[:, ::4, ::4].double() - ref).pow(2).mean().sqrt())  # noqa: E731
                model.set_weight_rounding_compensation(False)
                r_on, r_off = rms(y), rms(model(x.cuda()).cpu())
                model.set_weight_rounding_compensation(None)
                record_err(r_on / r_off, f"{fixture} {precision}: rms with / without the token-mean compensation")
                assert r_on <= r_off, f"{fixture} {precision}: rms {r_on:.3e} with the compensation, {r_off:.3e} without"
        del model
        torch.cuda.empty_cache()


def test_mixed_mode_meets_the_tolerance_under_both_roundings_of_the_fp16_weight_scale(golden_dir):
    """VERDICT r05 (weak item 1 / next item 8): the fp16 build multiplies a layer-scale-folded matrix by a power of two before the hi / lo split
    (GemmParams::wscale). Which matrices get one is a choice between two equally valid roundings of the same weights - the shipped rule (only below
    2^-5) and "scale every folded matrix" (mdpt_debug_set_wscale_policy) - and the max-error metric moves by +-10-20 % between them (profiles/r05_wscale_ab.txt:
    BEiT-L 8.4e-4 <-> 1.007e-3 in round 5). The mixed mode's 1e-3 has to hold under BOTH: ViT-L image 31 (the worst of the checked batch), the
    BEiT-L and SwinV2-L reference fixtures."""
    from muggled_dpt_amd import (make_beit_dpt_from_midas_v31_state_dict, make_depthanythingv2_dpt_from_original_state_dict,
                                 make_swinv2_dpt_from_midas_v31_state_dict)
    from muggled_dpt_amd.synthetic import make_synthetic_beit_state_dict, make_synthetic_swinv2_state_dict
    errs = {}
    # ViT-L 504x504, image 31 of the seeded batch (bits do not depend on the batch: run alone)
    osd, cfg, w = synthetic_model("vitl", 0)
    x31 = seeded_input((32, 3, 504, 504), 1)[31:32]
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    ref31 = _oracle().forward(w, cfg, x31)
    cases = [("vitl image 31", make_depthanythingv2_dpt_from_original_state_dict, osd, x31, lambda y: rel_err(y, ref31))]
    for fixture, make, synth in (("beit_large_384", make_beit_dpt_from_midas_v31_state_dict, make_synthetic_beit_state_dict),
                                 ("swin2_large_384", make_swinv2_dpt_from_midas_v31_state_dict, make_synthetic_swinv2_state_dict)):
        g = np.load(os.path.join(golden_dir, fixture + ".npz"))
        ref = torch.from_numpy(g["depth_strided"]).double()
        cases.append((fixture, make, synth(fixture, int(g["weight_seed"])), seeded_input((1, 3, 384, 384), int(g["input_seed"])),
                      lambda y, ref=ref: float((y[:, ::4, ::4].double() - ref).abs().max() / ref.abs().max())))
    for name, make, sd, x, err_of in cases:
        _, model = make(sd)
        model = model.to("cuda", torch.float32)
        model.set_precision("mixed")
        for scale_all in (False, True):
            model._debug_set_wscale_policy(scale_all)
            errs[(name, scale_all)] = record_err(err_of(model(x.cuda()).cpu()), f"{name}, weight scale on {'every folded matrix' if scale_all else 'matrices below 2^-5'}")
        del model
        torch.cuda.empty_cache()
    print({f"{k[0]} / {'all' if k[1] else 'rule'}": f"{v:.2e}" for k, v in errs.items()})
    for k, v in errs.items():  # (the tolerance is REL_TOL_MIXED = 1e-3; the table is chosen to leave 10 % under either rounding - round 6 reads 6.0e-4 ... 8.0e-4)
        assert v <= 0.9 * REL_TOL_MIXED, f"{k}: {v:.3e}"


def test_massive_activation_channels_in_the_residual_stream():
    """Real DINOv2 checkpoints carry "massive activation" channels - a few features of the residual stream sit tens of sigma away from the rest in
    every token (the synthetic weights have none: VERDICT r03, missing item 4). Emulated here by adding +100 / -60 to two channels of the position
    embedding (the same offset in every token): LayerNorm then sees rows dominated by two features, its output has a large shared component - the
    regime the token-mean compensation is built for - and every arithmetic mode has to stay inside its tolerance against the fp32 oracle."""
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components, get_model_config_from_state_dict
    from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
    from tests.helpers import emulated_tol
    osd = make_synthetic_original_state_dict("vits", 0)
    key = [k for k in osd if k.endswith("pos_embed")][0]
    osd[key] = osd[key].clone()
    osd[key][:, 1:, 7] += 100.0   # patch positions (index 0 is the cls entry)
    osd[key][:, 1:, 201] -= 60.0
    cfg = get_model_config_from_state_dict(osd)
    w = flatten_components(convert_state_dict_keys(cfg, osd))
    x = seeded_input((2, 3, 252, 252), 5)
    ref = _oracle().forward(w, cfg, x)
    assert float(ref.max()) > 0.1
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    model = model.to("cuda", torch.float32)
    errs = {}
    for precision, tol in (("bf16x3", REL_TOL_X3), ("fp16x3", REL_TOL_X3), ("mixed", REL_TOL_MIXED), ("fp16", REL_TOL_FP16),
                           ("bf16", emulated_tol(w, cfg, x, "bf16"))):
        model.set_precision(precision)
        y = model(x.cuda())
        assert bool(torch.isfinite(y).all())
        errs[precision] = rel_err(y.cpu(), ref)
        assert errs[precision] <= tol, f"{precision}: {errs[precision]:.3e} > {tol:.3e}"
    # the compensation must not hurt in this regime (rms: the max of the single-pass fp16 error is set by the decoder and moves by +-30 % with
    # any change upstream; measured max 2.1e-3 on vs 1.6e-3 off on this input, both inside REL_TOL_FP16)
    rms = lambda y: float((y.double() - ref.double()).pow(2).mean().sqrt())  # noqa: E731
    model.set_precision("fp16")
    model.set_weight_rounding_compensation(True)
    r_on = rms(model(x.cuda()).cpu())
    model.set_weight_rounding_compensation(False)
    r_off = rms(model(x.cuda()).cpu())
    assert r_on <= r_off * 1.05, f"compensation on: rms {r_on:.3e}, off: {r_off:.3e}"


def test_swinv2_token_mean_compensation_tile_variants_and_batch_invariance(golden_dir):
    """SwinV2 with the per-image bias tables (round 4: the fused Q / K / V epilogues of its QKV GEMM take them too, for stages of >= 256 tokens per
    image; smaller stages go through the strip epilogues): every GEMM tile variant gives the same bits, an image's result does not depend on its
    batch. Accuracy: the toy stages have 384 / 96 / 24 / 6 tokens per image - a mean over that few tokens is a noisy estimate of the shared
    component and there is little averaging behind it, so here the compensated run is only required to stay within 25 % (rms) of the
    uncompensated one (measured: fp16 equal, mixed +10 %); that it pays at full size is asserted on the SwinV2-L fixture above."""
    from muggled_dpt_amd import make_swinv2_dpt_from_midas_v31_state_dict
    from muggled_dpt_amd.synthetic import make_synthetic_swinv2_state_dict
    g = np.load(os.path.join(golden_dir, "swin2_tiny.npz"))
    osd = make_synthetic_swinv2_state_dict("swin2_tiny", int(g["weight_seed"]))
    x = torch.from_numpy(g["wide_input"])  # 64 x 96 image: 384 tokens in stage 0, 96 / 24 / 6 in the later stages
    x3 = torch.cat((x, x.flip(0), x[:1] * 0.5), dim=0)
    ref = torch.from_numpy(g["wide_depth"]).double()
    _, model = make_swinv2_dpt_from_midas_v31_state_dict(osd)
    model = model.to("cuda", torch.float32)
    for precision in ("fp16", "mixed"):
        model.set_precision(precision)
        model.set_weight_rounding_compensation(True)  # (the default of "mixed"; opt-in for "fp16")
        y = model(x3.cuda())
        for tile in (1, 2, 4, 5, 6):
            model.set_gemm_tile(tile)
            assert torch.equal(model(x3.cuda()), y), f"{precision}: tile variant {tile} changed the result"
        model.set_gemm_tile(0)
        for i in range(x3.shape[0]):
            assert torch.equal(model(x3[i:i + 1].cuda())[0], y[i]), f"{precision}: image {i} depends on its batch"
        rms = lambda t: float((t.cpu().double() - ref).pow(2).mean().sqrt())  # noqa: E731
        r_on = rms(y[:x.shape[0]])
        model.set_weight_rounding_compensation(False)
        r_off = rms(model(x.cuda()))
        model.set_weight_rounding_compensation(None)
        assert r_on <= r_off * 1.25, f"{precision}: rms {r_on:.3e} with the compensation, {r_off:.3e} without"

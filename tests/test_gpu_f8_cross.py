"""fp8 cross terms (MDPT_PASSES_2F8 / _3F8, csrc/f8_cross.h): the split products A_hi W_hi + A_lo W_hi (+ A_hi W_lo) with the cross terms on
e5m2 activation planes / row-scaled e4m3 weight planes through gfx950's block-scaled MFMA. Run with `pytest -m gpu` on an MI355X.

The yardstick is the CPU emulation of exactly this arithmetic (tests/precision_budget/emulate_operand_rounding.py, format "sf8": the same
quantisers - e5m2 of the residue times 2^16, e5m2 of the values, e4m3 with one power-of-two scale per output row - applied to the operands of
the oracle's contractions, products in fp32). GPU and emulation round the same values the same way, so they differ by accumulation order and
by where a bilinear upsample sits relative to a rounding (the library applies the 1x1 fusion projection before the x2 upsample: exact in real
arithmetic, a different operand to round) - two orders of magnitude below what a wrong scale, a wrong K order or a missing term would show.

Configuration: a Depth-Anything-shaped toy that IS eligible for the fp8 forms (every contraction length a multiple of 128): 128 features, 2
heads, 4 blocks, reassembly widths 128 / 128 / 256 / 256, fusion width 256. ViT-S / ViT-B widths are not multiples of 128: they run the fp16-plane
form of the same term count (test_ineligible_configurations_fall_back...)."""
import ctypes
import os
import sys

import pytest
import torch

from tests.helpers import rel_err, seeded_input

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "precision_budget"))

F8_TOY = dict(features_per_token=128, num_heads=2, num_blocks=4, reassembly_features_list=[128, 128, 256, 256], base_patch_grid_hw=(5, 5),
              fusion_channels=256, patch_size_px=14)
_CACHE = {}


def _toy(seed=0):
    from muggled_dpt_amd.state_dict_conversion import convert_state_dict_keys, flatten_components, get_model_config_from_state_dict
    from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
    if seed not in _CACHE:
        osd = make_synthetic_original_state_dict(F8_TOY, seed)
        cfg = get_model_config_from_state_dict(osd)
        _CACHE[seed] = (osd, cfg, flatten_components(convert_state_dict_keys(cfg, osd)))
    return _CACHE[seed]


def _model(passes: dict | None, precision="fp16", seed=0):
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    osd, cfg, w = _toy(seed)
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    model = model.to("cuda", torch.float32)
    model.set_precision(precision)
    model.set_weight_rounding_compensation(False)
    if passes:
        model.set_class_passes(passes)
    return model, cfg, w


def _emu():
    import emulate_operand_rounding as emu
    return emu


def _oracle():
    from oracle import dpt_oracle
    return dpt_oracle


def _policy(emu, **classes):
    p = {c: "f16" for c in emu.CLASSES + emu.FINE_CLASSES}
    p.update(classes)
    return p


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from muggled_dpt_amd import native
    native.load()


def test_f8_pass_values_are_accepted_for_their_classes_only_and_reported():
    from muggled_dpt_amd import native
    lib = native.load()
    model, _, _ = _model({"fusion": native.PASSES_3F8, "head": native.PASSES_2F8, "reasm": native.PASSES_3F8, "fusion_proj": native.PASSES_3F8, "fusion_in": native.PASSES_2F8})
    model(seeded_input((1, 3, 56, 56)).cuda())
    h = model._get_engine().handle
    got = ctypes.c_int(-1)
    for cls in native.OP_CLASSES:
        native.check(lib, lib.mdpt_get_class_f8(h, native.OP_CLASSES.index(cls), ctypes.byref(got)))
        assert got.value == (1 if cls in native.F8_CLASSES else 0), cls
    for cls in ("patch", "qkv", "attn", "proj", "fc1", "fc2", "head_tail"):
        assert lib.mdpt_set_class_passes(h, native.OP_CLASSES.index(cls), native.PASSES_2F8) != 0, cls
        with pytest.raises(ValueError):
            model.set_class_passes({cls: native.PASSES_3F8})
    # bf16 operands: the value is accepted, the class runs the 16-bit-plane form of the same term count
    model_b, _, _ = _model({"fusion": native.PASSES_3F8}, precision="bf16")
    y = model_b(seeded_input((1, 3, 56, 56)).cuda())
    native.check(lib, lib.mdpt_get_class_f8(model_b._get_engine().handle, native.OP_CLASSES.index("fusion"), ctypes.byref(got)))
    assert got.value == 0 and torch.isfinite(y).all()


def test_ineligible_configurations_fall_back_to_the_fp16_plane_form_bitwise():
    """ViT-S: reassembly widths 48 ... 384 and fusion width 64 are not multiples of the 128-element fp8 K tile - PASSES_3F8 IS three fp16-plane passes there."""
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict, native
    from tests.helpers import synthetic_model
    osd, _, _ = synthetic_model("vits")
    x = seeded_input((2, 3, 112, 112), 3).cuda()
    outs = []
    for v in (3, native.PASSES_3F8):
        _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
        model = model.to("cuda", torch.float32)
        model.set_precision("fp16")
        model.set_class_passes({c: v for c in native.F8_CLASSES})
        outs.append(model(x))
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("batch,size", [(2, 56), (32, 112)])  # small: lockstep tiles; big: 8-phase GEMM + halo-staged conv kernels (two-stream split)
def test_fusion_blocks_with_f8_cross_terms_match_the_cpu_emulation(batch, size):
    from muggled_dpt_amd import native
    emu, orc = _emu(), _oracle()
    # (head = 3: with a single-pass head class block 0 hands over a 16-bit map - the head's input - instead of the fp32 one)
    model, cfg, w = _model({"fusion": native.PASSES_3F8, "fusion_in": native.PASSES_3F8, "fusion_proj": native.PASSES_3F8, "head": 3})
    model16, _, _ = _model({"fusion": 3, "fusion_in": 3, "fusion_proj": 3, "head": 3})
    model1, _, _ = _model({"head": 3})
    g = size // 14
    sizes = [4 * g, 2 * g, g, g // 2]
    gen = torch.Generator().manual_seed(5)
    reasm = [torch.randn(batch, 256, s, s, generator=gen) * 2.0 for s in sizes]
    pol = _policy(emu, fusion="f16x3@sf8")
    prev = None
    for i in (3, 2, 1, 0):
        chk = slice(0, batch, max(1, batch - 1))  # images 0 and batch - 1 on the CPU
        args = (reasm[i][chk], None if prev is None else prev[chk])
        ref_emu = emu.emulated_call(orc.fusion_block, w, pol, w, i, *args)
        ref_f32 = orc.fusion_block(w, i, *args)
        dev = (reasm[i].cuda(),) if prev is None else (reasm[i].cuda(), prev.cuda())
        y8 = model.fusion.blocks[i](*dev).cpu()[chk]
        y16 = model16.fusion.blocks[i](*dev).cpu()[chk]
        y1 = model1.fusion.blocks[i](*dev).cpu()[chk]
        e_emu, e_16, e_1 = rel_err(y8, ref_emu), rel_err(y8, y16), rel_err(y1, y16)
        print(f"block {i} batch {batch}: fp8 cross terms vs emulation {e_emu:.2e}, vs fp16 cross terms {e_16:.2e}; dropping the cross terms {e_1:.2e}; "
              f"vs fp32 {rel_err(y8, ref_f32):.2e} (fp16 cross terms {rel_err(y16, ref_f32):.2e})")
        # (the library applies the 1x1 projection before the x2 upsample, the emulation after it: other values get rounded, the two differ at
        #  the level of the fp8 rounding itself - head conv 1 and the reassembly convs, which have no such step, agree to 1e-5 and better)
        assert e_emu <= 6e-5, f"block {i}"
        assert e_16 <= 0.25 * e_1, f"block {i}: the fp8 cross terms must carry what the fp16 cross terms carry"
        assert rel_err(y8, ref_f32) <= 0.15 * e_1, f"block {i}: what the fp8 rounding of the cross terms leaves (measured 5 ... 7 % of dropping them)"
        prev = orc.fusion_block(w, i, reasm[i], prev)  # feed the exact previous map so errors do not compound


@pytest.mark.parametrize("batch,size", [(2, 56), (32, 112)])
def test_head_conv1_with_f8_cross_terms_matches_the_cpu_emulation(batch, size):
    from muggled_dpt_amd import native
    emu, orc = _emu(), _oracle()
    emu.FINE = True
    try:
        for terms, f8v, pol_mode in ((2, native.PASSES_2F8, "f16x2a@sf8"), (3, native.PASSES_3F8, "f16x3@sf8")):
            model, cfg, w = _model({"head": f8v, "head_tail": 3})
            model16, _, _ = _model({"head": terms, "head_tail": 3})
            model1, _, _ = _model({"head_tail": 3})
            fused = torch.randn(batch, 256, 8 * (size // 14), 8 * (size // 14), generator=torch.Generator().manual_seed(9)) * 1.5
            chk = slice(0, batch, max(1, batch - 1))
            pol = _policy(emu, head_conv1=pol_mode, head_conv2="f16x3")
            ref_emu = emu.emulated_call(orc.head, w, pol, w, cfg, fused[chk])
            y8 = model.head(fused.cuda()).cpu()[chk]
            y16 = model16.head(fused.cuda()).cpu()[chk]
            y1 = model1.head(fused.cuda()).cpu()[chk]
            e_emu, e_16, e_1 = rel_err(y8, ref_emu), rel_err(y8, y16), rel_err(y1, y16)
            print(f"head, {terms} terms, batch {batch}: fp8 cross terms vs emulation {e_emu:.2e}, vs fp16 cross terms {e_16:.2e}; dropping the cross terms {e_1:.2e}")
            assert e_emu <= 4e-5
            assert e_16 <= 0.25 * e_1
    finally:
        emu.FINE = False


@pytest.mark.parametrize("batch,size", [(2, 56), (32, 112)])
def test_reassembly_with_f8_cross_terms_matches_the_cpu_emulation(batch, size):
    """1x1 on token rows (cls row skipped by the A-row generator), transposed conv as GEMM + depth-to-space, 3x3 stride 2, 3x3 projection."""
    from muggled_dpt_amd import native
    emu, orc = _emu(), _oracle()
    model, cfg, w = _model({"reasm": native.PASSES_3F8})
    model16, _, _ = _model({"reasm": 3})
    model1, _, _ = _model(None)
    g = size // 14
    gen = torch.Generator().manual_seed(13)
    toks = [torch.randn(batch, 1 + g * g, 128, generator=gen) for _ in range(4)]
    chk = slice(0, batch, max(1, batch - 1))
    pol = _policy(emu, reasm="f16x3@sf8")
    ref_emu = emu.emulated_call(orc.reassemble, w, pol, w, [t[chk] for t in toks], (g, g))
    dev = [t.cuda() for t in toks]
    y8, y16, y1 = model.reassemble(*dev, (g, g)), model16.reassemble(*dev, (g, g)), model1.reassemble(*dev, (g, g))
    for i in range(4):
        a, b, c1 = y8[i].cpu()[chk], y16[i].cpu()[chk], y1[i].cpu()[chk]
        e_emu, e_16, e_1 = rel_err(a, ref_emu[i]), rel_err(a, b), rel_err(c1, b)
        print(f"reassembly map {i}, batch {batch}: fp8 cross terms vs emulation {e_emu:.2e}, vs fp16 cross terms {e_16:.2e}; dropping the cross terms {e_1:.2e}")
        assert e_emu <= 4e-5, i
        assert e_16 <= 0.25 * e_1, i


def test_whole_model_with_f8_decoder_is_batch_invariant_and_as_accurate_as_fp16_cross_terms():
    """The fp8 forms decided per class never depend on the batch; an image's depth map does not depend on the batch it is part of (tile rules:
    lockstep 32x32x64 MFMAs at batch 1, 8-phase / halo-staged 16x16x128 MFMAs at batch 32 - tools/probes/f8_shape_equiv_probe.hip)."""
    from muggled_dpt_amd import native
    orc = _oracle()
    f8 = {"reasm": native.PASSES_3F8, "fusion": native.PASSES_3F8, "fusion_in": native.PASSES_2F8, "fusion_proj": native.PASSES_3F8, "head": native.PASSES_2F8, "head_tail": 2}
    f16 = {"reasm": 3, "fusion": 3, "fusion_in": 2, "fusion_proj": 3, "head": 2, "head_tail": 2}
    model, cfg, w = _model(f8)
    model16, _, _ = _model(f16)
    x = seeded_input((32, 3, 112, 112), 17)
    ref = orc.forward(w, cfg, x[[0, 31]])
    y8, y16 = model(x.cuda()).cpu(), model16(x.cuda()).cpu()
    e8, e16 = rel_err(y8[[0, 31]], ref), rel_err(y16[[0, 31]], ref)
    print(f"whole model, decoder classes with fp8 cross terms: {e8:.2e} vs fp32; with fp16 cross terms {e16:.2e}")
    assert e8 <= 1.25 * e16 + 2e-5
    for b in (1, 3, 8):
        yb = model(x[:b].cuda()).cpu()
        assert torch.equal(yb, y8[:b]), f"batch {b} differs from the same images inside batch 32"

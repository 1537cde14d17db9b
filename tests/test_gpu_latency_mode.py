"""Latency mode (mdpt_set_latency_mode), round 4: fc2 of a small batch splits K into two fixed halves; the LayerNorm that follows folds the
second half's partial sums into the residual stream (gemm.hip GemmParams::ksplit, elementwise.hip layernorm_addp_kernel, mdpt_stages.cpp
run_encoder). Checked here: the split computes the same sums (fp32-class mode against the oracle at 1e-4), for every way the residual
stream is consumed behind an fc2 (next block's LN1, the shared out-norm of a tap, both in a row for Depth-Anything V1's consecutive taps,
BEiT's un-normed taps and hooked blocks - where the split must stand down), and the default mode is untouched. `pytest -m gpu`."""
import pytest
import torch

from tests.helpers import REL_TOL_BF16, REL_TOL_X3, rel_err, seeded_input, synthetic_model

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import dpt_oracle
    return dpt_oracle


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from muggled_dpt_amd import native
    native.load()


def _split_from(model, k_tiles: int, big_k_tiles: int = 1 << 30):
    """Toy models have 4 K tiles in fc2 and 1 in proj (the production thresholds are far above): lower them through the test hook on the model's
    engine. big_k_tiles: from this many K tiles on the split is in four."""
    from muggled_dpt_amd import native
    eng = model._get_engine()
    native.check(eng.lib, eng.lib.mdpt_debug_set_ksplit_min(eng.handle, k_tiles, big_k_tiles))


def _profile_names(fn):
    """kernel names the library launched while fn() ran (mdpt_profile_enable / report)"""
    import ctypes
    import json
    from muggled_dpt_amd import native
    lib = native.load()
    native.check(lib, lib.mdpt_profile_enable(1))
    try:
        fn()
        torch.cuda.synchronize()
        buf = ctypes.create_string_buffer(1 << 20)
        native.check(lib, lib.mdpt_profile_report(buf, len(buf)))
    finally:
        native.check(lib, lib.mdpt_profile_enable(0))
    rep = json.loads(buf.value.decode())
    return {k["name"]: k for k in rep["kernels"]}


def test_vits_batch1_k_split_is_exact_in_the_fp32_class_mode_and_runs():
    """ViT-S at 504x504, batch 1 (BASELINE configs[1]): fc2 has K = 1536 = 24 K tiles -> split. fp32-class arithmetic (hi + lo planes, three
    passes per half) against the CPU oracle: the reassociated sum is the same sum."""
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    osd, cfg, w = synthetic_model("vits", 0)
    x = seeded_input((1, 3, 504, 504), 21)
    ref = _oracle().forward(w, cfg, x)
    for dtype, tol in ((torch.float32, REL_TOL_X3), (torch.bfloat16, REL_TOL_BF16)):
        _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
        model = model.to("cuda", dtype)
        xd = x.to("cuda", dtype)
        y_default = model(xd)
        model.set_latency_mode(True)
        names = _profile_names(lambda: model(xd))
        assert "layernorm_addp_kernel" in names, sorted(names)  # the split form did run
        y_fast = model(xd)
        assert rel_err(y_fast.float().cpu(), ref) <= tol
        # a fixed split: batch 2 gives image 0 the same K-split sums (the attention kernel's own latency form is chosen by launch size, so
        # this is asserted where it does not switch: fp32-class outputs agree to rounding level, not bitwise)
        assert rel_err(model(torch.cat([xd, xd]))[0].float().cpu(), y_fast[0].float().cpu()) <= (1e-5 if dtype == torch.float32 else 2e-2)
        model.set_latency_mode(False)
        assert torch.equal(model(xd), y_default)  # the default (batch-invariant) form is reproduced exactly
        names = _profile_names(lambda: model(xd))
        assert "layernorm_addp_kernel" not in names


@pytest.mark.parametrize("big", [False, True])
@pytest.mark.parametrize("family", ["v2", "v1", "beit"])
def test_every_consumer_of_the_residual_stream_behind_a_split_fc2(family, big):
    """Threshold lowered to the toy models' 4 K tiles. v2: LN1 of the next block / the out-norm of a tap block then LN1; v1 (8 blocks, taps on the
    last four): out-norm and LN1 alternate; BEiT: tap blocks export the raw stream (no LayerNorm) - the split stands down there."""
    import muggled_dpt_amd as mda
    from muggled_dpt_amd.state_dict_conversion import flatten_components
    orc = _oracle()
    if family == "beit":
        from muggled_dpt_amd import state_dict_conversion_beit as conv
        from muggled_dpt_amd.synthetic import make_synthetic_beit_state_dict
        osd = make_synthetic_beit_state_dict("beit_tiny", 3)
        cfg, model = mda.make_beit_dpt_from_midas_v31_state_dict(osd)
        w = flatten_components(conv.convert_state_dict_keys(cfg, osd))
        x = seeded_input((2, 3, 64, 96), 5)
        ref = orc.forward(w, cfg, x)
        model = model.to("cuda", torch.float32)
    elif family == "v1":
        from muggled_dpt_amd.synthetic import STANDARD_CONFIGS, make_synthetic_original_state_dict
        osd = make_synthetic_original_state_dict(dict(STANDARD_CONFIGS["tiny"], num_blocks=8), 3)
        cfg, model = mda.make_depthanythingv1_dpt_from_original_state_dict(osd)
        model = model.to("cuda", torch.float32)
        x = seeded_input((2, 3, 56, 84), 5)
        y_default = model(x.cuda())
        ref = y_default.float().cpu()  # the v1 default path is pinned to the reference fixture elsewhere (test_depth_anything_v1_family)
    else:
        osd, cfg, w = synthetic_model("tiny", 0)
        cfg, model = mda.make_depthanythingv2_dpt_from_original_state_dict(osd)
        model = model.to("cuda", torch.float32)
        x = seeded_input((2, 3, 56, 84), 5)
        ref = orc.forward(w, cfg, x)
    y_default = model(x.cuda())
    model.set_latency_mode(True)
    _split_from(model, 2, 4 if big else 1 << 30)  # big: fc2 (4 K tiles) in four ranges of one K tile, three partial-sum planes
    names = _profile_names(lambda: model(x.cuda()))
    assert "layernorm_addp_kernel" in names, sorted(names)
    y = model(x.cuda())
    assert rel_err(y.float().cpu(), ref) <= REL_TOL_X3
    assert rel_err(y.float().cpu(), y_default.float().cpu()) <= REL_TOL_X3
    assert not torch.equal(y, y_default) or family == "v1"  # a different summation order (the toy v1 sums may coincide)


def test_block_hooks_and_stage_calls_in_latency_mode():
    """A hooked block's output is exported raw (mdpt_encoder_probe_blocks): its fc2 must not leave partial sums pending. Stage-level
    calls (patch_embed -> imgencoder) go through the same encoder driver."""
    import muggled_dpt_amd as mda
    orc = _oracle()
    osd, cfg, w = synthetic_model("tiny", 0)
    cfg, model = mda.make_depthanythingv2_dpt_from_original_state_dict(osd)
    model = model.to("cuda", torch.float32)
    model.set_latency_mode(True)
    x = seeded_input((2, 3, 56, 84), seed=13)
    y_plain = model(x.cuda())  # engine exists now
    _split_from(model, 2)
    bps = cfg["num_blocks"] // 4
    blocks = [model.imgencoder.stages[i // bps].blocks[i % bps] for i in range(cfg["num_blocks"])]
    got = {}
    handles = [blocks[i].register_forward_hook(lambda mod, args, out, i=i: got.__setitem__(i, out)) for i in (0, len(blocks) - 1)]
    y = model(x.cuda())
    tokens, grid = orc.patch_embed(w, x)
    ref_blocks = []
    orc.image_encoder(w, cfg, tokens, grid, block_outputs=ref_blocks)
    for i in (0, len(blocks) - 1):
        assert rel_err(got[i].float().cpu(), ref_blocks[i]) <= 1e-4, f"block {i}"
    assert rel_err(y.float().cpu(), orc.forward(w, cfg, x)) <= REL_TOL_X3
    for h in handles:
        h.remove()
    tk, hw = model.patch_embed(x.cuda())
    taps = model.imgencoder(tk, hw)
    ref_taps = orc.image_encoder(w, cfg, tokens, grid)
    for t, r in zip(taps, ref_taps):
        assert rel_err(t.float().cpu(), r) <= 1e-4
    del y_plain


def test_latency_mode_is_deterministic_run_to_run():
    """The K split of latency mode is a fixed split whose partial sums are added by the NEXT kernel (the LayerNorm) in a fixed order: 25 forwards of
    ViT-S at batch 1 are bitwise equal, also with another shape in between. (A form that reduced the K ranges of the small decoder convs INSIDE the
    kernel - last workgroup to arrive - failed exactly this check in 1-10 % of the forwards until it carried device-scope fences that cost what it
    saved, and was removed: DESIGN.md section 3, tools/probes/gpu_ksplit_determinism.py.)"""
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    osd, cfg, w = synthetic_model("vits", 0)
    x = seeded_input((1, 3, 504, 504), 33)
    ref = _oracle().forward(w, cfg, x)
    for dtype, tol in ((torch.float32, REL_TOL_X3), (torch.bfloat16, REL_TOL_BF16)):
        _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
        model = model.to("cuda", dtype)
        model.set_latency_mode(True)
        xd = x.to("cuda", dtype)
        y0 = model(xd)
        assert rel_err(y0.float().cpu(), ref) <= tol
        for _ in range(25):
            assert torch.equal(model(xd), y0)
        x2 = seeded_input((1, 3, 252, 364), 34).to("cuda", dtype)
        y2 = model(x2)
        assert torch.equal(model(xd), y0) and torch.equal(model(x2), y2)


def test_long_k_decoder_convs_split_with_a_finishing_kernel():
    """Latency mode, batch 1 of the 1024-wide models: the 3x3 convs of the 18^2 / 36^2 levels (144 K tiles on 24 ... 96 workgroups) run as K ranges that
    store partial planes + ksplit_finish_kernel (mdpt_stages.cpp try_ksplit_conv). ViT-B-sized decoder here (hidden 768 at the coarse levels: 108 K
    tiles) at 252x252: fp32-class result against the oracle at 1e-4, bf16 inside its tolerance, run-to-run bitwise equal."""
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    osd, cfg, w = synthetic_model("vitb", 0)
    x = seeded_input((1, 3, 252, 252), 35)
    ref = _oracle().forward(w, cfg, x)
    for dtype, tol in ((torch.float32, REL_TOL_X3), (torch.bfloat16, REL_TOL_BF16)):
        _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
        model = model.to("cuda", dtype)
        xd = x.to("cuda", dtype)
        y_default = model(xd)
        model.set_latency_mode(True)
        names = _profile_names(lambda: model(xd))
        assert "ksplit_finish_kernel" in names, sorted(names)
        y = model(xd)
        assert rel_err(y.float().cpu(), ref) <= tol
        for _ in range(10):
            assert torch.equal(model(xd), y)
        model.set_latency_mode(False)
        assert torch.equal(model(xd), y_default)
        assert "ksplit_finish_kernel" not in _profile_names(lambda: model(xd))


@pytest.mark.parametrize("family", ["v2", "v1", "beit"])
def test_reassembly_branches_on_the_side_stream_change_no_bit(family):
    """Unsplit (small-batch) forwards queue every reassembly branch (reassembly_model.py:61-94: four independent branches) on the handle's side
    stream as soon as its encoder tap exists, beside the remaining blocks, and join before the fusion stage. Same kernels: bitwise equal to
    the one-stream order (mdpt_debug_set_reassemble_overlap 0) in the default mode (latency mode: same accuracy class), at batches 1 / 3 / 7 and several sizes,
    repeated (no race: 20 forwards each), and capturable into a hipGraph like the batch split. (The toggle's 2 = always; the library's own rule,
    1, takes the side stream only for encoders of width >= 1024 outside latency mode, where it measured faster.)"""
    import muggled_dpt_amd as m
    from muggled_dpt_amd import native
    from muggled_dpt_amd.synthetic import STANDARD_CONFIGS, make_synthetic_beit_state_dict, make_synthetic_original_state_dict
    if family == "beit":
        model, unit = m.make_beit_dpt_from_midas_v31_state_dict(make_synthetic_beit_state_dict("beit_tiny", 0))[1], 32
    elif family == "v1":
        model, unit = m.make_depthanythingv1_dpt_from_original_state_dict(make_synthetic_original_state_dict(dict(STANDARD_CONFIGS["tiny"], num_blocks=8), 0))[1], 28
    else:
        model, unit = m.make_depthanythingv2_dpt_from_original_state_dict(make_synthetic_original_state_dict("tiny", 0))[1], 28
    for dtype in (torch.bfloat16, torch.float32):
        model = model.to("cuda", dtype)
        eng = model._get_engine()
        for latency in (False, True):
            model.set_latency_mode(latency)
            for b, hh, ww in ((1, 2 * unit, 2 * unit), (3, 4 * unit, 2 * unit), (7, 2 * unit, 6 * unit)):
                x = torch.randn(b, 3, hh, ww, generator=torch.Generator().manual_seed(b)).to("cuda", dtype)
                native.check(eng.lib, eng.lib.mdpt_debug_set_reassemble_overlap(eng.handle, 0))
                want = model(x)
                native.check(eng.lib, eng.lib.mdpt_debug_set_reassemble_overlap(eng.handle, 2))
                first = model(x)
                if latency:  # the K split of the long-K decoder convs (a finishing-kernel form that owns the encoder's partial-sum planes) stands down
                    # on the side stream: another fixed summation order, as latency mode allows - same accuracy class, still deterministic
                    assert float((first.float() - want.float()).abs().max()) <= (2e-2 if dtype == torch.bfloat16 else 1e-4) * float(want.float().abs().max())
                else:
                    assert torch.equal(first, want), f"{family} {dtype} batch {b}"
                for _ in range(20):
                    assert torch.equal(model(x), first), f"{family} {dtype} latency={latency} batch {b}: not reproducible"
        model.set_latency_mode(False)
        x = torch.randn(1, 3, 2 * unit, 2 * unit, generator=torch.Generator().manual_seed(1)).to("cuda", dtype)
        want = model(x)  # (overlap still forced on)
        g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            model(x)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            y_graph = model(x)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(y_graph, want)


def test_side_stream_is_probed_to_run_beside_the_callers_stream():
    """The runtime multiplexes streams onto a few hardware queues; a side stream on the caller's queue silently serialises the split batch
    (csrc/stream_probe.hip). Whatever other streams the process holds, the handle's probe finds a candidate that the GPU runs concurrently
    (rejected < candidates), per caller stream, and the choice changes no bit of the result."""
    import ctypes
    import muggled_dpt_amd as m
    from muggled_dpt_amd import native
    from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
    hip = ctypes.CDLL("libamdhip64.so")
    x = torch.randn(8, 3, 56, 84, generator=torch.Generator().manual_seed(2)).to("cuda", torch.bfloat16)
    want, keep = None, []
    for extra in range(6):
        for probe in (1, 0):
            model = m.make_depthanythingv2_dpt_from_original_state_dict(make_synthetic_original_state_dict("tiny", 0))[1].to("cuda", torch.bfloat16)
            eng = model._get_engine()
            native.check(eng.lib, eng.lib.mdpt_debug_set_side_stream_probe(eng.handle, probe))
            y = model(x)  # batch 8: split in two halves, the second on the side stream
            other = torch.cuda.Stream()
            other.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(other):
                y2 = model(x)
            torch.cuda.current_stream().wait_stream(other)
            c, r = ctypes.c_int32(), ctypes.c_int32()
            native.check(eng.lib, eng.lib.mdpt_debug_side_stream_info(eng.handle, ctypes.byref(c), ctypes.byref(r)))
            if probe:
                assert 1 <= c.value <= 4 and r.value < 2 * c.value, (c.value, r.value)  # (two caller streams probed)
                assert r.value <= 2 * (c.value - 1), "no candidate ran beside the caller's stream"
            else:
                assert (c.value, r.value) == (1, 0)
            want = y if want is None else want
            assert torch.equal(y, want) and torch.equal(y2, want)
            del model
        s = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
        keep.append(s)
    for s in keep:
        hip.hipStreamDestroy(s)

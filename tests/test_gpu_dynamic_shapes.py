"""Dynamic-shape conformance: ONE finalized handle, a sequence of different (batch, height, width) calls - the in-scope counterpart of the
reference's ONNX export contract (experiments/export_onnx.py:119-148: one graph with dynamic batch / height / width axes). Nothing is
re-created or re-finalized between calls; the workspace shrinks and grows; every result is checked against the oracle and, where a shape
repeats, against its first result bit for bit. `pytest -m gpu`."""
import ctypes

import pytest
import torch

from tests.helpers import REL_TOL_BF16_TOY, REL_TOL_X3, rel_err, seeded_input, synthetic_model

pytestmark = pytest.mark.gpu

SEQUENCE = [(1, 56, 56), (3, 84, 140), (1, 28, 28), (8, 112, 56), (2, 56, 56), (1, 252, 196), (1, 56, 56), (5, 28, 84), (3, 84, 140)]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, REL_TOL_X3), (torch.bfloat16, REL_TOL_BF16_TOY)])
def test_one_handle_many_shapes_without_refinalize(dtype, tol):
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    from oracle import dpt_oracle
    osd, cfg, w = synthetic_model("tiny", 0)
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    model = model.to("cuda", dtype)
    eng = model._get_engine()
    handle_before, packed_before = eng.handle.value, eng.packed.data_ptr()
    first = {}
    for k, (b, h, wd) in enumerate(SEQUENCE):
        x = seeded_input((b, 3, h, wd), 100 + hash((b, h, wd)) % 1000)
        y = model(x.to("cuda", dtype))
        assert tuple(y.shape) == (b, h, wd) and y.dtype == dtype
        assert rel_err(y.float().cpu(), dpt_oracle.forward(w, cfg, x)) <= tol, (k, b, h, wd)
        if (b, h, wd) in first:
            assert torch.equal(y, first[(b, h, wd)]), f"call {k}: shape {(b, h, wd)} seen before gives different bits"
        first[(b, h, wd)] = y
    eng2 = model._get_engine()
    assert eng2 is eng and eng.handle.value == handle_before and eng.packed.data_ptr() == packed_before, "the engine was re-created"


def test_raw_c_abi_workspace_reuse_across_shapes():
    """The same through the bare C ABI with ONE caller-owned workspace sized for the largest call (what an exported-graph runtime does):
    mdpt_workspace_bytes per shape, mdpt_forward on a buffer that is larger than needed, smaller shapes after larger ones."""
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict, native
    from oracle import dpt_oracle
    osd, cfg, w = synthetic_model("tiny", 0)
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    model = model.to("cuda", torch.float32)
    eng = model._get_engine()
    lib = eng.lib
    need = []
    for (b, h, wd) in SEQUENCE:
        n = ctypes.c_size_t()
        native.check(lib, lib.mdpt_workspace_bytes(eng.handle, b, h, wd, ctypes.byref(n)))
        need.append(n.value)
    ws = torch.empty(max(need) + 256, dtype=torch.uint8, device="cuda")
    ws_ptr = (ws.data_ptr() + 255) & ~255
    stream = torch.cuda.current_stream().cuda_stream
    for (b, h, wd), n in zip(SEQUENCE, need):
        x = seeded_input((b, 3, h, wd), 7 + b + h + wd)
        xd = x.cuda()
        out = torch.empty((b, h, wd), device="cuda")
        ws.fill_(0xA5)  # stale contents of the previous (larger or smaller) call must not matter
        native.check(lib, lib.mdpt_forward(eng.handle, xd.data_ptr(), native.DTYPE_F32, b, h, wd, out.data_ptr(), native.DTYPE_F32, ws_ptr, max(need), stream))
        assert rel_err(out.cpu(), dpt_oracle.forward(w, cfg, x)) <= REL_TOL_X3, (b, h, wd)
        # a workspace that is too small for this shape is refused, not overrun
        rc = lib.mdpt_forward(eng.handle, xd.data_ptr(), native.DTYPE_F32, b, h, wd, out.data_ptr(), native.DTYPE_F32, ws_ptr, n // 2, stream)
        assert rc == -5, rc
    torch.cuda.synchronize()


def test_bf16_and_fp16_tensors_cross_the_c_abi_without_casts():
    """mdpt_forward takes dtype-tagged image / depth pointers: a bf16 (fp16) model hands its tensors over as they are and gets the depth
    back in the model dtype (dpt_model.py:105-107); mixing is allowed at the ABI (fp32 image -> bf16 depth)."""
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict, native
    from oracle import dpt_oracle
    osd, cfg, w = synthetic_model("tiny", 0)
    x = seeded_input((2, 3, 56, 84), 5)
    ref = dpt_oracle.forward(w, cfg, x)
    for dtype in (torch.bfloat16, torch.float16):
        _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
        model = model.to("cuda", dtype)
        y = model(x.to("cuda", dtype))
        assert y.dtype == dtype and rel_err(y.float().cpu(), ref) <= REL_TOL_BF16_TOY
        eng = model._get_engine()
        n = ctypes.c_size_t()
        native.check(eng.lib, eng.lib.mdpt_workspace_bytes(eng.handle, 2, 56, 84, ctypes.byref(n)))
        ws = torch.empty(n.value + 256, dtype=torch.uint8, device="cuda")
        xd32 = x.to(dtype).float().cuda()  # the same (rounded) pixel values as an fp32 tensor
        out = torch.empty((2, 56, 84), device="cuda", dtype=dtype)
        native.check(eng.lib, eng.lib.mdpt_forward(eng.handle, xd32.data_ptr(), native.DTYPE_F32, 2, 56, 84, out.data_ptr(), native.dtype_code(dtype),
                                                   (ws.data_ptr() + 255) & ~255, n.value, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert torch.equal(out, y), "the image dtype tag only changes how the pixels are read"


def test_in_place_parameter_edits_are_picked_up():
    """The engine packs a snapshot of the weights; an in-place edit of a parameter after the first forward (p.data.copy_, nn.init, mul_
    under no_grad) must be visible in the next forward, as in the reference, which reads the live tensors."""
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    from oracle import dpt_oracle
    osd, cfg, w = synthetic_model("tiny", 0)
    _, model = make_depthanythingv2_dpt_from_original_state_dict(osd)
    model = model.to("cuda", torch.float32)
    x = seeded_input((1, 3, 56, 56), 3)
    y0 = model(x.cuda())
    with torch.no_grad():
        model.head.proj_1ch[2].bias.add_(0.25)
    y1 = model(x.cuda())
    w2 = dict(w)
    w2["head.proj_1ch.2.bias"] = w["head.proj_1ch.2.bias"] + 0.25
    assert not torch.equal(y0, y1) and rel_err(y1.cpu(), dpt_oracle.forward(w2, cfg, x)) <= REL_TOL_X3
    assert model._get_engine() is model._get_engine(), "no edit, no re-pack"


@pytest.mark.parametrize("family", ["v2", "beit", "swinv2"])
def test_enable_cache_keeps_per_grid_constants_between_forwards_and_changes_no_bit(family):
    """The reference's enable_cache (position_encoder.py:152-227 GridCache; run_video.py:144): with it the resized position embedding / BEiT's
    relative-position tables / SwinV2's position-bias tables and the zero pads are computed by the first forward of a (workspace, shape) and reused.
    Same bits as the uncached model on repeated calls, across a shape change and back, after an in-place weight edit (re-finalise), with the
    two-stream batch split (batch 8) and through stage-level calls in between (they drop the cached state)."""
    import muggled_dpt_amd as m
    from muggled_dpt_amd.synthetic import make_synthetic_beit_state_dict, make_synthetic_original_state_dict, make_synthetic_swinv2_state_dict
    if family == "beit":
        make, osd, sizes = m.make_beit_dpt_from_midas_v31_state_dict, make_synthetic_beit_state_dict("beit_tiny", 0), [(64, 96), (96, 64)]
    elif family == "swinv2":
        make, osd, sizes = m.make_swinv2_dpt_from_midas_v31_state_dict, make_synthetic_swinv2_state_dict("swin2_tiny", 0), [(128, 192), (256, 128)]
    else:
        make, osd, sizes = m.make_depthanythingv2_dpt_from_original_state_dict, make_synthetic_original_state_dict("tiny", 0), [(56, 84), (112, 56)]
    cfg_c, cached = make(osd, True)
    _, plain = make(osd, False)
    assert cfg_c["enable_cache"] is True
    for dtype in (torch.bfloat16, torch.float32):
        cached, plain = cached.to("cuda", dtype), plain.to("cuda", dtype)
        from muggled_dpt_amd import native
        lib = native.load()
        xs = {hw: torch.randn(2, 3, *hw, generator=torch.Generator().manual_seed(hw[0])).to("cuda", dtype) for hw in sizes}
        x8 = torch.randn(8, 3, *sizes[0], generator=torch.Generator().manual_seed(5)).to("cuda", dtype)
        want = {hw: plain(x) for hw, x in xs.items()}
        want8 = plain(x8)
        for hw in (sizes[0], sizes[0], sizes[1], sizes[0], sizes[1], sizes[1]):
            assert torch.equal(cached(xs[hw]), want[hw]), f"{family} {dtype} {hw}"
        for _ in range(3):
            assert torch.equal(cached(x8), want8)  # two half-batch plans, two cached slots
        tokens, grid = cached.patch_embed(xs[sizes[1]])  # a stage-level call writes the constant regions for another grid
        assert torch.equal(cached(xs[sizes[0]]), want[sizes[0]])
        with torch.no_grad():
            for mdl in (cached, plain):
                next(p for n, p in mdl.named_parameters() if "posenc" in n or "relpos" in n or "bias_mlp" in n).mul_(1.5)
        y = plain(xs[sizes[0]])
        assert not torch.equal(y, want[sizes[0]])
        assert torch.equal(cached(xs[sizes[0]]), y) and torch.equal(cached(xs[sizes[0]]), y)
        with torch.no_grad():
            for mdl in (cached, plain):
                next(p for n, p in mdl.named_parameters() if "posenc" in n or "relpos" in n or "bias_mlp" in n).div_(1.5)

"""SURVEY §8(f) row 1 as written: prepare_image fused into the patch embedding's im2col kernel (mdpt_forward_bgr). DPTModel.inference
(reference dpt_model.py:87-109 = patch_embed.py:103-145 prepare_image_bgr, then forward) runs it; its depth map has to equal the two-step
route - mdpt_prepare_image writing the model-dtype tensor, mdpt_forward reading it - bit for bit, for every family, dtype and sizing rule."""
import numpy as np
import pytest
import torch

from tests.test_gpu_c_host import _family_model

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.view(torch.int16) if t.dtype != torch.float32 else t


@pytest.mark.parametrize("family,dtype,precision", [("v2", torch.float32, None), ("v2", torch.bfloat16, None), ("v2", torch.float32, "mixed"), ("v1", torch.float16, None),
                                                    ("beit", torch.bfloat16, None), ("beit", torch.float32, "bf16x3"), ("swinv2", torch.float32, "mixed"), ("swinv2", torch.bfloat16, None)])
def test_inference_equals_prepare_image_then_forward_bit_for_bit(family, dtype, precision):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    model, unit = _family_model(family)
    model = model.to("cuda", dtype)
    if precision:
        model.set_precision(precision)
    rng = np.random.default_rng(3)
    side = 4 * unit if family != "swinv2" else 128
    for (ih, iw), max_side, square in (((333, 517), side, True), ((333, 517), side + unit, False), ((61, 64), side, True), ((1200, 900), side, False)):
        img = rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8)
        y = model.inference(img, max_side, square)
        with torch.inference_mode():
            x = model.prepare_image_bgr(img, max_side, square)
            y2 = model(x)
        assert y.shape == y2.shape == (1, x.shape[2], x.shape[3]) and y.dtype == y2.dtype == dtype
        assert torch.equal(_bits(y), _bits(y2)), f"{family} {dtype} {precision} {ih}x{iw}: fused and two-step inference differ"
        assert float(y.float().abs().max()) > 0


def test_inference_with_listening_hooks_takes_the_stage_route():
    """A forward hook on an attention softmax module makes forward() go stage by stage (so the hook sees its tensor); inference() has to do the same."""
    from torch import nn
    from muggled_dpt_amd import make_depthanythingv2_dpt_from_original_state_dict
    from muggled_dpt_amd.synthetic import make_synthetic_original_state_dict
    _, model = make_depthanythingv2_dpt_from_original_state_dict(make_synthetic_original_state_dict("tiny", 0), enable_optimizations=False)
    model = model.to("cuda", torch.float32)
    img = np.random.default_rng(4).integers(0, 256, (100, 140, 3), dtype=np.uint8)
    y = model.inference(img, 112, True)
    seen = []
    last = [m for m in model.modules() if isinstance(m, nn.Softmax)][-1]
    hd = last.register_forward_hook(lambda m, a, out: seen.append(tuple(out.shape)))
    try:
        y2 = model.inference(img, 112, True)
    finally:
        hd.remove()
    assert seen, "the hook did not fire"
    assert torch.equal(y, y2)


def test_forward_bgr_rejects_bad_arguments():
    import ctypes
    from muggled_dpt_amd import native
    model, unit = _family_model("v2")
    model = model.to("cuda", torch.float32)
    eng = model._get_engine()
    lib = native.load()
    img = torch.zeros((32, 32, 3), dtype=torch.uint8, device="cuda")
    out = torch.empty((1, 2 * unit, 2 * unit), device="cuda")
    m3, s3 = (ctypes.c_float * 3)(0.5, 0.5, 0.5), (ctypes.c_float * 3)(0.5, 0.5, 0.5)
    ws_ptr, ws_bytes = eng.workspace(1, (2 * unit, 2 * unit))
    args = lambda **kw: [eng.handle, kw.get("img", img.data_ptr()), 32, 32, kw.get("dt", native.dtype_code(torch.float32)), 2 * unit, kw.get("w", 2 * unit), m3, s3,
                         kw.get("interp", native.INTERP_BILINEAR), out.data_ptr(), native.dtype_code(torch.float32), ws_ptr, ws_bytes, None]
    assert lib.mdpt_forward_bgr(*args()) == 0
    assert lib.mdpt_forward_bgr(*args(img=None)) == -1
    assert lib.mdpt_forward_bgr(*args(dt=77)) == -1
    assert lib.mdpt_forward_bgr(*args(interp=5)) == -6
    assert lib.mdpt_forward_bgr(*args(w=2 * unit + unit // 2)) == native.E_GRID  # odd patch grid: the reference fails in fusion (RuntimeError), the library before launching
    torch.cuda.synchronize()

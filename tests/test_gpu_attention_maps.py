"""enable_optimizations=False: forward hooks on the per-block nn.Softmax modules receive the [B, heads, N, N] attention weights
(how the reference's attention-map tooling captures them: demo_helpers/model_capture.py:54-59 on components/transformer_block.py:101).
The weights come from mdpt_encoder_probe's dump kernel; they are checked against the oracle's softmax output. `pytest -m gpu`."""
import pytest
import torch
import torch.nn as nn

from tests.helpers import rel_err, seeded_input, synthetic_model

pytestmark = pytest.mark.gpu

ABS_TOL_X3 = 1e-4  # softmax weights are in [0, 1]: absolute error is the meaningful figure
ABS_TOL_BF16 = 2e-2
MODES = [(torch.float32, ABS_TOL_X3, 1e-4), (torch.bfloat16, ABS_TOL_BF16, 3e-2)]  # rtol: depth of the toy models (bf16 measured 0.9e-2 ... 2.0e-2)


def _oracle():
    from oracle import dpt_oracle
    return dpt_oracle


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from muggled_dpt_amd import native
    native.load()


def _hook_all_softmax(model):
    captured, handles = [], []
    for m in model.modules():
        if isinstance(m, nn.Softmax):
            handles.append(m.register_forward_hook(lambda mod, args, out: captured.append(out)))
    return captured, handles


def _build(family, name, dtype, enable_optimizations):
    import muggled_dpt_amd as mda
    from muggled_dpt_amd.state_dict_conversion import flatten_components
    if family == "beit":
        from muggled_dpt_amd import state_dict_conversion_beit as conv
        from muggled_dpt_amd.synthetic import make_synthetic_beit_state_dict
        osd = make_synthetic_beit_state_dict(name, 3)
        cfg, model = mda.make_beit_dpt_from_midas_v31_state_dict(osd, enable_optimizations=enable_optimizations)
        w = flatten_components(conv.convert_state_dict_keys(cfg, osd))
    else:
        osd, cfg, w = synthetic_model(name, 0)
        cfg, model = mda.make_depthanythingv2_dpt_from_original_state_dict(osd, enable_optimizations=enable_optimizations)
    return model.to("cuda", dtype), cfg, w


@pytest.mark.parametrize("dtype,atol,rtol", MODES)
@pytest.mark.parametrize("family,name,hw", [("v2", "tiny", (56, 84)), ("beit", "beit_tiny", (64, 96))])
def test_softmax_hooks_receive_attention_weights(family, name, hw, dtype, atol, rtol):
    orc = _oracle()
    model, cfg, w = _build(family, name, dtype, enable_optimizations=False)
    x = seeded_input((2, 3, *hw), seed=11)
    captured, handles = _hook_all_softmax(model)
    assert len(handles) == cfg["num_blocks"]
    y = model(x.to("cuda", dtype))
    assert len(captured) == cfg["num_blocks"]

    tokens, grid = orc.patch_embed(w, x)
    ref = []
    (orc.beit_image_encoder if family == "beit" else orc.image_encoder)(w, cfg, tokens, grid, capture=ref)
    n = grid[0] * grid[1] + 1
    for blk, (got, want) in enumerate(zip(captured, ref)):
        assert got.dtype == dtype and tuple(got.shape) == (2, cfg["num_heads"], n, n)
        g = got.float().cpu()
        assert float((g - want).abs().max()) <= atol, f"block {blk}"
        assert float((g.sum(-1) - 1).abs().max()) <= (1e-5 if dtype == torch.float32 else 2e-2)
    # the prediction itself is the usual one, with or without listeners
    assert rel_err(y.float().cpu(), orc.forward(w, cfg, x)) <= rtol
    for h in handles:
        h.remove()
    y2 = model(x.to("cuda", dtype))  # no hooks left -> fused single-call path again
    assert len(captured) == cfg["num_blocks"]
    assert rel_err(y2.float().cpu(), y.float().cpu()) <= 1e-6 if dtype == torch.float32 else True


def test_single_block_hook_and_optimised_model_has_no_softmax_modules():
    orc = _oracle()
    model, cfg, w = _build("v2", "tiny", torch.float32, enable_optimizations=False)
    fast, _, _ = _build("v2", "tiny", torch.float32, enable_optimizations=True)
    assert not any(isinstance(m, nn.Softmax) for m in fast.modules())
    x = seeded_input((1, 3, 56, 56), seed=4)
    last = [m for m in model.modules() if isinstance(m, nn.Softmax)][-1]
    got = []
    last.register_forward_hook(lambda mod, args, out: got.append(out))
    model(x.cuda())
    assert len(got) == 1
    ref = []
    tokens, grid = orc.patch_embed(w, x)
    orc.image_encoder(w, cfg, tokens, grid, capture=ref)
    assert float((got[0].cpu() - ref[-1]).abs().max()) <= ABS_TOL_X3
    # stage-level call with hooks attached works as well (simple_examples/internal_features.py usage)
    tk, hw = model.patch_embed(x.cuda())
    model.imgencoder(tk, hw)
    assert len(got) == 2 and torch.equal(got[0], got[1])


# bf16: the cosine-attention logits are scaled by exp(logit_scale) ~ 10 before the softmax, which amplifies the bf16 rounding of the
# normalised q / k: measured max abs error 2.0e-2 on weights in [0, 1] (fp32-class mode: 3e-5)
@pytest.mark.parametrize("dtype,atol", [(torch.float32, ABS_TOL_X3), (torch.bfloat16, 4e-2)])
def test_swinv2_window_softmax_hooks_receive_attention_weights(dtype, atol):
    """SwinV2: every window-attention block owns a hookable nn.Softmax in the reference (v31_swinv2/components/windowed_attention.py:60-61,
    :119 - no enable_optimizations switch), output [B * windows, heads_of_the_stage, Nw, Nw]. Here the hook fires with the dump of
    mdpt_encoder_probe (cosine attention * logit scale + position-bias LUT + shift mask, softmax): shifted and unshifted blocks of all
    four stages of swin2_tiny on a grid where stage 0/1 shift (16x24 patches), against the oracle's capture."""
    import muggled_dpt_amd as mda
    from muggled_dpt_amd import state_dict_conversion_swinv2 as conv
    from muggled_dpt_amd.state_dict_conversion import flatten_components
    from muggled_dpt_amd.synthetic import make_synthetic_swinv2_state_dict
    orc = _oracle()
    osd = make_synthetic_swinv2_state_dict("swin2_tiny", 3)
    cfg, model = mda.make_swinv2_dpt_from_midas_v31_state_dict(osd)
    w = flatten_components(conv.convert_state_dict_keys(cfg, osd))
    model = model.to("cuda", dtype)
    p = cfg["patch_size_px"]
    x = seeded_input((2, 3, 16 * p, 24 * p), seed=13)
    captured, handles = _hook_all_softmax(model)
    nblk = sum(cfg["layers_per_stage"])
    assert len(handles) == nblk
    y = model(x.to("cuda", dtype))
    assert len(captured) == nblk
    tokens, grid = orc.patch_embed(w, x)
    ref = []
    orc.swin_image_encoder(w, cfg, tokens, grid, capture=ref)
    assert len(ref) == nblk
    for blk, (got, want) in enumerate(zip(captured, ref)):
        assert got.dtype == dtype and tuple(got.shape) == tuple(want.shape), (blk, tuple(got.shape), tuple(want.shape))
        g = got.float().cpu()
        assert float((g - want).abs().max()) <= atol, f"block {blk}"
        assert float((g.sum(-1) - 1).abs().max()) <= (1e-5 if dtype == torch.float32 else 2e-2)
    assert rel_err(y.float().cpu(), orc.forward(w, cfg, x)) <= (1e-4 if dtype == torch.float32 else 3e-2)
    for h in handles:
        h.remove()
    y2 = model(x.to("cuda", dtype))  # no hooks left -> fused single-call path (bf16: the stage-by-stage path rounds the stage outputs to the model dtype)
    assert torch.equal(y2, y) if dtype == torch.float32 else rel_err(y2.float().cpu(), y.float().cpu()) <= 3e-2
    # one listener on a shifted block of stage 1 only; stage-level call (simple_examples/internal_features.py usage)
    got = []
    blk_mod = model.imgencoder.stages[1].blocks[1].attn.softmax
    blk_mod.register_forward_hook(lambda mod, args, out: got.append(out))
    tk, hw = model.patch_embed(x.to("cuda", dtype))
    model.imgencoder(tk, hw)
    idx = cfg["layers_per_stage"][0] + 1
    assert len(got) == 1 and float((got[0].float().cpu() - ref[idx]).abs().max()) <= atol

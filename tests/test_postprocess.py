"""Depth post-processing: oracle vs fixtures made with the reference's helpers (CPU), HIP kernels vs both (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import dpt_oracle


def test_oracle_matches_reference_fixtures(golden_dir):
    g = np.load(os.path.join(golden_dir, "postprocess.npz"))
    depth = torch.from_numpy(g["depth"])
    for tag, wh in (("up", (200, 130)), ("down", (40, 30)), ("same", (84, 56))):
        s = dpt_oracle.scale_prediction(depth, wh)
        assert tuple(s.shape) == (2, wh[1], wh[0]) and torch.equal(s, torch.from_numpy(g[f"scaled_{tag}"]))
        assert torch.equal(dpt_oracle.convert_to_uint8(s), torch.from_numpy(g[f"scaled_{tag}_u8"]))
    assert torch.equal(dpt_oracle.normalize_01(depth[:1]), torch.from_numpy(g["norm01"]))
    packed = dpt_oracle.pack_depth_u24(depth[:1])
    assert np.array_equal(packed.numpy(), g["packed_u24"])
    # the three bytes reassemble to round(16777215 * norm), alpha untouched
    q = packed[..., 2].int() * 65536 + packed[..., 1].int() * 256 + packed[..., 0].int()
    assert torch.equal(q, torch.round(16777215 * torch.from_numpy(g["norm01"])).int().squeeze()) and int(packed[..., 3].max()) == 0
    assert int(dpt_oracle.pack_depth_u24(depth[:1], lossy=True)[..., :2].max()) == 0


def test_host_tensors_fail_loudly():
    from muggled_dpt_amd import postprocess
    with pytest.raises(RuntimeError):
        postprocess.normalize_01(torch.zeros(1, 4, 4))
    with pytest.raises(RuntimeError):
        postprocess.convert_to_uint8(np.zeros((4, 4), dtype=np.float32))


@pytest.mark.gpu
def test_hip_postprocess_vs_reference_fixtures(golden_dir):
    from muggled_dpt_amd import postprocess
    g = np.load(os.path.join(golden_dir, "postprocess.npz"))
    depth = torch.from_numpy(g["depth"]).cuda()
    for tag, wh in (("up", (200, 130)), ("down", (40, 30)), ("same", (84, 56))):
        s = postprocess.scale_prediction(depth, wh)
        ref = torch.from_numpy(g[f"scaled_{tag}"])
        assert s.is_cuda and tuple(s.shape) == tuple(ref.shape)
        assert float((s.cpu() - ref).abs().max()) <= 2e-6   # fp32 lerp, summation-order noise only
        # uint8 conversion of the REFERENCE's scaled map: integer result must match exactly (same fp32 expression)
        u8 = postprocess.convert_to_uint8(ref.cuda())
        assert u8.dtype == torch.uint8 and torch.equal(u8.cpu(), torch.from_numpy(g[f"scaled_{tag}_u8"]))
        # fused resize + min/max + convert: at most one grey level off where the 2e-6 resize noise crosses a truncation boundary
        fused = postprocess.scale_and_convert_to_uint8(depth, wh).cpu().int()
        diff = (fused - torch.from_numpy(g[f"scaled_{tag}_u8"]).int()).abs()
        assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 2e-3
    n01 = postprocess.normalize_01(depth[:1])
    assert torch.equal(n01.cpu(), torch.from_numpy(g["norm01"]))
    packed = postprocess.pack_depth_u24(depth[:1])
    assert tuple(packed.shape) == (56, 84, 4) and np.array_equal(packed.cpu().numpy(), g["packed_u24"])
    lossy = postprocess.pack_depth_u24(depth[:1], lossy=True).cpu().numpy()
    assert np.array_equal(lossy[..., 2], g["packed_u24"][..., 2]) and lossy[..., :2].max() == 0
    metric = postprocess.pack_depth_u24(torch.from_numpy(g["norm01"]).cuda(), is_metric=True)
    assert np.array_equal(metric.cpu().numpy(), g["packed_u24"])


@pytest.mark.gpu
def test_hip_postprocess_on_model_output_and_negative_values():
    """bf16 model output dtype round trip, negative inputs (ordered-int min/max), full-size map properties."""
    from muggled_dpt_amd import postprocess
    x = torch.randn(3, 504, 504, generator=torch.Generator().manual_seed(3)) * 4 - 1
    n = postprocess.normalize_01(x.cuda())
    assert float(n.min()) == 0.0 and float(n.max()) == 1.0
    assert torch.equal(n.cpu(), dpt_oracle.normalize_01(x))
    u8 = postprocess.convert_to_uint8(x.cuda())
    assert torch.equal(u8.cpu(), dpt_oracle.convert_to_uint8(x)) and int(u8.max()) == 255 and int(u8.min()) == 0
    xb = x.cuda().to(torch.bfloat16)
    assert postprocess.normalize_01(xb).dtype == torch.bfloat16 and postprocess.scale_prediction(xb, (100, 50)).shape == (3, 50, 100)


@pytest.mark.gpu
def test_hip_postprocess_on_random_shapes_vs_oracle():
    """Random map sizes / batch sizes / target sizes (strong up- and down-scales, 1-pixel targets excluded as in the demos):
    resize within fp32 noise, integer conversions exact given identical inputs."""
    from muggled_dpt_amd import postprocess
    rng = np.random.default_rng(5)
    for k in range(16):
        b, h, w = int(rng.integers(1, 4)), int(rng.integers(2, 300)), int(rng.integers(2, 300))
        tw, th = int(rng.integers(2, 700)), int(rng.integers(2, 700))
        x = torch.from_numpy(rng.standard_normal((b, h, w), dtype=np.float32) * float(rng.uniform(0.1, 50)) + float(rng.uniform(-5, 5)))
        s = postprocess.scale_prediction(x.cuda(), (tw, th))
        ref = dpt_oracle.scale_prediction(x, (tw, th))
        assert tuple(s.shape) == tuple(ref.shape) == (b, th, tw)
        scale = float(x.abs().max())
        assert float((s.cpu() - ref).abs().max()) <= 4e-6 * scale, (b, h, w, tw, th)
        assert torch.equal(postprocess.normalize_01(x.cuda()).cpu(), dpt_oracle.normalize_01(x))
        assert torch.equal(postprocess.convert_to_uint8(x.cuda()).cpu(), dpt_oracle.convert_to_uint8(x))
        assert torch.equal(postprocess.pack_depth_u24(x[:1].cuda()).cpu(), dpt_oracle.pack_depth_u24(x[:1]))


@pytest.mark.gpu
def test_nan_and_constant_maps_behave_like_torch_min_max():
    """torch's .min() / .max() propagate NaN (fminf / fmaxf would drop it), and a constant map has max == min: the fp32 normalisation
    yields NaN like the reference's (x - min) / (max - min); the integer conversions write 0 instead of converting a NaN."""
    from muggled_dpt_amd.postprocess import convert_to_uint8, normalize_01
    x = torch.rand(2, 40, 56, device="cuda")
    x[1, 3, 5] = float("nan")
    y = normalize_01(x)
    assert bool(torch.isnan(y).all()), "a NaN anywhere makes min and max NaN (torch semantics)"
    u = convert_to_uint8(x)  # NaN min / max -> every normalised value is NaN -> written as 0, never converted
    assert u.dtype == torch.uint8 and int(u.max()) == 0
    c = torch.full((1, 8, 8), 0.25, device="cuda")
    assert bool(torch.isnan(normalize_01(c)).all())
    assert int(convert_to_uint8(torch.zeros(1, 8, 8, device="cuda")).max()) == 0

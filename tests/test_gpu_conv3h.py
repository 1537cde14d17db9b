"""Halo-staged 3x3 conv kernel (csrc/conv3h.hip) vs a CPU fp32 conv on the same bf16-rounded operands, and bit-for-bit vs the
implicit-GEMM kernels of gemm.hip in every tile variant (a tile-rule change between batch sizes must not change a bit).
Reference ops: nn.Conv2d(C, C, 3, padding=1) of ResidualConv2D / the fusion blocks, + skip, + x2 bilinear (align_corners=True) of the
previous level (v2_depthanything/fusion_model.py:148-154,178-182,210-220)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _pack(w):  # [256][Cin][3][3] fp32 -> [256][9 Cin], k = (cb * 9 + ky * 3 + kx) * 64 + c  (MDPT_PACK_CONV3)
    n, cin = w.shape[:2]
    return w.view(n, cin // 64, 64, 3, 3).permute(0, 1, 3, 4, 2).reshape(n, 9 * cin).contiguous()


def _run(lib, native, path, tile, x, wp, bias, skip, up, want_f32, relu, dbg=None, x_lo=None, wp_lo=None, want_bf=True):
    B, H, W, Cin = x.shape
    cout = wp.shape[0]
    out_bf = torch.full((B, H, W, cout), float("nan"), device="cuda", dtype=torch.bfloat16) if want_bf else None
    out_lo = torch.full((B, H, W, cout), float("nan"), device="cuda", dtype=torch.bfloat16) if (want_bf and x_lo is not None) else None
    out_f32 = torch.full((B, H, W, cout), float("nan"), device="cuda", dtype=torch.float32) if want_f32 else None
    stream = torch.cuda.current_stream().cuda_stream
    native.check(lib, lib.mdpt_debug_conv3(x.data_ptr(), wp.data_ptr(), bias.data_ptr() if bias is not None else None,
                                           skip.data_ptr() if skip is not None else None, up.data_ptr() if up is not None else None,
                                           up.shape[1] if up is not None else 0, up.shape[2] if up is not None else 0,
                                           out_f32.data_ptr() if want_f32 else None, out_bf.data_ptr() if want_bf else None, int(relu), B, H, W, Cin, cout, path, tile, 1, stream,
                                           dbg.data_ptr() if dbg is not None else None, x_lo.data_ptr() if x_lo is not None else None,
                                           wp_lo.data_ptr() if wp_lo is not None else None, out_lo.data_ptr() if out_lo is not None else None))
    torch.cuda.synchronize()
    if x_lo is not None:
        return out_f32, (out_bf, out_lo)
    return out_f32, out_bf


# (skip, fp32 out, relu on bf16, upsample add): the four combinations the decoder uses
VARIANTS = [(False, False, True, False), (True, False, False, False), (False, True, True, False), (True, True, True, True)]
SHAPES = [(2, 32, 32, 128), (1, 48, 40, 256), (3, 16, 16, 128), (1, 18, 34, 128)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("variant", VARIANTS)
def test_conv3h_vs_cpu_conv_and_bitwise_vs_implicit_gemm(shape, variant):
    from muggled_dpt_amd import native
    lib = native.load()
    B, H, W, Cin = shape
    has_skip, want_f32, relu, has_up = variant
    if has_up and (H % 2 or W % 2):
        pytest.skip("the x2 prior needs even sizes")
    g = torch.Generator().manual_seed(1000 * H + W + Cin + 7 * int(has_skip) + 13 * int(has_up))
    x = (torch.randn(B, H, W, Cin, generator=g)).to(torch.bfloat16)
    w = (torch.randn(256, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(torch.bfloat16)
    bias = torch.randn(256, generator=g)
    skip = torch.randn(B, H, W, 256, generator=g) if has_skip else None
    up = torch.randn(B, H // 2, W // 2, 256, generator=g) if has_up else None
    # CPU reference in fp64 on the same bf16 operands
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), padding=1)
    if has_skip:
        ref = ref + skip.double().permute(0, 3, 1, 2)
    if has_up:
        ref = ref + F.interpolate(up.double().permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=True)
    ref = ref.permute(0, 2, 3, 1)
    ref_bf = ref.clamp_min(0) if relu else ref

    xd, wp, bd = x.cuda(), _pack(w.float()).to(torch.bfloat16).cuda(), bias.cuda()
    sd = skip.cuda() if has_skip else None
    ud = up.cuda() if has_up else None
    o32, obf = _run(lib, native, 1, 0, xd, wp, bd, sd, ud, want_f32, relu)
    scale = float(ref.abs().max())
    if want_f32:
        err = float((o32.double().cpu() - ref).abs().max()) / scale
        assert err < 2e-5, f"fp32 output: rel err {err:.3e}"
    err_bf = float((obf.double().cpu() - ref_bf).abs().max()) / scale
    assert err_bf < 6e-3, f"bf16 output: rel err {err_bf:.3e}"  # bf16 rounding of the result (2^-9 of the largest value)
    # bit-for-bit vs the implicit-GEMM kernels: 64x64 and 128x128 lockstep tiles, the lockstep 256x256 tile, the automatic rule
    for tile in (6, 1, 2, 0):
        r32, rbf = _run(lib, native, 0, tile, xd, wp, bd, sd, ud, want_f32, relu)
        assert torch.equal(rbf.view(torch.int16), obf.view(torch.int16)), f"bf16 output differs from implicit GEMM tile {tile}"
        if want_f32:
            assert torch.equal(r32.view(torch.int32), o32.view(torch.int32)), f"fp32 output differs from implicit GEMM tile {tile}"


def test_conv3h_is_deterministic_and_batch_independent():
    from muggled_dpt_amd import native
    lib = native.load()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 32, 48, 128, generator=g).to(torch.bfloat16).cuda()
    wp = _pack(torch.randn(256, 128, 3, 3, generator=g) / 30.0).to(torch.bfloat16).cuda()
    bias = torch.randn(256, generator=g).cuda()
    skip = torch.randn(6, 32, 48, 256, generator=g).cuda()
    _, full = _run(lib, native, 1, 0, x, wp, bias, skip, None, False, False)
    _, again = _run(lib, native, 1, 0, x, wp, bias, skip, None, False, False)
    assert torch.equal(full.view(torch.int16), again.view(torch.int16))
    _, one = _run(lib, native, 1, 0, x[4:5].contiguous(), wp, bias, skip[4:5].contiguous(), None, False, False)
    assert torch.equal(full[4:5].view(torch.int16), one.view(torch.int16))


@pytest.mark.parametrize("shape", [(2, 32, 32, 256), (1, 48, 40, 128), (1, 18, 34, 128)])
def test_conv3h_128_output_channels_head_conv1_form(shape):
    """The head's first conv (256 -> 128 channels, bias, no activation: head_model.py:74-76) in the halo-staged kernel's 128-channel
    form (two MFMA phases per K tile, four-deep weight ring) vs a CPU fp64 conv and bit-for-bit vs the implicit-GEMM kernels."""
    from muggled_dpt_amd import native
    lib = native.load()
    B, H, W, Cin = shape
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16)
    w = (torch.randn(128, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(torch.bfloat16)
    bias = torch.randn(128, generator=g)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    xd, wp, bd = x.cuda(), _pack(w.float()).to(torch.bfloat16).cuda(), bias.cuda()
    _, obf = _run(lib, native, 1, 0, xd, wp, bd, None, None, False, False)
    err = float((obf.double().cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err < 6e-3, f"rel err {err:.3e}"
    for tile in (6, 1, 0):
        _, rbf = _run(lib, native, 0, tile, xd, wp, bd, None, None, False, False)
        assert torch.equal(rbf.view(torch.int16), obf.view(torch.int16)), f"differs from implicit GEMM tile {tile}"


def _split(t):  # fp32 -> (hi, lo) bf16 planes, t ~= hi + lo
    hi = t.to(torch.bfloat16)
    return hi, (t - hi.float()).to(torch.bfloat16)


@pytest.mark.parametrize("variant", VARIANTS)
def test_conv3h_bf16x3_mode_fp32_class_accuracy_and_bitwise_vs_implicit_gemm(variant):
    """bf16x3 (hi + lo operand planes, three MFMA passes): the halo-staged kernel against an fp64 conv on the fp32 operands (fp32-class
    error) and bit-for-bit against the implicit-GEMM kernels (same pass order A_lo W_hi, A_hi W_lo, A_hi W_hi)."""
    from muggled_dpt_amd import native
    lib = native.load()
    B, H, W, Cin = 2, 32, 48, 128
    has_skip, want_f32, relu, has_up = variant
    g = torch.Generator().manual_seed(77 + 3 * int(has_skip) + 5 * int(has_up))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(256, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    bias = torch.randn(256, generator=g)
    skip = torch.randn(B, H, W, 256, generator=g) if has_skip else None
    up = torch.randn(B, H // 2, W // 2, 256, generator=g) if has_up else None
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), padding=1)
    if has_up:
        ref = ref + F.interpolate(up.double().permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=True)
    if has_skip:
        ref = ref + skip.double().permute(0, 3, 1, 2)
    ref = ref.permute(0, 2, 3, 1)
    xh, xl = _split(x)
    wh, wl = _split(_pack(w))
    args = dict(x_lo=xl.cuda(), wp_lo=wl.cuda())
    sd = skip.cuda() if has_skip else None
    ud = up.cuda() if has_up else None
    o32, (ohi, olo) = _run(lib, native, 1, 0, xh.cuda(), wh.cuda(), bias.cuda(), sd, ud, want_f32, relu, **args)
    got = ohi.double().cpu() + olo.double().cpu()
    want = ref.clamp_min(0) if relu else ref
    err = float((got - want).abs().max()) / float(ref.abs().max())
    assert err < 3e-5, f"hi + lo planes: rel err {err:.3e}"
    if want_f32:
        e32 = float((o32.double().cpu() - ref).abs().max()) / float(ref.abs().max())
        assert e32 < 1e-5, f"fp32 map: rel err {e32:.3e}"
    for tile in (6, 1, 0):
        r32, (rhi, rlo) = _run(lib, native, 0, tile, xh.cuda(), wh.cuda(), bias.cuda(), sd, ud, want_f32, relu, **args)
        assert torch.equal(rhi.view(torch.int16), ohi.view(torch.int16)) and torch.equal(rlo.view(torch.int16), olo.view(torch.int16)), f"tile {tile}"
        if want_f32:
            assert torch.equal(r32.view(torch.int32), o32.view(torch.int32)), f"fp32 map differs, tile {tile}"


@pytest.mark.parametrize("variant", VARIANTS)
def test_conv3h_two_pass_activation_split_form_and_bitwise_vs_implicit_gemm(variant):
    """Two passes (GemmParams::npass == 2, mdpt_set_class_passes(..., 2)): the input keeps its hi + lo planes, the weights ONE rounded plane,
    A_lo W_hi + A_hi W_hi. Against an fp64 conv of the fp32 input with the ROUNDED weights it is fp32-class (only the weights' rounding is left of
    the single-pass error); bit for bit against the implicit-GEMM kernels (same pass order)."""
    from muggled_dpt_amd import native
    lib = native.load()
    B, H, W, Cin = 2, 32, 48, 128
    has_skip, want_f32, relu, has_up = variant
    g = torch.Generator().manual_seed(177 + 3 * int(has_skip) + 5 * int(has_up))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = (torch.randn(256, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(torch.bfloat16).float()  # one plane: already representable
    bias = torch.randn(256, generator=g)
    skip = torch.randn(B, H, W, 256, generator=g) if has_skip else None
    up = torch.randn(B, H // 2, W // 2, 256, generator=g) if has_up else None
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), padding=1)
    if has_up:
        ref = ref + F.interpolate(up.double().permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=True)
    if has_skip:
        ref = ref + skip.double().permute(0, 3, 1, 2)
    ref = ref.permute(0, 2, 3, 1)
    xh, xl = _split(x)
    wh = _pack(w).to(torch.bfloat16)
    args = dict(x_lo=xl.cuda(), wp_lo=None)
    sd = skip.cuda() if has_skip else None
    ud = up.cuda() if has_up else None
    o32, (ohi, olo) = _run(lib, native, 1, 0, xh.cuda(), wh.cuda(), bias.cuda(), sd, ud, want_f32, relu, **args)
    got = ohi.double().cpu() + olo.double().cpu()
    want = ref.clamp_min(0) if relu else ref
    err = float((got - want).abs().max()) / float(ref.abs().max())
    assert err < 3e-5, f"hi + lo planes: rel err {err:.3e}"
    if want_f32:
        e32 = float((o32.double().cpu() - ref).abs().max()) / float(ref.abs().max())
        assert e32 < 1e-5, f"fp32 map: rel err {e32:.3e}"
    for tile in (6, 1, 0):
        r32, (rhi, rlo) = _run(lib, native, 0, tile, xh.cuda(), wh.cuda(), bias.cuda(), sd, ud, want_f32, relu, **args)
        assert torch.equal(rhi.view(torch.int16), ohi.view(torch.int16)) and torch.equal(rlo.view(torch.int16), olo.view(torch.int16)), f"tile {tile}"
        if want_f32:
            assert torch.equal(r32.view(torch.int32), o32.view(torch.int32)), f"fp32 map differs, tile {tile}"


def test_conv3h_128_channels_fp32_map_bf16x3_head_form():
    """The bf16x3 head keeps conv 1's output as an fp32 map (its bilinear upsample reads fp32): 128 channels, fp32 store only."""
    from muggled_dpt_amd import native
    lib = native.load()
    B, H, W, Cin = 1, 32, 32, 256
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(128, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)
    bias = torch.randn(128, generator=g)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    xh, xl = _split(x)
    wh, wl = _split(_pack(w))
    o32, _ = _run(lib, native, 1, 0, xh.cuda(), wh.cuda(), bias.cuda(), None, None, True, False, x_lo=xl.cuda(), wp_lo=wl.cuda(), want_bf=False)
    err = float((o32.double().cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err < 1e-5, f"rel err {err:.3e}"
    r32, _ = _run(lib, native, 0, 6, xh.cuda(), wh.cuda(), bias.cuda(), None, None, True, False, x_lo=xl.cuda(), wp_lo=wl.cuda(), want_bf=False)
    assert torch.equal(r32.view(torch.int32), o32.view(torch.int32))


@pytest.mark.parametrize("shape", [(2, 32, 32, 256), (1, 48, 64, 128), (1, 34, 18, 128)])
def test_conv3h_with_the_x2_upsample_folded_into_its_halo_interpolation(shape):
    """Head conv 1 on the x2 bilinear upsample (align_corners=True) of the last fusion projection (fusion_model.py:182 -> head_model.py:74-76):
    the halo-staged kernel interpolates its 18x18 input patches in LDS from the half-resolution bf16 map (no upsampled map in memory).
    Checked against an fp64 upsample + conv on the bf16 source, and bit for bit against the stand-alone bf16 upsample kernel followed by
    the implicit-GEMM conv (the small-launch path of the same model)."""
    from muggled_dpt_amd import native
    lib = native.load()
    B, H, W, Cin = shape
    Hs, Ws = H // 2, W // 2
    g = torch.Generator().manual_seed(H * 31 + W)
    src = torch.randn(B, Hs, Ws, Cin, generator=g).to(torch.bfloat16)
    w = (torch.randn(128, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).to(torch.bfloat16)
    bias = torch.randn(128, generator=g)
    up = F.interpolate(src.double().permute(0, 3, 1, 2), size=(H, W), mode="bilinear", align_corners=True).to(torch.bfloat16).double()
    ref = F.conv2d(up, w.double(), bias.double(), padding=1).permute(0, 2, 3, 1)
    sd, wp, bd = src.cuda(), _pack(w.float()).to(torch.bfloat16).cuda(), bias.cuda()
    stream = torch.cuda.current_stream().cuda_stream

    def run(path, tile):
        out = torch.full((B, H, W, 128), float("nan"), device="cuda", dtype=torch.bfloat16)
        scratch = torch.full((B, H, W, Cin), float("nan"), device="cuda", dtype=torch.bfloat16)
        native.check(lib, lib.mdpt_debug_conv3(sd.data_ptr(), wp.data_ptr(), bd.data_ptr(), None, None, Hs, Ws, None, out.data_ptr(), 0, B, H, W, Cin, 128,
                                               path, tile, 1, stream, None, None, None, scratch.data_ptr()))
        torch.cuda.synchronize()
        return out, scratch

    got, _ = run(2, 0)
    err = float((got.double().cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err < 1.2e-2, f"rel err {err:.3e}"  # bf16 rounding of the interpolated input (may differ by one bf16 ulp from the fp64 interpolation) and of the output
    for tile in (6, 1, 0):
        want, scratch = run(3, tile)
        assert torch.equal(want.view(torch.int16), got.view(torch.int16)), f"differs from upsample + implicit GEMM (tile {tile})"
    # the stand-alone upsample itself vs torch (one bf16 ulp: fp32 vs fp64 interpolation before the rounding)
    e_up = float((scratch.double().cpu() - up.permute(0, 2, 3, 1)).abs().max()) / float(up.abs().max())
    assert e_up < 8e-3, f"upsample rel err {e_up:.3e}"

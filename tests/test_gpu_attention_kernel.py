"""The fused attention kernel alone (csrc/attention.hip) against an fp64 softmax(q k^T) v on the same bf16 operands, including inputs that
FORCE the deferred-maximum rescale branch: the online softmax only moves a query's reference point when a tile's maximum exceeds it by
more than e^5.5, so bounded random scores never take that branch after the first tile - outlier keys late in the sequence do
(reference op: F.scaled_dot_product_attention, v2_depthanything/components/transformer_block.py:164)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(q, k, v, N):
    """q, k, v: [B, H, N, 64] fp32 (q already scaled by 1/8) -> kernel output [B, N, H*64] and the bf16-rounded operands"""
    from muggled_dpt_amd import native
    lib = native.load()
    B, H = q.shape[:2]
    npad, npadv = (N + 7) // 8 * 8, (N + 63) // 64 * 64
    qb = torch.zeros(B, H, npad, 64, dtype=torch.bfloat16); qb[:, :, :N] = q.to(torch.bfloat16)
    kb = torch.zeros(B, H, npad, 64, dtype=torch.bfloat16); kb[:, :, :N] = k.to(torch.bfloat16)
    vt = torch.zeros(B, H, 64, npadv, dtype=torch.bfloat16); vt[:, :, :, :N] = v.to(torch.bfloat16).transpose(2, 3)
    out = torch.zeros(B * npad, H * 64, dtype=torch.bfloat16, device="cuda")
    qd, kd, vd = qb.cuda(), kb.cuda(), vt.cuda()
    native.check(lib, lib.mdpt_debug_attention(qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), out.data_ptr(), B, H, N, npad, npadv, 1,
                                               torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    o = out.view(B, npad, H, 64)[:, :N].permute(0, 2, 1, 3).float().cpu()  # [B, H, N, 64]
    return o, qb[:, :, :N].double(), kb[:, :, :N].double(), vt[:, :, :, :N].transpose(2, 3).double()


def _ref(q, k, v):
    return torch.softmax(q @ k.transpose(2, 3), dim=-1) @ v


@pytest.mark.parametrize("N", [1297, 325, 64])
def test_attention_kernel_vs_fp64_softmax_on_random_and_outlier_keys(N):
    g = torch.Generator().manual_seed(N)
    B, H = 2, 16  # 2 x 16 x ceil(1304 / 256) >= 512 workgroups at N = 1297: the 64-queries-per-wave form; smaller N: the 128-query form
    q = torch.randn(B, H, N, 64, generator=g) * 0.125 * 2.0
    k = torch.randn(B, H, N, 64, generator=g)
    v = torch.randn(B, H, N, 64, generator=g)
    # outlier keys: late keys aligned with a few queries so that those queries' maxima jump by far more than e^5.5 in the LAST tiles,
    # others by a little less than the threshold (the reference point stays, p > 1)
    for (qi, ki, gain) in ((3, N - 2, 6.0), (N // 2, N - 40 if N > 64 else N - 3, 3.0), (N - 1, N - 1, 1.2), (17, max(N - 70, 1), 0.6)):
        k[:, :, ki] = q[:, :, qi] / q[:, :, qi].norm(dim=-1, keepdim=True) * 8.0 * gain
    out, qd, kd, vd = _run(q, k, v, N)
    ref = _ref(qd, kd, vd)
    err = float((out.double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 8e-3, f"N={N}: rel err {err:.3e}"  # bf16 rounding of P (2^-9 relative) and of the output
    # the rows that took the rescale branch late must be as good as the rest
    for qi in (3, N // 2, N - 1, 17):
        e = float((out[:, :, qi].double() - ref[:, :, qi]).abs().max()) / float(ref.abs().max())
        assert e < 8e-3, f"N={N}, query {qi}: rel err {e:.3e}"


def test_attention_kernel_rows_do_not_depend_on_their_wave_partners():
    """A query's bits must not depend on which other queries share its wave (the deferred-maximum decision is per lane)."""
    g = torch.Generator().manual_seed(11)
    N = 700
    q = torch.randn(1, 16, N, 64, generator=g) * 0.25
    k = torch.randn(1, 16, N, 64, generator=g)
    v = torch.randn(1, 16, N, 64, generator=g)
    out_a, *_ = _run(q, k, v, N)
    q2 = q.clone()
    q2[:, :, 1::2] *= 7.0  # every other query becomes a very different row (much larger maxima)
    out_b, *_ = _run(q2, k, v, N)
    assert torch.equal(out_a[:, :, 0::2], out_b[:, :, 0::2])


def test_attention_kernel_64_and_32_queries_per_wave_give_the_same_bits():
    """The launcher runs 256-query workgroups (64 queries per wave) when there are >= 512 of them AND the query count pads by <= 8 % to a
    multiple of 256, else 128-query workgroups - so the form depends on the batch size. N = 1024 fills: one image x 16 heads takes the
    narrow form (64 wide workgroups), eight images the wide one (512). Image 0 must come out identical, and both must match fp64."""
    g = torch.Generator().manual_seed(23)
    N = 1024
    q = torch.randn(8, 16, N, 64, generator=g) * 0.25
    k = torch.randn(8, 16, N, 64, generator=g)
    v = torch.randn(8, 16, N, 64, generator=g)
    k[:, :, 900] *= 6.0  # an outlier key late in the sequence: the deferred-maximum branch runs in both forms
    wide, qd, kd, vd = _run(q, k, v, N)
    narrow, *_ = _run(q[:1], k[:1], v[:1], N)
    assert torch.equal(wide[:1], narrow)
    ref = _ref(qd[:1], kd[:1], vd[:1])
    assert float((narrow.double() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())

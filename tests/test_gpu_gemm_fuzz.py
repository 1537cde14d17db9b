"""Randomised shapes through the bare GEMM entry point (mdpt_debug_gemm) for every main-loop variant and the epilogue forms it exposes:
fp32 strip, bf16 direct, bf16 + erf-GELU, in-place residual (read-modify-write epilogue and residual-initialised accumulators). Ragged M exercises the bounds-check-drop stores of the direct epilogues
(tail tiles), N / K the tile rules. Reference: fp32 matmul of the same bf16-rounded operands. `pytest -m gpu`."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from muggled_dpt_amd import native
    return native.load()


def _run(lib, a, w, mode, tile):
    from muggled_dpt_amd import native
    M, K = a.shape
    N = w.shape[0]
    stream = torch.cuda.current_stream().cuda_stream
    out32 = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.float32)      # 3 guard rows behind the matrix
    out16 = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
    flags = tile | (2 << 8 if mode == "gelu" else 0) | (1 << 10 if mode == "resid" else 0) | (1 << 11 if mode == "rinit" else 0)
    if mode in ("resid", "rinit"):
        aux = torch.zeros(max(M * N, 4 * N), device="cuda", dtype=torch.bfloat16)  # carries bias (N floats) and gamma (N floats)
        bias = torch.linspace(-1, 1, N, device="cuda")
        gamma = torch.linspace(0.5, 1.5, N, device="cuda")
        aux.view(torch.float32)[:N] = bias
        aux.view(torch.float32)[N:2 * N] = gamma
        x0 = torch.randn(M, N, device="cuda", generator=torch.Generator("cuda").manual_seed(M + N))
        out32[:M] = x0
        native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), out32.data_ptr(), aux.data_ptr(), M, N, K, flags, 1, stream, None))
        torch.cuda.synchronize()
        return out32, (x0, bias, gamma)
    o32 = out32.data_ptr() if mode == "f32" else None
    o16 = out16.data_ptr() if mode in ("bf16", "gelu") else None
    native.check(lib, lib.mdpt_debug_gemm(a.data_ptr(), w.data_ptr(), o32, o16, M, N, K, flags, 1, stream, None))
    torch.cuda.synchronize()
    return (out32 if mode == "f32" else out16), None


SHAPES = [(1, 256, 128), (17, 256, 256), (255, 512, 128), (256, 256, 1024), (257, 1024, 256), (300, 768, 384), (777, 1024, 1024),
          (1304, 3072, 1024), (1304, 1024, 4096), (2000, 256, 2304), (4099, 512, 640)]


@pytest.mark.parametrize("tile", [5, 2, 1, 6, 4])
@pytest.mark.parametrize("mode", ["f32", "bf16", "gelu", "resid", "rinit"])
def test_gemm_variants_on_ragged_shapes(lib, tile, mode):
    rng = np.random.default_rng(tile * 10 + len(mode))
    for (M, N, K) in SHAPES:
        # (tile 5 with an odd number of K tiles falls back to the lockstep 256x256 loop inside the launcher: K = 2304, 640, 128)
        a = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).cuda().to(torch.bfloat16)
        w = torch.from_numpy(rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).cuda().to(torch.bfloat16)
        out, extra = _run(lib, a, w, mode, tile)
        ref = a.float() @ w.float().t()
        if mode == "gelu":
            ref = torch.nn.functional.gelu(ref)
        if mode == "resid":
            x0, bias, gamma = extra
            ref = x0 + gamma * (ref + bias)
        if mode == "rinit":  # accumulators start at the residual, bias added by the epilogue (layer scale folded into W by the caller)
            x0, bias, _ = extra
            ref = x0 + ref + bias
        got = out[:M].float()
        scale = float(ref.abs().max())
        tol = 2e-5 * scale if mode in ("f32", "resid", "rinit") else 6e-3 * scale
        err = float((got - ref).abs().max())
        assert err <= tol, f"tile {tile} mode {mode} M={M} N={N} K={K}: max err {err:.3e} > {tol:.3e}"
        assert torch.all(out[M:] == 7.0), f"tile {tile} mode {mode} M={M} N={N} K={K}: wrote past row M"


def test_gemm_against_a_cpu_float64_product(lib):
    """The fuzz above takes its reference from a GPU matmul (rocBLAS through torch); this one does not depend on another GPU library
    being right: one ragged shape, every tile variant, against numpy float64 on the CPU (bf16 operands are exact in float64, so the
    only error left is the kernel's fp32 accumulation order: measured < 1e-6 of the output range)."""
    M, N, K = 777, 1024, 1024
    rng = np.random.default_rng(5)
    a = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(torch.bfloat16)
    w = torch.from_numpy(rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).to(torch.bfloat16)
    ref = a.float().numpy().astype(np.float64) @ w.float().numpy().astype(np.float64).T
    scale = float(np.abs(ref).max())
    for tile in (5, 2, 1, 6, 4):
        out, _ = _run(lib, a.cuda(), w.cuda(), "f32", tile)
        err = float(np.abs(out[:M].cpu().numpy().astype(np.float64) - ref).max())
        assert err <= 2e-6 * scale, f"tile {tile}: max err {err:.3e} (range {scale:.3f})"


def test_residual_initialised_accumulators_are_bitwise_identical_across_tile_variants(lib):
    """out = (resid + A W^T) + bias with the accumulators STARTING at the residual: the 8-phase kernel (16-byte loads, swapped MFMA operands)
    and the lockstep kernels (dword loads, plain order) must agree bit for bit - batch invariance of the encoder rests on it."""
    rng = np.random.default_rng(3)
    for (M, N, K) in [(1304, 1024, 1024), (777, 1024, 4096), (300, 768, 384), (4099, 512, 256)]:
        a = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).cuda().to(torch.bfloat16)
        w = torch.from_numpy(rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K)).cuda().to(torch.bfloat16)
        outs = [_run(lib, a, w, "rinit", tile)[0][:M].clone() for tile in (5, 2, 1, 6, 4)]
        for tile, o in zip((2, 1, 6, 4), outs[1:]):
            assert torch.equal(o, outs[0]), f"M={M} N={N} K={K}: tile {tile} differs from the 8-phase kernel"

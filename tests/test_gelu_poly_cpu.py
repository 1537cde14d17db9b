"""The erf-GELU of the GEMM epilogues (csrc/gemm_common.inc gelu_erf / gelu_erf2; reference: nn.GELU() default = exact erf form,
/root/reference/muggled_dpt/v2_depthanything/components/misc_helpers.py:113) restated in numpy with the SHIPPED coefficients
(read out of the source file) and fp32 fused multiply-adds, against scipy's erf. Pins the accuracy claim in the kernel's comment
(|error of erf| <= 1e-7) and the behaviour at the ends of the range (no NaN / inf from a finite input, gelu(-big) = 0, gelu(big) = big)."""
import os
import re

import numpy as np
from scipy.special import erf

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "muggled_dpt_amd", "csrc", "gemm_common.inc")
f32 = np.float32


def shipped_coefficients():
    text = open(SRC).read()
    q = [float(m) for m in re.findall(r"#define MDPT_GELU_Q\d\s+(-?[0-9.e+-]+)f", text)]
    assert len(q) == 8, q
    return [f32(c) for c in q]


def fma(a, b, c):
    # an fp32 product is exact in fp64; the fp64 sum then rounds once more to fp32 (double rounding differs from a true fma in < 1e-9 of cases)
    return (a.astype(np.float64) * np.float64(b) + np.float64(c)).astype(f32)


def gelu_kernel_arithmetic(v):
    q = shipped_coefficients()
    a = np.abs(v)
    acc = np.full_like(a, q[7])
    with np.errstate(over="ignore"):
        for c in q[6::-1]:
            acc = (acc.astype(np.float64) * a.astype(np.float64) + np.float64(c)).astype(f32)
        arg = (-(acc.astype(np.float64) * a.astype(np.float64))).astype(f32)
        erf_abs = f32(1.0) - np.exp2(arg.astype(np.float64)).astype(f32)
    h = f32(0.5) * v
    with np.errstate(invalid="ignore"):
        return fma(h, np.copysign(erf_abs, v), h), erf_abs


def test_erf_of_the_epilogue_is_within_1e7_of_scipy():
    v = np.linspace(-12.0, 12.0, 1_200_001).astype(f32)
    g, e = gelu_kernel_arithmetic(v)
    vd = v.astype(np.float64)
    assert np.abs(e - erf(np.abs(vd) / np.sqrt(2.0))).max() <= 1.0e-7
    ref = 0.5 * vd * (1.0 + erf(vd / np.sqrt(2.0)))
    # absolute error of the GELU: the erf error times |v| / 2, plus fp32 rounding of the result itself
    assert (np.abs(g - ref) <= 1.0e-7 * np.abs(vd) / 2 + 6.0e-8 * np.maximum(np.abs(ref), 1.0)).all()


def test_ends_of_the_range():
    v = np.concatenate([np.logspace(np.log10(5.8), np.log10(3.0e38), 200_001), [65504.0, 57344.0]]).astype(f32)
    g, e = gelu_kernel_arithmetic(v)
    assert (e == 1.0).all() and (g == v).all()
    g, e = gelu_kernel_arithmetic(-v)
    assert (e == 1.0).all() and (g == 0.0).all() and not np.isnan(g).any()
    g, _ = gelu_kernel_arithmetic(np.array([0.0, -0.0, 1e-30, -1e-30], f32))
    assert (np.abs(g) <= 1e-30).all()

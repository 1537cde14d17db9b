"""The OCP MX quantiser of the precision study (tests/precision_budget/emulate_operand_rounding.py): known values of the four
element formats, block scaling, and the property the plan rests on - a cross term computed on MX operands changes a split product by far less
than dropping the cross term does. CPU only."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "precision_budget"))
import emulate_operand_rounding as emu  # noqa: E402


def _blk(vals):
    x = torch.zeros(1, 32)
    x[0, :len(vals)] = torch.tensor(vals)
    return x


def test_mx_element_formats_known_values():
    # block maximum 7.9 -> shared scale 2^(2 - emax): e2m1 grid {0, .5, 1, 1.5, 2, 3, 4, 6}, saturating
    x = _blk([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, 0.3, 5.1, 7.9, -2.4])
    assert emu.mx_quant(x, "mxfp4", 1)[0, :12].tolist() == [0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0, 0.5, 6.0, 6.0, -2.0]
    # e2m3: quantum 1/8 below 2, 1/4 below 4, 1/2 up to 7.5
    assert emu.mx_quant(x, "mxfp6", 1)[0, 8:12].tolist() == [0.25, 5.0, 7.5, -2.5]
    # e4m3 with a block scale: 7900 -> scale 2^(12 - 8) = 16; 300 / 16 = 18.75 -> 18 (quantum 2) -> 288
    y = emu.mx_quant(x * 1000, "mxfp8", 1)[0]
    assert y[8].item() == 288.0 and y[10].item() == 7168.0 and y[1].item() == 512.0
    # blocks are independent: a second block of small values keeps its own scale
    z = torch.cat([x, x * 2.0 ** -10], dim=1)
    q = emu.mx_quant(z, "mxfp6", 1)
    assert torch.equal(q[:, 32:], q[:, :32] * 2.0 ** -10)
    # an all-zero block stays zero; a K that is not a multiple of 32 is padded, not truncated
    assert torch.equal(emu.mx_quant(torch.zeros(2, 40), "mxfp8", 1), torch.zeros(2, 40))
    assert emu.mx_quant(torch.full((1, 33), 3.0), "mxfp4", 1).tolist() == [[3.0] * 33]


def test_cross_terms_on_mx_operands_keep_the_split_product():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(64, 256, generator=g).relu() * 3.0
    w = torch.randn(48, 256, generator=g) / 16.0
    exact = (a.double() @ w.double().t())
    ah, wh = emu.rnd(a, "f16"), emu.rnd(w, "f16")
    one_pass = (ah.double() @ wh.double().t())
    two_pass = one_pass + ((a - ah).double() @ wh.double().t())
    err = lambda y: float((y - exact).abs().max() / exact.abs().max())
    for fmt, slack in (("mxfp8", 1.05), ("mxfp6", 1.05), ("mxfp4", 1.3)):
        mx = one_pass + (emu.mx_quant(a - ah, fmt, 1).double() @ emu.mx_quant(wh, fmt, 1).double().t())
        assert err(mx) <= slack * err(two_pass) + 2e-6, (fmt, err(mx), err(two_pass))
        assert err(mx) < 0.75 * err(one_pass), (fmt, err(mx), err(one_pass))

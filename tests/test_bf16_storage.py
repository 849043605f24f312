"""bf16 ACTIVATION STORAGE (the "bf16" mode of BASELINE configs 3 / 5 as a real engine: one bf16 plane is the only stored form of every
feature map between kernels; fp32 accumulate, statistics, master weights).

Kernel level: every `*_io` entry point on a bf16 tensor must compute EXACTLY what its fp32 form computes on the same values widened to
fp32 (same arithmetic, same order), the result rounded to nearest even once when it is stored -- so the bf16 engine differs from the
fp32-storage engine only by the roundings at the storage points (what oracle-level tolerance tests then bound).
Each test runs on the simulator (CPU) and on the MI355X (`hip`, gpu-marked) through the C ABI.
"""
import pytest
import torch

from test_kernels import P, err, nhwc, r4


def bf(t):
    """fp32 -> the nearest bf16 value (round to nearest even), as a bfloat16 tensor."""
    return t.detach().to(torch.bfloat16)


def rbf(t):
    return t.cpu().float().to(torch.bfloat16)


def test_bf16_to_f32_and_back(be):
    torch.manual_seed(0)
    x = torch.randn(37, 24) * 3
    xb = be.to(bf(x))
    y = be.empty(37, 28)
    be.L.cdf_bf16_to_f32(P(xb), 24, P(y), 28, 37, 24, be.stream())
    assert torch.equal(y[:, :24].cpu(), bf(x).float())                            # widening is exact
    back = torch.zeros(37, 24, dtype=torch.bfloat16, device=be.device)
    be.L.cdf_split_bf16(P(be.to(x)), 24, P(back), 0, 24, 37, 24, be.stream())     # ... and cdf_split_bf16(lo = NULL) is the rounding
    assert torch.equal(back.cpu(), bf(x))


@pytest.mark.parametrize("B,C,H", [(2, 8, 40), (1, 64, 16), (1, 36, 12)])
def test_dwconv7_bf16_io(be, B, C, H):
    torch.manual_seed(1)
    Cp = r4(C)
    x, res = torch.randn(B, H, H, Cp), torch.randn(B, H, H, Cp)
    w = torch.randn(C, 1, 7, 7) / 7
    wp = be.empty(49, Cp)
    be.L.cdf_pack_weight(P(be.to(w)), P(wp), 49, 1, C, Cp, 1, 0, 49, be.stream())
    bias, sb = be.to(torch.randn(Cp)), be.to(torch.randn(B, Cp))
    xb, rb = be.to(bf(x)), be.to(bf(res))
    xf, rf = be.to(bf(x).float()), be.to(bf(res).float())
    for flip, with_res, with_bias in ((0, False, True), (1, True, False), (1, False, False)):
        yref = be.empty(B, H, H, Cp)
        y = torch.zeros(B, H, H, Cp, dtype=torch.bfloat16, device=be.device)
        args = (P(wp), Cp, P(bias) if with_bias else 0, P(sb) if with_bias else 0, Cp)
        be.L.cdf_dwconv7(P(xf), Cp, *args, P(yref), Cp, B, H, H, Cp, flip, 0, P(rf) if with_res else 0, Cp, be.stream())
        be.L.cdf_dwconv7_io(P(xb), Cp, *args, P(y), Cp, B, H, H, Cp, flip, 0, P(rb) if with_res else 0, Cp, 1, be.stream())
        assert torch.equal(y.cpu(), rbf(yref)), (flip, with_res)
    # accumulate: y (bf16) += conv(x), read back in its own type
    y0 = torch.randn(B, H, H, Cp)
    yacc, yref = be.to(bf(y0)), be.to(bf(y0).float())
    be.L.cdf_dwconv7(P(xf), Cp, P(wp), Cp, 0, 0, 0, P(yref), Cp, B, H, H, Cp, 1, 1, 0, 0, be.stream())
    be.L.cdf_dwconv7_io(P(xb), Cp, P(wp), Cp, 0, 0, 0, P(yacc), Cp, B, H, H, Cp, 1, 1, 0, 0, 1, be.stream())
    assert torch.equal(yacc.cpu(), rbf(yref))
    # weight / bias / per-sample-bias gradients: fp32 results, bit-identical to the fp32 kernel on the widened tensors
    dy = torch.randn(B, H, H, Cp)
    dyb, dyf = be.to(bf(dy)), be.to(bf(dy).float())
    nch = be.L.cdf_dwconv7_wgrad_nchunk(H)
    out = []
    for io in (0, 1):
        ws, dw, dbias, dsb = be.empty(B * nch * 50 * C), be.zeros(C, 1, 7, 7), be.zeros(C), be.zeros(B, Cp)
        if io:
            be.L.cdf_dwconv7_wgrad_io(P(xb), Cp, P(dyb), Cp, P(dw), P(dbias), P(dsb), Cp, P(ws), B, H, H, C, 0, 1, be.stream())
        else:
            be.L.cdf_dwconv7_wgrad(P(xf), Cp, P(dyf), Cp, P(dw), P(dbias), P(dsb), Cp, P(ws), B, H, H, C, 0, be.stream())
        out.append((dw.cpu(), dbias.cpu(), dsb.cpu()))
    assert all(torch.equal(a, b) for a, b in zip(*out))


@pytest.mark.parametrize("M,C", [(70, 64), (33, 128), (9, 512), (40, 8)])
def test_layernorm_bf16_io(be, M, C):
    torch.manual_seed(2)
    x, g, b = torch.randn(M, C + 8), torch.randn(C), torch.randn(C)
    xb, xf, gd, bd = be.to(bf(x)), be.to(bf(x).float()), be.to(g), be.to(b)
    ld = C + 8
    outs = []
    for io in (0, 1):
        y, mo, ro = be.empty(M, C), be.empty(M), be.empty(M)
        yh = torch.zeros(M, C, dtype=torch.bfloat16, device=be.device)
        if io:
            be.L.cdf_layernorm_c_fwd_io(P(xb), ld, P(y), C, P(gd), P(bd), P(mo), P(ro), M, C, 1e-5, P(yh), 0, C, 1, be.stream())
        else:
            be.L.cdf_layernorm_c_fwd(P(xf), ld, P(y), C, P(gd), P(bd), P(mo), P(ro), M, C, 1e-5, P(yh), 0, C, be.stream())
        outs.append((y.cpu(), mo.cpu(), ro.cpu(), yh.cpu()))
    assert all(torch.equal(a, c) for a, c in zip(*outs))
    _, mo, ro, _ = (be.to(t) for t in outs[0])
    nb = be.L.cdf_layernorm_blocks(M, C)
    dy, add = torch.randn(M, C), torch.randn(M, C + 4)
    # ConvNeXt block form (io 7): dy, x, dx bf16
    part, dxr, dgr, dbr = be.empty(nb * 2 * C), be.empty(M, C), be.zeros(C), be.zeros(C)
    be.L.cdf_layernorm_c_bwd(P(be.to(bf(dy).float())), C, P(xf), ld, P(gd), P(mo), P(ro), P(dxr), C, 0, 0, P(dgr), P(dbr), P(part), M, C, 0, 0, be.stream())
    dx, dg, db = torch.zeros(M, C, dtype=torch.bfloat16, device=be.device), be.zeros(C), be.zeros(C)
    be.L.cdf_layernorm_c_bwd_io(P(be.to(bf(dy))), C, P(xb), ld, P(gd), P(mo), P(ro), P(dx), C, 0, 0, P(dg), P(db), P(part), M, C, 0, 0, 7, be.stream())
    assert torch.equal(dx.cpu(), rbf(dxr)) and torch.equal(dg.cpu(), dgr.cpu()) and torch.equal(db.cpu(), dbr.cpu())
    # attention block form (io 14): fp32 dy (the block's own fp32 gradient), bf16 x / dx / add (the stream)
    dxr2 = be.empty(M, C)
    be.L.cdf_layernorm_c_bwd(P(be.to(dy)), C, P(xf), ld, P(gd), P(mo), P(ro), P(dxr2), C, P(be.to(bf(add).float())), C + 4, P(dgr), P(dbr), P(part), M, C, 0, 0, be.stream())
    dx2 = torch.zeros(M, C, dtype=torch.bfloat16, device=be.device)
    be.L.cdf_layernorm_c_bwd_io(P(be.to(dy)), C, P(xb), ld, P(gd), P(mo), P(ro), P(dx2), C, P(be.to(bf(add))), C + 4, P(dg), P(db), P(part), M, C, 0, 0, 14, be.stream())
    assert torch.equal(dx2.cpu(), rbf(dxr2))
    from colddiff._lib import CdfError
    with pytest.raises(CdfError, match="io_bf16 = 5"):        # (a combination that is not instantiated is refused, not mis-read)
        be.L.cdf_layernorm_c_bwd_io(P(be.to(dy)), C, P(xb), ld, P(gd), P(mo), P(ro), P(dx2), C, 0, 0, P(dg), P(db), P(part), M, C, 0, 0, 5, be.stream())


@pytest.mark.parametrize("B,Cin,Cout,H", [(2, 64, 128, 16), (1, 128, 64, 16)])
def test_conv_gemm_bf16x_io_epilogue(be, B, Cin, Cout, H):
    """The pre-split GEMM with bf16 epilogue operands (residual read, pre-activation written, GELU' source read) against the same
    launch with those tensors in fp32: pre / output planes are the roundings of the fp32 launch's."""
    from colddiff import convdesc as cd
    torch.manual_seed(3)
    x = torch.randn(B, H, H, Cin)
    w = torch.randn(Cout, Cin, 3, 3) / (3 * Cin ** 0.5)
    bias = be.to(torch.randn(Cout))
    res, mul = torch.randn(B, H, H, Cout), torch.randn(B, H, H, Cout)
    plan = cd.conv_fwd(H, H, 3, 3, 1, 1, 1, 1, 1)
    ldk = (Cin + 31) // 32 * 32
    whi = torch.zeros(9, Cout, ldk, dtype=torch.int16, device=be.device)
    be.L.cdf_pack_weight_bf16(P(be.to(w)), P(whi), 0, 9, Cout, Cin, ldk, 1, Cin * 9, 9, be.stream())
    xb = be.to(bf(x))
    zero = be.zeros(64)
    outs = []
    for io in (0, 7):
        r_, m_ = (be.to(bf(res)), be.to(bf(mul))) if io else (be.to(bf(res).float()), be.to(bf(mul).float()))
        pre = torch.zeros(B, H, H, Cout, dtype=torch.bfloat16 if io else torch.float32, device=be.device)
        yh = torch.zeros(B, H, H, Cout, dtype=torch.bfloat16, device=be.device)
        yh2 = torch.zeros_like(yh)
        # forward form: bias -> pre -> GELU -> + res, planes only
        be.L.cdf_conv_gemm_bf16x_io(P(xb), 0, Cin, P(zero), P(whi), 0, ldk, 0, Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, plan.desc, P(bias), 0, 0,
                                    P(r_), Cout, P(pre), Cout, 0, 0, 1, 0, 0, io & 3, P(yh), 0, Cout, 0, 0, be.tune.ptr, be.stream())
        # data-gradient form: v *= GELU'(mul), planes only
        be.L.cdf_conv_gemm_bf16x_io(P(xb), 0, Cin, P(zero), P(whi), 0, ldk, 0, Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1, 1, plan.desc, 0, 0, 0,
                                    0, 0, 0, 0, P(m_), Cout, 0, 1, 0, io & 4, P(yh2), 0, Cout, 0, 0, be.tune.ptr, be.stream())
        outs.append((pre.cpu(), yh.cpu(), yh2.cpu()))
    assert torch.equal(outs[1][0], rbf(outs[0][0])) and torch.equal(outs[1][1], outs[0][1]) and torch.equal(outs[1][2], outs[0][2])
    assert outs[0][1].float().abs().max() > 0.1

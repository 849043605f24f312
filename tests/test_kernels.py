"""Kernel-level parity tests: every C-ABI entry point against a plain torch fp32 CPU reference of
the reference op it replaces.  Each test runs on two backends (see conftest.Backend):
  emu : host SIMT simulator build, small shapes (not gpu)       — indexing logic
  hip : libcolddiff_hip.so on a real MI355X (@pytest.mark.gpu)  — the product
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from colddiff import convdesc as cd


def P(t):
    return 0 if t is None else t.data_ptr()


def r4(c):
    return (c + 3) // 4 * 4


def nhwc(x, ld=None):
    B, C, H, W = x.shape
    ld = ld or r4(C)
    out = torch.zeros(B, H, W, ld)
    out[..., :C] = x.detach().permute(0, 2, 3, 1)
    return out


def err(a, b):
    return (a.cpu() - b.cpu()).abs().max().item()


# ---------------------------------------------------------------------------------------------
# degradations
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,k,mode", [(16, 3, "circular"), (16, 11, "reflect"), (32, 11, "circular"), (16, 5, "reflect"),
                                      (32, 15, "reflect"), (12, 3, "reflect")])
def test_blur_chain(be, H, k, mode):
    torch.manual_seed(0)
    B, C, T = 3, 3, 5
    x = torch.randn(B, C, H, H)
    taps = torch.rand(T, C, k, k)
    taps /= taps.sum((2, 3), keepdim=True)
    t = torch.tensor([4, 2, 0])
    xd, tapsd, td = be.to(x), be.to(taps), be.to(t)
    y, snap = be.empty(B, C, H, H), be.empty(B, C, H, H)
    be.L.cdf_blur_chain(P(xd), P(y), P(snap), 0, P(tapsd), P(td), B, C, H, H, k, 0, 0, 0 if mode == "circular" else 1, -1, 0, be.stream())

    def step(z, i):
        return F.conv2d(F.pad(z, (k // 2,) * 4, mode=mode), taps[i].unsqueeze(1), groups=C)

    ref, refp = [], []
    for b in range(B):
        z = x[b:b + 1]
        prev = z
        for i in range(int(t[b]) + 1):
            prev, z = z, step(z, i)
        ref.append(z)
        refp.append(prev)
    ref, refp = torch.cat(ref), torch.cat(refp)
    assert err(y, ref) <= 2e-6 and err(snap, refp) <= 2e-6
    # Algorithm-2 combine: img - D_t + D_{t-1}
    img = torch.randn(B, C, H, H)
    out = be.empty(B, C, H, H)
    be.L.cdf_blur_chain(P(xd), P(out), 0, P(be.to(img)), P(tapsd), P(td), B, C, H, H, k, 0, 0, 0 if mode == "circular" else 1, -1, 0, be.stream())
    assert err(out, img - ref + refp) <= 4e-6
    # discrete: mean collapse after step 2 + 8-bit truncation
    yq = be.empty(B, C, H, H)
    be.L.cdf_blur_chain(P(xd), P(yq), 0, 0, P(tapsd), 0, B, C, H, H, k, 0, 2, 0 if mode == "circular" else 1, 2, 1, be.stream())
    z = x
    for i in range(3):
        z = step(z, i)
    z = z.mean((2, 3), keepdim=True).expand_as(x)
    zq = ((z + 1) * 0.5 * 255).int().float() / 255 * 2 - 1
    d = (yq.cpu() - zq).abs()
    # truncation may flip one 8-bit level (2/255) where the mean differs in the last ulp
    assert (d <= 1e-6).logical_or((d - 2 / 255).abs() <= 1e-6).all()


@pytest.mark.parametrize("H,k,mode", [(16, 3, "circular"), (16, 11, "reflect"), (32, 15, "reflect"), (32, 15, "circular"),
                                      (12, 5, "reflect"), (40, 27, "reflect")])
def test_blur_chain_separable(be, H, k, mode):
    """Rank-one (Gaussian g (x) g) kernels through the separable LDS-resident chain: same semantics as cdf_blur_chain
    (per-sample t, snapshot of the previous state, Alg. 2 combine, discrete collapse + quantise), results equal to the
    dense depthwise conv of the reference (DEBLUR:351-361) up to rounding."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "cold-diffusion-models_amd"))
    from colddiff import degrade as D
    torch.manual_seed(0)
    B, C, T = 3, 3, 5
    x = torch.randn(B, C, H, H)
    taps = torch.stack([torch.stack([D.gaussian_kernel2d(k, 0.6 + 0.7 * i + 0.1 * c) for c in range(C)]) for i in range(T)])
    t1 = D.separable_taps(taps)
    assert t1 is not None and t1.shape == (T, C, 2, k)
    assert D.separable_taps(torch.rand(T, C, k, k)) is None          # generic kernels must be refused
    t = torch.tensor([4, 2, 0])
    pm = 0 if mode == "circular" else 1
    xd, t1d, td = be.to(x), be.to(t1), be.to(t)
    y, snap = be.empty(B, C, H, H), be.empty(B, C, H, H)
    be.L.cdf_blur_chain_sep(P(xd), P(y), P(snap), 0, P(t1d), P(td), B, C, H, H, k, 0, 0, pm, -1, 0, be.stream())

    def step(z, i):
        return F.conv2d(F.pad(z, (k // 2,) * 4, mode=mode), taps[i].unsqueeze(1), groups=C)

    ref, refp = [], []
    for b in range(B):
        z = x[b:b + 1]
        prev = z
        for i in range(int(t[b]) + 1):
            prev, z = z, step(z, i)
        ref.append(z)
        refp.append(prev)
    ref, refp = torch.cat(ref), torch.cat(refp)
    assert err(y, ref) <= 3e-6 and err(snap, refp) <= 3e-6
    img = torch.randn(B, C, H, H)
    out = be.empty(B, C, H, H)
    be.L.cdf_blur_chain_sep(P(xd), P(out), 0, P(be.to(img)), P(t1d), P(td), B, C, H, H, k, 0, 0, pm, -1, 0, be.stream())
    assert err(out, img - ref + refp) <= 6e-6
    out2, snap2 = be.empty(B, C, H, H), be.empty(B, C, H, H)
    be.L.cdf_blur_chain_sep(P(xd), P(out2), P(snap2), P(be.to(img)), P(t1d), P(td), B, C, H, H, k, 0, 0, pm, -1, 0, be.stream())
    assert err(out2, img - ref + refp) <= 6e-6 and err(snap2, refp) <= 3e-6
    yq = be.empty(B, C, H, H)
    be.L.cdf_blur_chain_sep(P(xd), P(yq), 0, 0, P(t1d), 0, B, C, H, H, k, 0, 2, pm, 2, 1, be.stream())
    z = x
    for i in range(3):
        z = step(z, i)
    z = z.mean((2, 3), keepdim=True).expand_as(x)
    zq = ((z + 1) * 0.5 * 255).int().float() / 255 * 2 - 1
    d = (yq.cpu() - zq).abs()
    assert (d <= 1e-6).logical_or((d - 2 / 255).abs() <= 1e-6).all()


def test_blur_step_global(be):
    torch.manual_seed(1)
    B, C, H, W, k = 2, 3, 10, 14, 5
    x, taps = torch.randn(B, C, H, W), torch.rand(C, k, k)
    y = be.empty(B, C, H, W)
    be.L.cdf_blur_step(P(be.to(x)), P(y), P(be.to(taps)), B, C, H, W, k, 1, be.stream())
    ref = F.conv2d(F.pad(x, (k // 2,) * 4, mode="reflect"), taps.unsqueeze(1), groups=C)
    assert err(y, ref) <= 2e-6


@pytest.mark.parametrize("H", [16, 32])
@pytest.mark.parametrize("mode_i,mode", [(0, "area"), (1, "bilinear"), (2, "bicubic")])
@pytest.mark.parametrize("routine", ["factor2", "incr"])
def test_pixelate_chain(be, H, mode_i, mode, routine):
    torch.manual_seed(0)
    B, C = 2, 3
    T = 3 if routine == "factor2" else 6
    sizes = [H // 2 ** (i + 1) for i in range(T)] if routine == "factor2" else [H - i for i in range(T)]
    x = torch.randn(B, C, H, H)
    t = torch.tensor([T - 1, 1])
    y, snap = be.empty(B, C, H, H), be.empty(B, C, H, H)
    sz = be.to(torch.tensor(sizes, dtype=torch.int32))
    be.L.cdf_pixelate_chain(P(be.to(x)), P(y), P(snap), 0, P(sz), P(be.to(t)), B, C, H, 0, 0, mode_i, be.stream())

    def step(z, i):
        z1 = F.interpolate(z, size=sizes[i], mode=mode, antialias=False)
        return F.interpolate(z1, size=H, mode="nearest-exact")

    ref, refp = [], []
    for b in range(B):
        z = x[b:b + 1]
        prev = z
        for i in range(int(t[b]) + 1):
            prev, z = z, step(z, i)
        ref.append(z)
        refp.append(prev)
    tol = 0.0 if mode == "area" else 1e-5   # area + nearest-exact indexing is bit-exact
    assert err(y, torch.cat(ref)) <= tol and err(snap, torch.cat(refp)) <= tol


def test_mask_noise_loss_layout(be):
    torch.manual_seed(0)
    L, S = be.L, be.stream()
    B, C, H, T = 3, 3, 16, 6
    x, masks = torch.randn(B, C, H, H), torch.rand(T, 2 * H + 1, 2 * H + 1)
    t, oy, ox = torch.tensor([5, 0, 3]), torch.tensor([0, 7, 16]), torch.tensor([3, 0, 16])
    y, snap = be.empty(B, C, H, H), be.empty(B, C, H, H)
    L.cdf_mask_chain(P(be.to(x)), P(y), P(snap), 0, P(be.to(masks)), P(be.to(t)), P(be.to(oy)), P(be.to(ox)),
                     B, C, H, H, 2 * H + 1, 2 * H + 1, 0, 0, 0, S)
    ref = []
    for b in range(B):
        z = x[b]
        for i in range(int(t[b]) + 1):
            z = masks[i][oy[b]:oy[b] + H, ox[b]:ox[b] + H] * z
        ref.append(z)
    assert err(y, torch.stack(ref)) == 0.0          # sequential products, bit-exact
    ca, cb = torch.rand(10), torch.rand(10)
    x0, eps, t = torch.randn(B, C, H, H), torch.randn(B, C, H, H), torch.tensor([9, 0, 4])
    out = be.empty(B, C, H, H)
    L.cdf_noise_qsample(P(be.to(x0)), P(be.to(eps)), P(be.to(ca)), P(be.to(cb)), P(be.to(t)), P(out), B, C * H * H, S)
    assert err(out, ca[t].view(-1, 1, 1, 1) * x0 + cb[t].view(-1, 1, 1, 1) * eps) == 0.0
    for tt in (5, 1):
        for est in (0, 1):
            img, x1 = torch.randn(B, C, H, H), torch.randn(B, C, H, H)
            L.cdf_noise_step(P(be.to(img)), P(be.to(x1)), P(be.to(eps)), P(be.to(ca)), P(be.to(cb)), tt, est, P(out), img.numel(), S)
            x2 = (img - ca[tt - 1] * x1) / cb[tt - 1] if est else eps
            xt = ca[tt - 1] * x1 + cb[tt - 1] * x2
            xs = ca[tt - 2] * x1 + cb[tt - 2] * x2 if tt - 1 != 0 else x1
            assert err(out, img - xt + xs) <= (2e-6 if est else 0.0)
    a, b = torch.randn(4, 3, 16, 16), torch.randn(4, 3, 16, 16)
    for l2 in (0, 1):
        o, part, g, gy = be.zeros(1), be.zeros(1024), be.to(torch.tensor([0.5])), be.empty(4, 3, 16, 16)
        L.cdf_loss_fwd(P(be.to(a)), P(be.to(b)), P(o), P(part), a.numel(), l2, S)
        L.cdf_loss_bwd(P(be.to(a)), P(be.to(b)), P(g), P(gy), a.numel(), l2, S)
        bb = b.clone().requires_grad_()
        ref = (a - bb).abs().mean() if not l2 else F.mse_loss(a, bb)
        (ref * 0.5).backward()
        assert abs(o.item() - ref.item()) <= 1e-6 * max(1, abs(ref.item())) and err(gy, bb.grad) <= 1e-7
    xx, yy = torch.randn(2, 5, 4, 4), be.zeros(2, 4, 4, 8)
    L.cdf_nchw_to_nhwc(P(be.to(xx)), P(yy), 2, 5, 16, 8, S)
    assert err(yy[..., :5], xx.permute(0, 2, 3, 1)) == 0.0
    zz = be.empty(2, 5, 4, 4)
    L.cdf_nhwc_to_nchw(P(yy), P(zz), P(be.to(xx)), 2, 5, 16, 8, S)
    assert err(zz, 2 * xx) == 0.0


# ---------------------------------------------------------------------------------------------
# MFMA implicit-GEMM convolutions (fwd / dgrad / wgrad)
# ---------------------------------------------------------------------------------------------
def _pack(be, w, T, R, C, s_t, s_r, s_c):
    dst = be.empty(T, R, r4(C))
    be.L.cdf_pack_weight(P(be.to(w)), P(dst), T, R, C, r4(C), s_t, s_r, s_c, be.stream())
    return dst


def _gemm(be, plan, x, w, B, Cin, Cout, bias=None, sbias=None, res=None, pre=None, mul=None, act=0, mul_mode=0, acc=0, y=None):
    if y is None:
        y = be.zeros(B, plan.OH, plan.OW, r4(Cout))
    ld = lambda t_: 0 if t_ is None else t_.shape[-1]
    be.L.cdf_conv_gemm(P(x), x.shape[-1], P(w), w.shape[-1], P(y), y.shape[-1], B, plan.H, plan.W, Cin, plan.OH, plan.OW, Cout,
                       plan.QH, plan.QW, plan.os, plan.istride, plan.nphase, plan.desc, P(bias), P(sbias), ld(sbias), P(res), ld(res),
                       P(pre), ld(pre), P(mul), ld(mul), act, mul_mode, acc, 0, 1, 0, 0, 0, 1, 0, 0, 0, be.stream())
    return y


CONV_CASES = [(2, 8, 8, 8, 3, 1, 1, False), (1, 3, 20, 8, 3, 1, 1, False), (2, 16, 40, 8, 1, 1, 0, False), (1, 8, 8, 8, 4, 2, 1, False),
              (1, 8, 8, 4, 4, 2, 1, True), (1, 36, 130, 8, 3, 1, 1, False), (3, 20, 3, 4, 1, 1, 0, False)]
CONV_CASES_GPU = [(4, 64, 128, 32, 3, 1, 1, False), (2, 128, 64, 32, 3, 1, 1, False), (2, 64, 64, 32, 4, 2, 1, False),
                  (2, 64, 64, 16, 4, 2, 1, True), (2, 256, 512, 16, 3, 1, 1, False), (3, 64, 3, 32, 1, 1, 0, False), (2, 3, 128, 64, 3, 1, 1, False)]


def _conv_case(be, B, Cin, Cout, H, k, s, p, transposed):
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, H, requires_grad=True)
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    w = (torch.randn(*wshape) * (1.0 / math.sqrt(Cin * k * k))).requires_grad_()
    bias = torch.randn(Cout)
    yref = F.conv_transpose2d(x, w, bias, stride=s, padding=p) if transposed else F.conv2d(x, w, bias, stride=s, padding=p)
    gy = torch.randn_like(yref)
    yref.backward(gy)
    xn, gyn, KK = be.to(nhwc(x)), be.to(nhwc(gy)), k * k
    if not transposed:
        plan, pd, wg = cd.conv_fwd(H, H, k, k, s, p, p, p, p), cd.conv_dgrad(H, H, k, k, s, p, p, p, p), cd.conv_wgrad(H, H, k, k, s, p, p, p, p)
        wp, wd = _pack(be, w, KK, Cin, Cout, 1, KK, Cin * KK), _pack(be, w, KK, Cout, Cin, 1, Cin * KK, KK)
        s_r, s_c = KK, Cin * KK
    else:
        plan, pd, wg = cd.convT_fwd(H, H, k, k, s, p), cd.convT_dgrad(H, H, k, k, s, p), cd.convT_wgrad(H, H, k, k, s, p)
        wp, wd = _pack(be, w, KK, Cin, Cout, 1, Cout * KK, KK), _pack(be, w, KK, Cout, Cin, 1, KK, Cout * KK)
        s_r, s_c = Cout * KK, KK
    y = _gemm(be, plan, xn, wp, B, Cin, Cout, bias=be.to(bias))
    dx = _gemm(be, pd, gyn, wd, B, Cout, Cin)
    M = B * wg.QH * wg.QW
    ns = max(1, min(be.L.cdf_wgrad_nsplit(M, Cin, Cout, KK), M // 16))
    ldo = r4(Cout)
    ws = be.empty(ns, KK, Cin, ldo)
    bsum = be.empty(ns, ldo) if not transposed else None
    be.L.cdf_conv_wgrad(P(xn), xn.shape[-1], P(gyn), gyn.shape[-1], P(ws), ldo, B, wg.QH, wg.QW, wg.HA, wg.WA, wg.sa, wg.HB, wg.WB, wg.sb,
                        Cin, Cout, wg.ntaps, wg.desc, ns, 1, 0, 0, 0, P(bsum), be.stream())
    dw = be.zeros(*wshape)
    be.L.cdf_unpack_reduce(P(ws), P(dw), ns, KK, Cin, Cout, ldo, 1, s_r, s_c, 0, 1, be.stream())
    if bsum is not None:                        # fused bias gradient = column sums of dY
        db = be.zeros(Cout)
        be.L.cdf_unpack_reduce(P(bsum), P(db), ns, 1, 1, Cout, ldo, 0, 0, 1, 0, 1, be.stream())
        assert err(db, gy.sum((0, 2, 3))) <= 2e-6 * max(1.0, gy.sum((0, 2, 3)).abs().max().item()) * math.sqrt(M)
    tol = lambda ref: 2e-6 * max(1.0, ref.abs().max().item()) * math.sqrt(max(Cin * KK, 16))
    assert err(y[..., :Cout].permute(0, 3, 1, 2), yref) <= tol(yref)
    assert err(dx[..., :Cin].permute(0, 3, 1, 2), x.grad) <= tol(x.grad)
    assert err(dw, w.grad) <= 2e-6 * max(1.0, w.grad.abs().max().item()) * math.sqrt(M)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_gemm(be, case):
    _conv_case(be, *case)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CONV_CASES_GPU)
def test_conv_gemm_large(case):
    from conftest import Backend
    _conv_case(Backend("hip"), *case)


def test_conv_epilogue_and_batched(be):
    """bias + per-sample bias + GELU (+pre) + residual; activation-gradient multiply; batched NT GEMM."""
    torch.manual_seed(0)
    B, Cin, Cout, H = 2, 8, 12, 4
    x, w = torch.randn(B, Cin, H, H), torch.randn(Cout, Cin, 3, 3) * 0.2
    bias, sb, res = torch.randn(Cout), torch.randn(B, Cout), torch.randn(B, Cout, H, H)
    plan = cd.conv_fwd(H, H, 3, 3, 1, 1, 1, 1, 1)
    wp = _pack(be, w, 9, Cin, Cout, 1, 9, Cin * 9)
    pre = be.zeros(B, H, H, r4(Cout))
    y = _gemm(be, plan, be.to(nhwc(x)), wp, B, Cin, Cout, bias=be.to(bias), sbias=be.to(sb), res=be.to(nhwc(res)), pre=pre, act=1)
    z = F.conv2d(x, w, bias, padding=1) + sb[:, :, None, None]
    assert err(pre[..., :Cout].permute(0, 3, 1, 2), z) <= 1e-5
    assert err(y[..., :Cout].permute(0, 3, 1, 2), F.gelu(z) + res) <= 1e-5
    mulsrc = torch.randn(B, Cout, H, H)
    for mode, fn in ((1, lambda v: torch.autograd.functional.jvp(F.gelu, v, torch.ones_like(v))[1]),
                     (2, lambda v: torch.autograd.functional.jvp(F.silu, v, torch.ones_like(v))[1]), (3, lambda v: v)):
        y2 = _gemm(be, plan, be.to(nhwc(x)), wp, B, Cin, Cout, mul=be.to(nhwc(mulsrc)), mul_mode=mode)
        assert err(y2[..., :Cout].permute(0, 3, 1, 2), F.conv2d(x, w, padding=1) * fn(mulsrc)) <= 1e-5
    y3 = _gemm(be, plan, be.to(nhwc(x)), wp, B, Cin, Cout, acc=1, y=be.to(nhwc(res)))
    assert err(y3[..., :Cout].permute(0, 3, 1, 2), F.conv2d(x, w, padding=1) + res) <= 1e-5
    # batched S[b] = q[b] @ k[b]^T  (b_trans) and O[b] = P[b] @ v[b]
    nb, n, C = 3, 20, 16
    q, k = torch.randn(nb, n, C), torch.randn(nb, n, C)
    S = be.zeros(nb, n, r4(n))
    one = cd.conv_fwd(1, n, 1, 1, 1, 0, 0, 0, 0)
    qd, kd = be.to(q), be.to(k)
    be.L.cdf_conv_gemm(P(qd), C, P(kd), C, P(S), r4(n), 1, 1, n, C, 1, n, n, 1, n, 1, 1, 1, one.desc, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                       0, 0, 0, 1, nb, n * C, n * C, n * r4(n), 1, 0, 0, 0, be.stream())
    assert err(S[..., :n], q @ k.transpose(1, 2)) <= 1e-5
    # wgrad kernel as batched A^T B:  dv[b] = P[b]^T dO[b]
    Pm, dO = torch.randn(nb, n, r4(n)), torch.randn(nb, n, C)
    ws = be.empty(nb, n, C)
    tap = cd.conv_wgrad(1, n, 1, 1, 1, 0, 0, 0, 0)
    be.L.cdf_conv_wgrad(P(be.to(Pm)), r4(n), P(be.to(dO)), C, P(ws), C, 1, 1, n, 1, n, 1, 1, n, 1, n, C, 1, tap.desc, 1, nb,
                        n * r4(n), n * C, n * C, 0, be.stream())
    assert err(ws, Pm[..., :n].transpose(1, 2) @ dO) <= 1e-5


def test_colsum(be):
    torch.manual_seed(0)
    x = torch.randn(3, 700, 72)
    nch = be.L.cdf_colsum_nchunk(700)
    ws, out = be.empty(3 * nch * 70), be.zeros(3, 72)
    be.L.cdf_colsum(P(be.to(x)), P(out), P(ws), 3, 700, 70, 72, 72, 0, be.stream())
    assert err(out[:, :70], x[..., :70].sum(1)) <= 2e-4
    # bf16 input (cdf_colsum_io): the same sums of the widened values
    xb = x.bfloat16()
    out2 = be.zeros(3, 72)
    be.L.cdf_colsum_io(P(be.to(xb.view(torch.int16))), P(out2), P(ws), 3, 700, 70, 72, 72, 0, 1, be.stream())
    assert err(out2[:, :70], xb.float()[..., :70].sum(1)) <= 2e-4


# ---------------------------------------------------------------------------------------------
# norms, depthwise conv, attention, small ops
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,C", [(37, 8), (20, 64), (9, 128), (5, 512), (3, 1024), (6, 96)])
def test_layernorm_c(be, M, C):
    torch.manual_seed(0)
    x, g, b = torch.randn(M, C, requires_grad=True), torch.randn(C, requires_grad=True), torch.randn(C, requires_grad=True)
    var, mean = torch.var(x, dim=1, unbiased=False, keepdim=True), x.mean(1, keepdim=True)
    yref = (x - mean) / (var + 1e-5).sqrt() * g + b
    dy = torch.randn(M, C)
    yref.backward(dy)
    xd, gd, bd = be.to(x), be.to(g), be.to(b)
    y, mo, ro = be.empty(M, C), be.empty(M), be.empty(M)
    be.L.cdf_layernorm_c_fwd(P(xd), C, P(y), C, P(gd), P(bd), P(mo), P(ro), M, C, 1e-5, 0, 0, 0, be.stream())
    nb = be.L.cdf_layernorm_blocks(M, C)
    part, dx, dg, db = be.empty(nb * 2 * C), be.empty(M, C), be.zeros(C), be.zeros(C)
    be.L.cdf_layernorm_c_bwd(P(be.to(dy)), C, P(xd), C, P(gd), P(mo), P(ro), P(dx), C, 0, 0, P(dg), P(db), P(part), M, C, 0, 0, be.stream())
    assert err(y, yref) <= 5e-6 and err(dx, x.grad) <= 1e-5 and err(dg, g.grad) <= 2e-5 and err(db, b.grad) <= 2e-5
    # dx = grad + add (the residual branch, a pitched second tensor) and the in-place accumulate form
    addt = torch.randn(M, C + 4)
    addd, dx2 = be.to(addt), be.empty(M, C)
    be.L.cdf_layernorm_c_bwd(P(be.to(dy)), C, P(xd), C, P(gd), P(mo), P(ro), P(dx2), C, P(addd), C + 4, P(dg), P(db), P(part), M, C, 0, 0, be.stream())
    assert err(dx2, x.grad + addt[:, :C]) <= 1e-5
    dx3 = be.to(addt[:, :C].contiguous())
    be.L.cdf_layernorm_c_bwd(P(be.to(dy)), C, P(xd), C, P(gd), P(mo), P(ro), P(dx3), C, 0, 0, P(dg), P(db), P(part), M, C, 1, 0, be.stream())
    assert torch.equal(dx3.cpu(), dx2.cpu())
    if C % 8 == 0:
        # fused operand split: the bf16 hi / lo planes must equal cdf_split_bf16 of the stored output, bit for bit
        y2 = be.empty(M, C)
        yh = torch.zeros(M, C, dtype=torch.int16, device=be.device)
        yl = torch.zeros_like(yh)
        be.L.cdf_layernorm_c_fwd(P(xd), C, P(y2), C, P(gd), P(bd), 0, 0, M, C, 1e-5, P(yh), P(yl), C, be.stream())
        rh, rl = _split(be, y2)
        assert torch.equal(y2.cpu(), y.cpu()) and torch.equal(yh.cpu(), rh.cpu()) and torch.equal(yl.cpu(), rl.cpu())
        # ... and the backward's dx (cdf_layernorm_c_bwd_planes, with the residual operand): same dx, planes = its cdf_split_bf16
        dx4, ph, pl = be.empty(M, C), torch.zeros(M, C, dtype=torch.int16, device=be.device), torch.zeros(M, C, dtype=torch.int16, device=be.device)
        be.L.cdf_layernorm_c_bwd_planes(P(be.to(dy)), C, P(xd), C, P(gd), P(mo), P(ro), P(dx4), C, P(addd), C + 4, P(dg), P(db), P(part), M, C, 0, 0,
                                        P(ph), P(pl), C, be.stream())
        rh, rl = _split(be, dx4)
        assert torch.equal(dx4.cpu(), dx2.cpu()) and torch.equal(ph.cpu(), rh.cpu()) and torch.equal(pl.cpu(), rl.cpu())


@pytest.mark.parametrize("B,HW,C,silu", [(2, 16, 32, 1), (3, 64, 64, 1), (1, 16, 128, 0), (2, 300, 96, 1)])
def test_groupnorm(be, B, HW, C, silu):
    torch.manual_seed(0)
    x, ga, bt = torch.randn(B, HW, C, requires_grad=True), torch.randn(C, requires_grad=True), torch.randn(C, requires_grad=True)
    z = F.group_norm(x.permute(0, 2, 1).reshape(B, C, HW, 1), 32, ga, bt, eps=1e-6)
    yref = (z * torch.sigmoid(z) if silu else z).reshape(B, C, HW).permute(0, 2, 1)
    dy = torch.randn(B, HW, C)
    yref.backward(dy)
    nch = be.L.cdf_groupnorm_nchunk(HW)
    ws, mean, rstd, y = be.empty(B * nch * 2 * C + B * 2 * C + B * 64), be.empty(B * 32), be.empty(B * 32), be.empty(B, HW, C)
    xd, gd, bd = be.to(x), be.to(ga), be.to(bt)
    be.L.cdf_groupnorm_fwd(P(xd), C, P(y), C, P(gd), P(bd), P(mean), P(rstd), P(ws), B, HW, C, 32, 1e-6, silu, be.stream())
    dx, dga, dbe = be.empty(B, HW, C), be.zeros(C), be.zeros(C)
    be.L.cdf_groupnorm_bwd(P(be.to(dy)), C, P(xd), C, P(gd), P(bd), P(mean), P(rstd), P(dx), C, P(dga), P(dbe), P(ws), B, HW, C, 32, silu, 0, 0, be.stream())
    assert err(y, yref) <= 1e-5 and err(dx, x.grad) <= 2e-5 and err(dga, ga.grad) <= 1e-4 and err(dbe, bt.grad) <= 1e-4
    # the fused tail of ResnetBlock.forward: dropout with cdf_dropout's mask, the output again as bf16 planes (bit-equal to
    # cdf_split_bf16 of it), with and without the fp32 copy
    pdrop, seed, ld8 = 0.25, 12345, (C + 7) // 8 * 8
    yd, y2 = be.empty(B, HW, C), be.empty(B, HW, C)
    be.L.cdf_dropout(P(y), C, P(yd), C, B * HW, C, pdrop, seed, be.stream())
    hi, lo = (torch.zeros(B, HW, ld8, dtype=torch.int16, device=be.device) for _ in range(2))
    be.L.cdf_groupnorm_fwd_ex(P(xd), C, P(y2), C, P(gd), P(bd), P(mean), P(rstd), P(ws), B, HW, C, 32, 1e-6, silu, pdrop, seed, P(hi), P(lo), ld8,
                              be.stream())
    assert torch.equal(y2.cpu(), yd.cpu()) and 0.5 < float((yd.cpu() != 0).float().mean()) < 0.95
    rh, rl = (torch.zeros(B, HW, ld8, dtype=torch.int16, device=be.device) for _ in range(2))
    be.L.cdf_split_bf16(P(yd), C, P(rh), P(rl), ld8, B * HW, C, be.stream())
    assert torch.equal(hi.cpu(), rh.cpu()) and torch.equal(lo.cpu(), rl.cpu())
    hi2 = torch.zeros_like(hi)
    be.L.cdf_groupnorm_fwd_ex(P(xd), C, 0, 0, P(gd), P(bd), P(mean), P(rstd), P(ws), B, HW, C, 32, 1e-6, silu, pdrop, seed, P(hi2), 0, ld8, be.stream())
    assert torch.equal(hi2.cpu(), rh.cpu())


@pytest.mark.parametrize("B,C,H", [(2, 8, 8), (1, 3, 8), (2, 64, 4), (1, 68, 12), (3, 8, 40), (1, 36, 40), (5, 4, 16)])
def test_dwconv7(be, B, C, H):
    """(40 x 40 images: 2 x 5 tiles of 32 x 8 pixels per image, ragged at the right edge; 16 x 16: the square tile.)"""
    torch.manual_seed(0)
    Cp = r4(C)
    x, w = torch.randn(B, C, H, H, requires_grad=True), (torch.randn(C, 1, 7, 7) / 7).requires_grad_()
    bias, sb = torch.randn(C, requires_grad=True), torch.randn(B, C, requires_grad=True)
    yref = F.conv2d(x, w, bias, padding=3, groups=C) + sb[:, :, None, None]
    dy = torch.randn_like(yref)
    yref.backward(dy)
    xn, dyn = be.to(nhwc(x)), be.to(nhwc(dy))
    wp = be.empty(49, Cp)
    be.L.cdf_pack_weight(P(be.to(w)), P(wp), 49, 1, C, Cp, 1, 0, 49, be.stream())
    bp, sbp = torch.zeros(Cp), torch.zeros(B, Cp)
    bp[:C], sbp[:, :C] = bias.detach(), sb.detach()
    y, dx = be.empty(B, H, H, Cp), be.empty(B, H, H, Cp)
    be.L.cdf_dwconv7(P(xn), Cp, P(wp), Cp, P(be.to(bp)), P(be.to(sbp)), Cp, P(y), Cp, B, H, H, Cp, 0, 0, 0, 0, be.stream())
    rs = torch.randn(B, H, H, Cp)
    be.L.cdf_dwconv7(P(dyn), Cp, P(wp), Cp, 0, 0, 0, P(dx), Cp, B, H, H, Cp, 1, 0, 0, 0, be.stream())
    dxr = be.empty(B, H, H, Cp)
    be.L.cdf_dwconv7(P(dyn), Cp, P(wp), Cp, 0, 0, 0, P(dxr), Cp, B, H, H, Cp, 1, 0, P(be.to(rs)), Cp, be.stream())     # fused residual
    assert err(dxr[..., :C], dx[..., :C].cpu() + rs[..., :C]) <= 1e-6
    dxa = be.to(rs.clone())                                                                                     # += old contents
    be.L.cdf_dwconv7(P(dyn), Cp, P(wp), Cp, 0, 0, 0, P(dxa), Cp, B, H, H, Cp, 1, 1, 0, 0, be.stream())
    assert err(dxa[..., :C], dx[..., :C].cpu() + rs[..., :C]) <= 1e-6
    rs2 = torch.randn(B, H, H, Cp)
    dxb = be.to(rs.clone())                                                                                     # both at once
    be.L.cdf_dwconv7(P(dyn), Cp, P(wp), Cp, 0, 0, 0, P(dxb), Cp, B, H, H, Cp, 1, 1, P(be.to(rs2)), Cp, be.stream())
    assert err(dxb[..., :C], dx[..., :C].cpu() + rs[..., :C] + rs2[..., :C]) <= 2e-6
    if Cp % 8 == 0:
        # cdf_dwconv7_planes: the same result, also as bf16 hi / lo planes = cdf_split_bf16 of it, bit for bit
        dxp = be.empty(B, H, H, Cp)
        ph, pl = torch.zeros(B, H, H, Cp, dtype=torch.int16, device=be.device), torch.zeros(B, H, H, Cp, dtype=torch.int16, device=be.device)
        be.L.cdf_dwconv7_planes(P(dyn), Cp, P(wp), Cp, 0, 0, 0, P(dxp), Cp, B, H, H, Cp, 1, 0, P(be.to(rs)), Cp, P(ph), P(pl), Cp, be.stream())
        rh, rl = _split(be, dxp.view(-1, Cp))
        assert torch.equal(dxp.cpu(), dxr.cpu()) and torch.equal(ph.cpu().view(-1, Cp), rh.cpu()) and torch.equal(pl.cpu().view(-1, Cp), rl.cpu())
    nch = be.L.cdf_dwconv7_wgrad_nchunk(H)
    ws, dw, dbias, dsb = be.empty(B * nch * 50 * C), be.zeros(C, 1, 7, 7), be.zeros(C), be.zeros(B, Cp)
    be.L.cdf_dwconv7_wgrad(P(xn), Cp, P(dyn), Cp, P(dw), P(dbias), P(dsb), Cp, P(ws), B, H, H, C, 0, be.stream())
    assert err(y[..., :C].permute(0, 3, 1, 2), yref) <= 1e-5 and err(dx[..., :C].permute(0, 3, 1, 2), x.grad) <= 1e-5
    rel = lambda ref: 5e-5 * max(1.0, ref.abs().max().item())          # (sums over B * H * W pixels)
    assert err(dw, w.grad) <= rel(w.grad) and err(dbias, bias.grad) <= rel(bias.grad) and err(dsb[:, :C], sb.grad) <= rel(sb.grad)


@pytest.mark.parametrize("B,n", [(1, 16), (2, 64), (1, 600)])
def test_linear_attention(be, B, n):
    from einops import rearrange
    torch.manual_seed(0)
    heads, HD, scale = 4, 128, 32 ** -0.5
    qkv = torch.randn(B, n, 3 * HD, requires_grad=True)
    q, k, v = [rearrange(t_, "b n (h c) -> b h c n", h=heads) for t_ in qkv.chunk(3, dim=2)]
    q, k = q * scale, k.softmax(dim=-1)
    ctxr = torch.einsum("b h d n, b h e n -> b h d e", k, v)
    outr = rearrange(torch.einsum("b h d e, b h d n -> b h e n", ctxr, q), "b h c n -> b n (h c)")
    do = torch.randn(B, n, HD)
    outr.backward(do)
    from colddiff import ops, runtime
    saved = runtime._lib_override
    if be.kind == "emu":
        runtime._lib_override = be.L
    try:
        qd = be.to(qkv).view(B, 1, n, 3 * HD)
        out, ctx, ctxs, kmax, ksum = ops.linattn_fwd(qd, heads, scale)
        dqkv = ops.linattn_bwd(qd, be.to(do).view(B, 1, n, HD), ctx, ctxs, kmax, ksum, heads, scale)
        out, dqkv = out.view(B, n, HD), dqkv.view(B, n, 3 * HD)
    finally:
        runtime._lib_override = saved
    assert err(out, outr) <= 2e-6 and err(ctx, ctxr) <= 2e-6 and err(dqkv, qkv.grad) <= 5e-6


@pytest.mark.parametrize("B,n,heads,koff", [(2, 700, 4, 128), (1, 256, 2, 0), (3, 1030, 4, 0), (1, 37, 4, 128)])
def test_linattn_context_one_pass(be, B, n, heads, koff):
    """cdf_linattn_context, one-pass (per-tile maxima + rescaled partials) and two-pass forms, (q|k|v) and (k|v) row layouts, ragged
    last tile, and k columns whose maxima differ by ~60 between tiles (the rescaling weights span e^-60 .. 1)."""
    torch.manual_seed(n)
    HD, scale = heads * 32, 32 ** -0.5
    ld = koff + 2 * HD
    t = torch.randn(B, n, ld)
    t[:, :, koff:koff + HD] += torch.linspace(-30, 30, n).view(1, n, 1) * (torch.arange(HD) % 3 - 1).float().view(1, 1, HD)
    k, v = t[..., koff:koff + HD].double(), t[..., koff + HD:].double()
    kmax_ref = k.max(1).values
    e = torch.exp(k - kmax_ref[:, None])
    ksum_ref = e.sum(1)
    Pn = (e / ksum_ref[:, None]).view(B, n, heads, 32)
    ctx_ref = torch.einsum("bnhd,bnhe->bhde", Pn, v.view(B, n, heads, 32)).float()
    td = be.to(t)
    res = []
    try:
        for onepass in (1, 0):
            ctx, ctxs, kmax, ksum = be.empty(B, heads, 32, 32), be.empty(B, heads, 32, 32), be.empty(B, HD), be.empty(B, HD)
            ws = be.empty(be.L.cdf_linattn_ws_floats(B, n, heads))
            be.L.cdf_linattn_context(P(td), ld, koff, P(ctx), P(ctxs), P(kmax), P(ksum), P(ws), B, n, heads, scale, onepass, be.stream())
            assert torch.equal(kmax.cpu(), kmax_ref.float())
            assert err(ksum, ksum_ref.float()) <= 1e-5 * ksum_ref.max().item()
            assert err(ctx, ctx_ref) <= 1e-5 * max(1.0, ctx_ref.abs().max().item())
            assert err(ctxs, ctx_ref * scale) <= 1e-5 * max(1.0, ctx_ref.abs().max().item())
            res.append(ctx.cpu().clone())
    finally:
        pass
    assert (res[0] - res[1]).abs().max().item() <= 1e-5 * max(1.0, ctx_ref.abs().max().item())


def test_pack_many_matches_single_packs(be):
    """cdf_pack_many (one launch, device-resident descriptor table) against cdf_pack_weight / cdf_pack_weight_bf16 entry by entry: 3x3 and
    4x4 conv weights in forward and data-gradient layouts (taps contiguous in the source: the all-taps-per-thread path), a transposed-conv
    weight, a 1x1 conv and a linear layer (element-wise path), ragged channel counts, fp32 and bf16 hi / lo planes, hi only."""
    import struct
    torch.manual_seed(0)
    ents = []          # (src tensor, T, R, C, ldc, s_t, s_r, s_c, bf16, want_lo)
    def conv(Co, Ci, k, bf16, want_lo=True):
        w = torch.randn(Co, Ci, k, k)
        KK = k * k
        r32 = lambda v: (v + 31) // 32 * 32
        r4_ = lambda v: (v + 3) // 4 * 4
        if bf16:
            ents.append((w, KK, Co, Ci, r32(Ci), 1, Ci * KK, KK, True, want_lo))          # conv_fwd_sp
            ents.append((w, KK, Ci, Co, r32(Co), 1, KK, Ci * KK, True, want_lo))          # conv_dgrad_sp
        else:
            ents.append((w, KK, Ci, Co, r4_(Co), 1, KK, Ci * KK, False, False))           # conv_fwd
            ents.append((w, KK, Co, Ci, r4_(Ci), 1, Ci * KK, KK, False, False))           # conv_dgrad
    conv(40, 24, 3, True)
    conv(33, 70, 3, False)
    conv(64, 32, 4, True, want_lo=False)
    conv(48, 20, 1, True)
    conv(64, 96, 3, True)            # 32-aligned channel counts: the data-gradient layouts take the LDS-tiled path (source-contiguous reads)
    conv(128, 64, 3, False)
    conv(96, 32, 4, True)
    wl = torch.randn(50, 36)
    ents.append((wl, 1, 36, 50, 52, 0, 1, 36, False, False))                              # lin_fwd
    L, S = be.L, be.stream()
    recs, first, outs, refs = [], 0, [], []
    keep = []
    for (w, T, R, C, ldc, s_t, s_r, s_c, bf16, want_lo) in ents:
        wd = be.to(w)
        keep.append(wd)
        if bf16:
            d0 = torch.full((T, R, ldc), -1, dtype=torch.int16, device=be.device)
            d1 = torch.full((T, R, ldc), -1, dtype=torch.int16, device=be.device) if want_lo else None
            r0, r1 = torch.zeros_like(d0), (torch.zeros_like(d0) if want_lo else None)
            L.cdf_pack_weight_bf16(P(wd), P(r0), P(r1), T, R, C, ldc, s_t, s_r, s_c, S)
            outs.append((d0, d1)); refs.append((r0, r1))
        else:
            d0, d1 = be.empty(T, R, ldc), None
            r0 = be.empty(T, R, ldc)
            L.cdf_pack_weight(P(wd), P(r0), T, R, C, ldc, s_t, s_r, s_c, S)
            outs.append((d0, None)); refs.append((r0, None))
        recs.append(struct.pack("<QQQqqqiiiiii", P(wd), P(d0), P(d1), s_t, s_r, s_c, T, R, C, ldc, 1 if bf16 else 0, first))
        first += L.cdf_pack_blocks(T, R, ldc, s_t)
    blob = b"".join(recs)
    assert len(blob) == len(recs) * L.cdf_pack_entry_bytes()
    tab = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(be.device)
    L.cdf_pack_many(P(tab), len(recs), first, S)
    for (d0, d1), (r0, r1) in zip(outs, refs):
        assert torch.equal(d0.cpu(), r0.cpu())
        if d1 is not None:
            assert torch.equal(d1.cpu(), r1.cpu())


@pytest.mark.parametrize("B,n,dim,slots,split", [(2, 256, 64, 512, 3), (1, 640, 64, 1, 3), (3, 384, 128, 4, 3), (1, 256, 96, 1, 1)])
def test_linattn_kvctx_fused(be, B, n, dim, slots, split):
    """cdf_linattn_kvctx + cdf_linattn_finalize: the k | v projection and the softmax context in one pass -- against torch (kv to the
    split-precision GEMM tolerance, kmax exact for the kv it wrote, ctx / ksum from that kv), with one tile per block and with blocks that
    walk several tiles (running max + rescaled accumulators: slots hook), k columns drifting by +-20 along the pixels."""
    torch.manual_seed(n + dim)
    heads, HD, scale = 4, 128, 32 ** -0.5
    xn = torch.randn(B, n, dim)
    w = torch.randn(3 * HD, dim) / math.sqrt(dim)
    w[HD:2 * HD] *= 3.0                                    # wider k range: the per-tile maxima differ
    xn[..., 0] += torch.linspace(-6, 6, n)                 # a drift along the pixels that k picks up through column 0 of W
    kv_ref = xn.double() @ w[HD:].double().t()             # [B, n, 2 HD]
    ldk = (dim + 31) // 32 * 32
    wd = be.to(w)
    whi = torch.zeros(1, 2 * HD, ldk, dtype=torch.int16, device=be.device)
    wlo = torch.zeros_like(whi) if split == 3 else None
    be.L.cdf_pack_weight_bf16(P(wd) + 4 * HD * dim, P(whi), P(wlo), 1, 2 * HD, dim, ldk, 1, dim, 1, be.stream())
    kv = be.empty(B, n, 2 * HD)
    try:
        parts = be.L.cdf_linattn_kvctx_parts(B, n, slots)
        assert parts == min(max(slots // B, 1), n // 128) or parts >= 1
        ws = be.empty(B * parts * (2 * HD + heads * 1024))
        be.L.cdf_linattn_kvctx(P(be.to(xn)), dim, P(whi), P(wlo), ldk, P(kv), 2 * HD, P(ws), B, n, dim, heads, slots, be.stream())
    finally:
        pass
    ctx, ctxs, kmax, ksum = be.empty(B, heads, 32, 32), be.empty(B, heads, 32, 32), be.empty(B, HD), be.empty(B, HD)
    be.L.cdf_linattn_finalize(P(ws), parts, P(ctx), P(ctxs), P(kmax), P(ksum), B, heads, scale, be.stream())
    rel = 3e-5 if split == 3 else 2e-2
    assert err(kv, kv_ref.float()) <= rel * kv_ref.abs().max().item()
    kvc = kv.cpu().double()
    k, v = kvc[..., :HD], kvc[..., HD:]
    kmax_ref = k.max(1).values
    e = torch.exp(k - kmax_ref[:, None])
    ksum_ref = e.sum(1)
    ctx_ref = torch.einsum("bnhd,bnhe->bhde", (e / ksum_ref[:, None]).view(B, n, heads, 32), v.view(B, n, heads, 32)).float()
    assert torch.equal(kmax.cpu(), kmax_ref.float())
    assert err(ksum, ksum_ref.float()) <= 1e-5 * ksum_ref.max().item()
    assert err(ctx, ctx_ref) <= 2e-5 * max(1.0, ctx_ref.abs().max().item())
    assert err(ctxs, ctx_ref * scale) <= 2e-5 * max(1.0, ctx_ref.abs().max().item())


def test_small_ops(be):
    torch.manual_seed(0)
    L, S = be.L, be.stream()
    s, sc = torch.randn(3, 50, 50, requires_grad=True), 0.3
    pr = F.softmax(s * sc, dim=2)
    dp = torch.randn(3, 50, 50)
    pr.backward(dp)
    sp, dpp = torch.zeros(3, 50, 52), torch.zeros(3, 50, 52)
    sp[..., :50], dpp[..., :50] = s.detach(), dp
    p, ds = be.empty(3, 50, 52), be.empty(3, 50, 52)
    L.cdf_softmax_rows_fwd(P(be.to(sp)), P(p), 150, 50, 52, sc, S)
    L.cdf_softmax_rows_bwd(P(p), P(be.to(dpp)), P(ds), 150, 50, 52, sc, S)
    assert err(p[..., :50], pr) <= 1e-6 and err(ds[..., :50], s.grad) <= 1e-6
    t = torch.tensor([0, 5, 199, 999])
    out = be.empty(4, 64)
    e = torch.exp(torch.arange(32) * -(math.log(10000) / 31))
    L.cdf_sinusoidal(P(be.to(t)), P(be.to(e)), P(out), 64, 4, 64, S)
    e = t[:, None] * e[None, :]
    assert err(out, torch.cat((e.sin(), e.cos()), -1)) <= 5e-6
    x = torch.randn(5, 7, requires_grad=True)
    for act, fn in ((1, F.gelu), (2, F.silu)):
        y = be.empty(5, 8)
        L.cdf_act_fwd(P(be.to(x)), 7, P(y), 8, 5, 7, act, S)
        yr, dy = fn(x), torch.randn(5, 7)
        x.grad = None
        yr.backward(dy)
        dx = be.empty(5, 7)
        L.cdf_act_bwd(P(be.to(x)), 7, P(be.to(dy)), 7, P(dx), 7, 5, 7, act, 0, S)
        assert err(y[:, :7], yr) <= 1e-6 and err(dx, x.grad) <= 1e-6
    xx = torch.randn(2, 3, 5, 8)
    y, dx = be.empty(2, 6, 10, 8), be.empty(2, 3, 5, 8)
    L.cdf_upsample2(P(be.to(xx)), 8, P(y), 8, 2, 3, 5, 8, S)
    L.cdf_upsample2_bwd(P(y), 8, P(dx), 8, 2, 3, 5, 8, 0, S)
    assert err(y, F.interpolate(xx.permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1)) == 0.0
    assert err(dx, 4 * xx) <= 1e-6
    ones, y = be.to(torch.ones(100, 10)), be.empty(100, 10)
    L.cdf_dropout(P(ones), 10, P(y), 10, 100, 10, 0.1, 1234, S)
    assert 0.85 <= (y > 0).float().mean().item() <= 0.95 and abs(y.max().item() - 1 / 0.9) < 1e-6
    d = be.to(torch.ones(4, 6))
    L.cdf_axpby(P(d), 6, P(be.to(torch.full((4, 6), 2.0))), 6, 4, 6, 1.0, 0.5, S)
    assert err(d, torch.full((4, 6), 2.0)) == 0.0


def test_adam_ema_match_torch(be):
    """cdf_adam_step / cdf_ema_update reproduce torch.optim.Adam (defaults) and EMA bit for bit."""
    torch.manual_seed(0)
    n = 1000
    p = torch.randn(n)
    p0 = p.clone().requires_grad_()
    opt = torch.optim.Adam([p0], lr=2e-5)
    pd, m, v = be.to(p), be.zeros(n), be.zeros(n)
    for step in range(1, 5):
        g = torch.randn(n)
        p0.grad = g.clone()
        opt.step()
        be.L.cdf_adam_step(P(pd), P(be.to(g)), P(m), P(v), n, 2e-5, 0.9, 0.999, 1e-8, step, be.stream())
        assert err(pd, p0.detach()) <= 1e-9
    ma = torch.randn(n)
    mad = be.to(ma)
    be.L.cdf_ema_update(P(mad), P(pd), n, 0.995, be.stream())
    assert err(mad, ma * 0.995 + (1 - 0.995) * pd.cpu()) <= 1e-9


# ---------------------------------------------------------------------------------------------
# split-precision bf16 MFMA conv (bf16x3: fp32-grade parity; bf16: plain)
# ---------------------------------------------------------------------------------------------
SP_CASES = [(2, 32, 40, 8, 3, 1, 1, False), (1, 64, 32, 8, 1, 1, 0, False), (1, 36, 130, 8, 3, 1, 1, False), (1, 32, 32, 8, 4, 2, 1, False),
            (1, 32, 32, 4, 4, 2, 1, True), (1, 96, 160, 8, 1, 1, 0, False)]      # (last: 3 and 5 K steps -- both parities of the two-step lookahead)
SP_CASES_GPU = [(4, 64, 128, 32, 3, 1, 1, False), (2, 128, 64, 32, 3, 1, 1, False), (2, 64, 64, 32, 4, 2, 1, False),
                (2, 64, 64, 16, 4, 2, 1, True), (2, 256, 512, 16, 3, 1, 1, False), (3, 64, 384, 32, 1, 1, 0, False)]


def _sp_case(be, split, B, Cin, Cout, H, k, s, p, transposed):
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, H, requires_grad=True)
    wshape = (Cin, Cout, k, k) if transposed else (Cout, Cin, k, k)
    w = (torch.randn(*wshape) * (1.0 / math.sqrt(Cin * k * k))).requires_grad_()
    bias = torch.randn(Cout)
    yref = F.conv_transpose2d(x, w, bias, stride=s, padding=p) if transposed else F.conv2d(x, w, bias, stride=s, padding=p)
    gy = torch.randn_like(yref)
    yref.backward(gy)
    KK = k * k

    def pack_sp(N, K, s_n, s_k):
        ldk = (K + 31) // 32 * 32
        hi = torch.empty(KK, N, ldk, dtype=torch.int16, device=be.device)
        lo = torch.empty(KK, N, ldk, dtype=torch.int16, device=be.device)
        be.L.cdf_pack_weight_bf16(P(wd_), P(hi), P(lo), KK, N, K, ldk, 1, s_n, s_k, be.stream())
        be._keep += [hi, lo]
        return hi, lo

    wd_ = be.to(w)
    if not transposed:
        plan, pd = cd.conv_fwd(H, H, k, k, s, p, p, p, p), cd.conv_dgrad(H, H, k, k, s, p, p, p, p)
        wf, wb = pack_sp(Cout, Cin, Cin * KK, KK), pack_sp(Cin, Cout, KK, Cin * KK)
    else:
        plan, pd = cd.convT_fwd(H, H, k, k, s, p), cd.convT_dgrad(H, H, k, k, s, p)
        wf, wb = pack_sp(Cout, Cin, KK, Cout * KK), pack_sp(Cin, Cout, Cout * KK, KK)

    def run(pl, xin, wpair, Ci, Co, bias_):
        y = be.zeros(B, pl.OH, pl.OW, r4(Co))
        be.L.cdf_conv_gemm_bf16(P(xin), xin.shape[-1], P(wpair[0]), P(wpair[1]), wpair[0].shape[-1], P(y), y.shape[-1], B, pl.H, pl.W, Ci,
                                pl.OH, pl.OW, Co, pl.QH, pl.QW, pl.os, pl.istride, pl.nphase, pl.desc, P(bias_), 0, 0, 0, 0, 0, 0, 0, 0,
                                0, 0, 0, split, be.stream())
        return y

    y = run(plan, be.to(nhwc(x)), wf, Cin, Cout, be.to(bias))
    dx = run(pd, be.to(nhwc(gy)), wb, Cout, Cin, None)
    if split == 3 and Cin % 4 == 0:               # weight gradient on the split-precision path (+ fused bias gradient)
        wg = cd.convT_wgrad(H, H, k, k, s, p) if transposed else cd.conv_wgrad(H, H, k, k, s, p, p, p, p)
        M = B * wg.QH * wg.QW
        ns, ldo = max(1, min(3, M // 32)), r4(Cout)
        ws, bsum = be.empty(ns, KK, Cin, ldo), be.empty(ns, ldo)
        xn, gyn = be.to(nhwc(x)), be.to(nhwc(gy))
        be.L.cdf_conv_wgrad_bf16(P(xn), xn.shape[-1], P(gyn), gyn.shape[-1], P(ws), ldo, B, wg.QH, wg.QW, wg.HA, wg.WA, wg.sa, wg.HB, wg.WB,
                                 wg.sb, Cin, Cout, wg.ntaps, wg.desc, ns, 0 if transposed else P(bsum), be.stream())
        dw = be.zeros(*wshape)
        s_r, s_c = (Cout * KK, KK) if transposed else (KK, Cin * KK)
        be.L.cdf_unpack_reduce(P(ws), P(dw), ns, KK, Cin, Cout, ldo, 1, s_r, s_c, 0, 1, be.stream())
        assert err(dw, w.grad) <= 3e-5 * max(1.0, w.grad.abs().max().item()) * math.sqrt(M / 16)
        if not transposed:
            db = be.zeros(Cout)
            be.L.cdf_unpack_reduce(P(bsum), P(db), ns, 1, 1, Cout, ldo, 0, 0, 1, 0, 1, be.stream())
            assert err(db, gy.sum((0, 2, 3))) <= 1e-5 * max(1.0, gy.sum((0, 2, 3)).abs().max().item()) * math.sqrt(M)
    rel = 3e-5 if split == 3 else 2e-2            # bf16x3 keeps 16 mantissa bits per operand; bf16 keeps 8
    tol = lambda ref: rel * max(1.0, ref.abs().max().item())
    assert err(y[..., :Cout].permute(0, 3, 1, 2), yref) <= tol(yref)
    assert err(dx[..., :Cin].permute(0, 3, 1, 2), x.grad) <= tol(x.grad)


@pytest.mark.parametrize("split", [3, 1])
@pytest.mark.parametrize("case", SP_CASES)
def test_conv_gemm_bf16(be, split, case):
    _sp_case(be, split, *case)


@pytest.mark.gpu
@pytest.mark.parametrize("split", [3, 1])
@pytest.mark.parametrize("case", SP_CASES_GPU)
def test_conv_gemm_bf16_large(split, case):
    from conftest import Backend
    _sp_case(Backend("hip"), split, *case)


# ---------------------------------------------------------------------------------------------
# pre-split operand variants (activation split once by cdf_split_bf16)
# ---------------------------------------------------------------------------------------------
SPX_CASES = [(2, 32, 40, 8, 3, 1, 1), (1, 64, 32, 8, 1, 1, 0), (1, 40, 136, 8, 3, 1, 1), (1, 32, 32, 8, 4, 2, 1),
             (1, 72, 24, 8, 3, 1, 1), (1, 136, 72, 6, 3, 1, 1)]   # wgrad tiles 64x64, 64x128, 128x64, 128x128
SPX_CASES_GPU = [(4, 64, 128, 32, 3, 1, 1), (2, 128, 64, 32, 3, 1, 1), (2, 256, 512, 16, 3, 1, 1), (2, 64, 64, 32, 4, 2, 1),
                 (4, 64, 128, 128, 3, 1, 1), (4, 128, 64, 128, 3, 1, 1), (2, 64, 96, 64, 3, 1, 1)]   # 128 / 64-wide images: LDS-resident input tiles


def _split(be, t, single=False):
    """fp32 NHWC [.., C] -> (hi, lo) bf16 planes with pitch roundup8(C); single: hi only (lo = None, single-pass bf16 operands)."""
    C = t.shape[-1]
    ld = (C + 7) // 8 * 8
    rows = t.numel() // C
    hi = torch.zeros(t.shape[:-1] + (ld,), dtype=torch.int16, device=be.device)
    lo = None if single else torch.zeros_like(hi)
    be.L.cdf_split_bf16(P(t), C, P(hi), P(lo), ld, rows, C, be.stream())
    be._keep += [hi, lo]
    return hi, lo


SPX_SINGLE = False      # module switch: run _spx_case with hi-only planes (the "bf16" arithmetic mode: NS = 1 kernel instantiations)


def _spx_case(be, B, Cin, Cout, H, k, s, p):
    single = SPX_SINGLE
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, H, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k) * (1.0 / math.sqrt(Cin * k * k))).requires_grad_()
    bias = torch.randn(Cout)
    yref = F.conv2d(x, w, bias, stride=s, padding=p)
    gy = torch.randn_like(yref)
    yref.backward(gy)
    if single:
        # reference of the single-pass mode: the SAME fp32 arithmetic on operands rounded to bf16 (round to nearest even), so the
        # tolerances below stay at accumulation-order level
        bf = lambda t: t.detach().bfloat16().float()
        xq, wq, gq = bf(x).requires_grad_(), bf(w).requires_grad_(), bf(gy)
        yref = F.conv2d(xq, wq, bias, stride=s, padding=p)
        yref.backward(gq)
        dx_ref, dw_ref, db_ref = xq.grad, wq.grad, gq.sum((0, 2, 3))
    else:
        dx_ref, dw_ref, db_ref = x.grad, w.grad, gy.sum((0, 2, 3))
    yref = yref.detach()
    KK = k * k
    zero = be.zeros(16)
    wd_ = be.to(w)

    def pack_sp(N, K, s_n, s_k):
        ldk = (K + 31) // 32 * 32
        hi = torch.empty(KK, N, ldk, dtype=torch.int16, device=be.device)
        lo = None if single else torch.empty(KK, N, ldk, dtype=torch.int16, device=be.device)
        be.L.cdf_pack_weight_bf16(P(wd_), P(hi), P(lo), KK, N, K, ldk, 1, s_n, s_k, be.stream())
        be._keep += [hi, lo]
        return hi, lo

    plan, pd, wg = cd.conv_fwd(H, H, k, k, s, p, p, p, p), cd.conv_dgrad(H, H, k, k, s, p, p, p, p), cd.conv_wgrad(H, H, k, k, s, p, p, p, p)
    wf, wb = pack_sp(Cout, Cin, Cin * KK, KK), pack_sp(Cin, Cout, KK, Cin * KK)
    xn = torch.zeros(B, H, H, Cin)
    xn[...] = x.detach().permute(0, 2, 3, 1)
    gyn = torch.zeros(B, plan.OH, plan.OW, Cout)
    gyn[...] = gy.permute(0, 2, 3, 1)
    xs, gs = _split(be, be.to(xn), single), _split(be, be.to(gyn), single)

    def run(pl, xsplit, wpair, Ci, Co, bias_):
        y = be.zeros(B, pl.OH, pl.OW, r4(Co))
        be.L.cdf_conv_gemm_bf16x(P(xsplit[0]), P(xsplit[1]), xsplit[0].shape[-1], P(zero), P(wpair[0]), P(wpair[1]), wpair[0].shape[-1], P(y),
                                 y.shape[-1], B, pl.H, pl.W, Ci, pl.OH, pl.OW, Co, pl.QH, pl.QW, pl.os, pl.istride, pl.nphase, pl.desc,
                                 P(bias_), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, be.tune.ptr, be.stream())
        # small grids: the same launch with a split-K workspace (taps shared out over block groups + finish kernel)
        Mq = B * pl.QH * pl.QW
        try:
            ks = be.L.cdf_conv_gemm_bf16x_ksplit(Mq, Co, pl.nphase, pl.desc[2], be.tune.ptr)
            if ks > 1:
                ws = be.empty(ks * Mq * r4(Co))
                y_ws = be.zeros(B, pl.OH, pl.OW, r4(Co))
                be.L.cdf_conv_gemm_bf16x(P(xsplit[0]), P(xsplit[1]), xsplit[0].shape[-1], P(zero), P(wpair[0]), P(wpair[1]), wpair[0].shape[-1],
                                         P(y_ws), y_ws.shape[-1], B, pl.H, pl.W, Ci, pl.OH, pl.OW, Co, pl.QH, pl.QW, pl.os, pl.istride, pl.nphase,
                                         pl.desc, P(bias_), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, P(ws), ks * Mq * r4(Co), be.tune.ptr, be.stream())
                assert err(y_ws[..., :Co], y[..., :Co].cpu()) <= 2e-6 * max(1.0, y.abs().max().item()) * math.sqrt(ks)
                return y_ws
        finally:
            pass
        return y

    y = run(plan, xs, wf, Cin, Cout, be.to(bias))
    dx = run(pd, gs, wb, Cout, Cin, None)
    if Cout % 4 == 0:
        # the epilogue's fused operand split must equal cdf_split_bf16 of the stored output, bit for bit
        ld8 = (Cout + 7) // 8 * 8
        yh = torch.zeros(B, plan.OH, plan.OW, ld8, dtype=torch.int16, device=be.device)
        yl = None if single else torch.zeros_like(yh)
        y2 = be.zeros(B, plan.OH, plan.OW, r4(Cout))
        be.L.cdf_conv_gemm_bf16x(P(xs[0]), P(xs[1]), xs[0].shape[-1], P(zero), P(wf[0]), P(wf[1]), wf[0].shape[-1], P(y2), y2.shape[-1], B,
                                 plan.H, plan.W, Cin, plan.OH, plan.OW, Cout, plan.QH, plan.QW, plan.os, plan.istride, plan.nphase,
                                 plan.desc, P(be.to(bias)), 0, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, P(yh), P(yl), ld8, 0, 0, be.tune.ptr, be.stream())
        rh, rl = _split(be, y2[..., :Cout].contiguous(), single)
        assert torch.equal(yh.cpu(), rh.cpu()) and (single or torch.equal(yl.cpu(), rl.cpu()))
    M = B * wg.QH * wg.QW
    ns, ldo = max(1, min(3, M // 32)), r4(Cout)
    ws, bsum = be.empty(ns, KK, Cin, ldo), be.empty(ns, ldo)
    be.L.cdf_conv_wgrad_bf16x(P(xs[0]), P(xs[1]), xs[0].shape[-1], P(gs[0]), P(gs[1]), gs[0].shape[-1], P(zero), P(ws), ldo, B, wg.QH, wg.QW,
                              wg.HA, wg.WA, wg.sa, wg.HB, wg.WB, wg.sb, Cin, Cout, wg.ntaps, wg.desc, ns, P(bsum), be.tune.ptr, be.stream())
    dw, db = be.zeros(Cout, Cin, k, k), be.zeros(Cout)
    be.L.cdf_unpack_reduce(P(ws), P(dw), ns, KK, Cin, Cout, ldo, 1, KK, Cin * KK, 0, 1, be.stream())
    be.L.cdf_unpack_reduce(P(bsum), P(db), ns, 1, 1, Cout, ldo, 0, 0, 1, 0, 1, be.stream())
    tol = lambda ref: 3e-5 * max(1.0, ref.abs().max().item())
    assert err(y[..., :Cout].permute(0, 3, 1, 2), yref) <= tol(yref)
    assert err(dx[..., :Cin].permute(0, 3, 1, 2), dx_ref) <= tol(dx_ref)
    assert err(dw, dw_ref) <= 3e-5 * max(1.0, dw_ref.abs().max().item()) * math.sqrt(M / 16)
    assert err(db, db_ref) <= 3e-5 * max(1.0, db_ref.abs().max().item()) * math.sqrt(M)


@pytest.mark.parametrize("case", SPX_CASES)
def test_conv_presplit(be, case):
    _spx_case(be, *case)


@pytest.mark.parametrize("tile", [(256, 128), (128, 128), (128, 64), (64, 128), (64, 64)])
def test_conv_presplit_forced_tiles(be, tile):
    """Every block-tile instantiation of the pre-split GEMM on a shape with ragged M and N tiles (the automatic choice would only
    ever pick the 64-row tiles at emulator-sized problems)."""
    be.tune.set(tile_bm=tile[0], tile_bn=tile[1])
    try:
        _spx_case(be, 3, 40, 72, 7, 3, 1, 1)
    finally:
        be.tune.set(tile_bm=0, tile_bn=0)


@pytest.mark.parametrize("case", [(1, 64, 96, 16, 3, 1, 1), (2, 96, 40, 16, 3, 1, 1), (1, 64, 72, 32, 3, 1, 1)])
def test_conv_presplit_halo(be, case):
    """3 x 3 stride-1 layers with the input tile resident in LDS (conv_igemm_halo_kernel): 128-pixel strips of 16- and 32-wide
    images, 128- and 64-wide N tiles, several channel chunks, forward taps and the mirrored taps of the data gradient; the
    generic kernel must give the same numbers to rounding."""
    be.tune.set(halo=31, halo_min_tiles=1)
    try:
        for bm in (128, 256):                         # both tile heights of the LDS-resident kernel
            be.tune.set(halo_bm=bm)
            _spx_case(be, *case)
        be.tune.set(halo=0, halo_min_tiles=1)
        _spx_case(be, *case)
    finally:
        be.tune.set(halo=47, halo_min_tiles=1)
        be.tune.set(halo_bm=0)


def test_conv_presplit_halo_small_grid_n64(be):
    """Small grids (sampling batches, the 16 x 16 level): the LDS-resident-input kernel takes 64-wide N tiles for layers with MORE than
    64 output channels when 128-wide ones would leave most CUs idle (cdf_gemm_tuning.small_n64).  192 and 128 output channels =
    3 / 2 column tiles of 64 (forward / data gradient); the 128-wide choice must give the same numbers to rounding."""
    case = (1, 128, 192, 16, 3, 1, 1)
    _spx_case(be, *case)
    be.tune.set(small_n64=0)
    try:
        _spx_case(be, *case)
    finally:
        be.tune.set(small_n64=1)


@pytest.mark.parametrize("case", [(1, 136, 72, 16, 3, 1, 1), (2, 64, 136, 16, 3, 1, 1), (1, 136, 40, 32, 3, 1, 1)])
def test_wgrad_presplit_row_of_taps(be, case):
    """Weight gradient of 3 x 3 same-size convolutions by one block per row of taps (conv_wgrad_row3_kernel): 16-wide images
    (a chunk spans two image rows) and 32-wide ones, the three tile shapes 128x128 / 64x128 / 128x64 with ragged channel tiles,
    several splits; and the per-tap kernel on the same data."""
    _spx_case(be, *case)
    be.tune.set(wgrad_row3=0)
    try:
        _spx_case(be, *case)
    finally:
        be.tune.set(wgrad_row3=1)


@pytest.mark.parametrize("case", [(1, 64, 96, 16, 3, 1, 1), (2, 96, 40, 16, 3, 1, 1), (1, 64, 72, 32, 3, 1, 1), (1, 128, 72, 16, 3, 1, 1),
                                  (3, 64, 40, 32, 3, 1, 1), (5, 64, 136, 16, 3, 1, 1), (9, 128, 40, 16, 3, 1, 1)])
def test_conv_presplit_rowhalo_emu(case):
    """Row-halo stream form of the 3 x 3 GEMM (256-pixel tiles, input shared by the dx taps of a row, resident blocks) forced at every
    width (bit 64 of the halo hook; by default it serves the 128-pixel layers, which the GPU cases cover).  It takes 64 / 128 input
    channels (2 / 4 chunks per tap row, K loop fully unrolled); the simulator has 8 "CUs", so the last two cases (12 and 10 tiles) have
    blocks with two tiles and with one.  96 input channels, and every case again with rowhalo_stream = 0, fall through to the
    LDS-resident-input kernel (same sums).  Simulator only, to keep the GPU suite short."""
    from conftest import Backend
    be = Backend("emu")
    be.tune.set(halo=64 | 47, halo_min_tiles=1)
    try:
        _spx_case(be, *case)
        if case[1] in (64, 128):
            be.tune.set(rowhalo_stream=0)
            _spx_case(be, *case)
    finally:
        be.tune.set(halo=47, halo_min_tiles=1, rowhalo_stream=1)
        be._keep.clear()


@pytest.mark.parametrize("reserve", [5, 32])
def test_conv_presplit_rowhalo_resident_reserve_emu(reserve):
    """cdf_gemm_tuning.resident_reserve: the resident row-halo blocks leave CUs free for concurrent kernels (multi-rank training: the
    gradient exchange's collectives) -- the grid every multi-rank step launches (224 of 256 blocks on the MI355X).  The simulator has
    8 "CUs" and takes the reserve literally: 3 blocks (reserve 5) and ONE block (reserve 32) walk the 10 tiles, i.e. several tiles per
    resident block with the operand stream running across tile boundaries; same sums, checked against the fp32 reference."""
    from conftest import Backend
    be = Backend("emu")
    be.tune.set(halo=64 | 47, halo_min_tiles=1, resident_reserve=reserve)
    try:
        _spx_case(be, 5, 64, 136, 16, 3, 1, 1)
    finally:
        be.tune.set(halo=47, halo_min_tiles=1, resident_reserve=0)
        be._keep.clear()


def test_conv_presplit_row_tiles(be):
    """A tile that is exactly one image row (W = 64 with the 64-row tile): the 3 x 3 taps run in a per-tile row-group
    order -- every tap must still be taken exactly once, forward and data gradient."""
    be.tune.set(tile_bm=64, tile_bn=64)
    try:
        _spx_case(be, 1, 8, 16, 64, 3, 1, 1)
    finally:
        be.tune.set(tile_bm=0, tile_bn=0)


@pytest.mark.gpu
@pytest.mark.parametrize("case", SPX_CASES_GPU)
def test_conv_presplit_large(case):
    from conftest import Backend
    be = Backend("hip")
    _spx_case(be, *case)
    be.tune.set(halo=31, halo_min_tiles=1)          # LDS-resident input tiles at every width (128 is off by default), both tile heights
    try:
        for bm in (128, 256):
            be.tune.set(halo_bm=bm)
            _spx_case(be, *case)
    finally:
        be.tune.set(halo=47, halo_min_tiles=1)
        be.tune.set(halo_bm=0)


@pytest.mark.parametrize("ns", [1, 3, 7, 31, 32, 37, 64, 70, 227])
def test_unpack_reduce(be, ns):
    """Slab reduction + scatter into the PyTorch weight layout, both lane counts (4 below 32 slabs, 16 from there) and every
    tail length of the four-chain loop; accumulate on top of an existing gradient."""
    torch.manual_seed(ns)
    for C in (13, 12):                                   # 13: scalar loads; 12 (C % 4 == 0): the 16-byte-load form
        T, R, ldc = 9, 5, 16
        ws = torch.randn(ns, T, R, ldc)
        g0 = torch.randn(C, R, T)
        g = be.to(g0)
        be.L.cdf_unpack_reduce(P(be.to(ws)), P(g), ns, T, R, C, ldc, 1, T, R * T, 1, 1, be.stream())
        ref = g0 + ws[..., :C].double().sum(0).permute(2, 1, 0).float()
        assert err(g, ref) <= 2e-6 * math.sqrt(ns) * max(1.0, ref.abs().max().item())
        be.L.cdf_unpack_reduce(P(be.to(ws)), P(g), ns, T, R, C, ldc, 1, T, R * T, 0, 1, be.stream())
        assert err(g, ref - g0) <= 2e-6 * math.sqrt(ns) * max(1.0, ref.abs().max().item())
        # with the bias-partial reduction folded into the same launch
        bws, gb0 = torch.randn(ns, ldc), torch.randn(C)
        g, gb = be.to(g0), be.to(gb0)
        be.L.cdf_unpack_reduce_bias(P(be.to(ws)), P(g), ns, T, R, C, ldc, 1, T, R * T, P(be.to(bws)), P(gb), ldc, 1, 1, be.stream())
        assert err(g, ref) <= 2e-6 * math.sqrt(ns) * max(1.0, ref.abs().max().item())
        assert err(gb, gb0 + bws[:, :C].double().sum(0).float()) <= 2e-6 * math.sqrt(ns) * 4


@pytest.mark.parametrize("T,R,C,conv_t", [(9, 10, 70, False), (9, 3, 128, False), (16, 5, 33, False), (16, 7, 64, True), (1, 45, 100, False)])
@pytest.mark.parametrize("ns", [1, 5, 19])
def test_unpack_reduce_tiled(be, T, R, C, conv_t, ns):
    """The LDS-tiled transposing form of the slab reduction (conv weights: the slab's fast index c is the layout's slowest): ragged
    tiles in c and r, 3x3 / 4x4 / 1x1 tap counts, transposed-conv strides, fused bias row, accumulate on and off -- against the
    element-wise kernel's contract and against that kernel itself (hook off)."""
    torch.manual_seed(ns + T)
    ldc = (C + 3) // 4 * 4
    ws = torch.randn(ns, T, R, ldc)
    if conv_t:                                           # weight [R = Cin][C = Cout][T]: s_r = C T, s_c = T
        g0, s_r, s_c = torch.randn(R, C, T), C * T, T
        ref_add = ws[..., :C].double().sum(0).permute(1, 2, 0).float()
    else:                                                # weight [C = Cout][R = Cin][T]
        g0, s_r, s_c = torch.randn(C, R, T), T, R * T
        ref_add = ws[..., :C].double().sum(0).permute(2, 1, 0).float()
    bws, gb0 = torch.randn(ns, ldc), torch.randn(C)
    tol = 2e-6 * math.sqrt(ns) * max(1.0, (g0 + ref_add).abs().max().item())
    outs = []
    if True:
        for tiled in (1, 1, 0):
            g, gb = be.to(g0), be.to(gb0)
            be.L.cdf_unpack_reduce_bias(P(be.to(ws)), P(g), ns, T, R, C, ldc, 1, s_r, s_c, P(be.to(bws)), P(gb), ldc, 1, tiled, be.stream())
            assert err(g, g0 + ref_add) <= tol
            assert err(gb, gb0 + bws[:, :C].double().sum(0).float()) <= 2e-6 * math.sqrt(ns) * 4
            g2 = be.to(g0)
            be.L.cdf_unpack_reduce(P(be.to(ws)), P(g2), ns, T, R, C, ldc, 1, s_r, s_c, 0, tiled, be.stream())
            assert err(g2, ref_add) <= tol
            outs.append(g2.detach().cpu().clone())
    assert (outs[0] - outs[2]).abs().max().item() <= tol and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,K,N", [(32, 64, 256), (5, 256, 40), (70, 48, 130)])
def test_linear_small(be, M, K, N):
    """Skinny linear layer kernels against torch: forward (packed [K][N] weight), data gradient (PyTorch [N][K] weight),
    accumulating weight / bias gradients."""
    torch.manual_seed(0)
    x = torch.randn(M, K, requires_grad=True)
    lin = torch.nn.Linear(K, N)
    y = lin(x)
    g = torch.randn(M, N)
    y.backward(g)
    Np = r4(N)
    wp = be.empty(1, K, Np)
    be.L.cdf_pack_weight(P(be.to(lin.weight.detach())), P(wp), 1, K, N, Np, 0, 1, K, be.stream())
    xd, gd, Wd, bd = be.to(x.detach()), be.to(g), be.to(lin.weight.detach()), be.to(lin.bias.detach())
    yo = be.empty(M, Np)
    be.L.cdf_linear_small(P(xd), K, P(wp), Np, P(bd), P(yo), Np, M, K, N, be.stream())
    assert err(yo[:, :N], y.detach()) <= 2e-5 and (yo[:, N:].cpu() == 0).all()
    Kp = r4(K)
    dx = be.empty(M, Kp)
    be.L.cdf_linear_small(P(gd), N, P(Wd), K, 0, P(dx), Kp, M, N, K, be.stream())
    assert err(dx[:, :K], x.grad) <= 2e-5
    dW, db = be.to(torch.ones(N, K)), be.to(torch.ones(N))
    be.L.cdf_linear_small_wgrad(P(gd), N, P(xd), K, P(dW), P(db), M, N, K, be.stream())
    assert err(dW, lin.weight.grad + 1) <= 5e-5 and err(db, lin.bias.grad + 1) <= 5e-5


@pytest.mark.parametrize("B,H,Cin,Cout,k,act", [(2, 12, 3, 128, 3, 1), (1, 9, 3, 64, 1, 0), (3, 8, 1, 32, 3, 2), (1, 20, 4, 256, 3, 0)])
def test_conv_cin4_direct(be, B, H, Cin, Cout, k, act):
    """Direct convolution for <= 4 input channels (image-side convs, DEBLUR:145-165 with dim = channels): forward with fused
    bias / activation / pre-activation / bf16 planes, data gradient, weight + bias gradient against torch."""
    torch.manual_seed(0)
    x = torch.randn(B, Cin, H, H, requires_grad=True)
    conv = torch.nn.Conv2d(Cin, Cout, k, padding=k // 2)
    pre_ref = conv(x)
    yref = F.gelu(pre_ref) if act == 1 else (F.silu(pre_ref) if act == 2 else pre_ref)
    g = torch.randn_like(pre_ref)
    pre_ref.backward(g)                                      # gradients w.r.t. the pre-activation (the activation's own
    KK = k * k                                               # derivative is fused elsewhere)
    xn = torch.zeros(B, H, H, 4)
    xn[..., :Cin] = x.detach().permute(0, 2, 3, 1)
    xd, gd = be.to(xn), be.to(g.permute(0, 2, 3, 1).contiguous())
    wp = be.empty(KK, 4, Cout)
    be.L.cdf_pack_cin4(P(be.to(conv.weight.detach())), P(wp), Cout, Cout, Cin, k, be.stream())
    y, pre = be.empty(B, H, H, Cout), be.empty(B, H, H, Cout)
    yh = torch.zeros(B, H, H, Cout, dtype=torch.int16, device=be.device)
    yl = torch.zeros_like(yh)
    be.L.cdf_conv_cin4_fwd(P(xd), P(wp), Cout, P(be.to(conv.bias.detach())), P(y), Cout, P(pre), Cout, P(yh), P(yl), Cout, B, H, H, Cout, k, act,
                           be.stream())
    assert err(y.permute(0, 3, 1, 2), yref.detach()) <= 2e-5 and err(pre.permute(0, 3, 1, 2), pre_ref.detach()) <= 2e-5
    rh, rl = _split(be, y)
    assert torch.equal(yh.cpu(), rh.cpu()) and torch.equal(yl.cpu(), rl.cpu())
    dx = be.empty(B, H, H, 4)
    be.L.cdf_conv_cin4_dgrad(P(gd), Cout, P(wp), Cout, P(dx), B, H, H, Cout, k, 0, be.stream())
    assert err(dx[..., :Cin].permute(0, 3, 1, 2), x.grad) <= 5e-5 * max(1.0, x.grad.abs().max().item())
    if k == 3:
        # the two-stage form: z = dY . W[Cout][9 Cin] per pixel (torch here; cdf_conv_gemm in the package), then the tap sum
        ldz = (9 * Cin + 3) // 4 * 4
        z = torch.zeros(B, H, H, ldz)
        z[..., :9 * Cin] = g.permute(0, 2, 3, 1).reshape(-1, Cout) .matmul(conv.weight.detach().reshape(Cout, 9 * Cin)).reshape(B, H, H, 9 * Cin)
        dx2 = be.to(torch.full((B, H, H, 4), 7.0))
        be.L.cdf_conv_cin4_tapsum3(P(be.to(z)), ldz, P(dx2), B, H, H, Cin, 0, be.stream())
        assert err(dx2[..., :Cin].permute(0, 3, 1, 2), x.grad) <= 5e-5 * max(1.0, x.grad.abs().max().item())
        assert dx2[..., Cin:].abs().max().item() == 0.0 if Cin < 4 else True
        be.L.cdf_conv_cin4_tapsum3(P(be.to(z)), ldz, P(dx2), B, H, H, Cin, 1, be.stream())
        assert err(dx2[..., :Cin].permute(0, 3, 1, 2), 2 * x.grad) <= 1e-4 * max(1.0, x.grad.abs().max().item())
    nch = be.L.cdf_conv_cin4_nchunk(B * H * H)
    part, bsum = be.empty(nch, KK * Cin, Cout), be.empty(nch, Cout)
    be.L.cdf_conv_cin4_wgrad(P(xd), P(gd), Cout, P(part), P(bsum), B, H, H, Cin, Cout, k, be.stream())
    dw, db = be.zeros(Cout, Cin, k, k), be.zeros(Cout)
    be.L.cdf_unpack_reduce(P(part), P(dw), nch, KK, Cin, Cout, Cout, 1, KK, Cin * KK, 0, 1, be.stream())
    be.L.cdf_unpack_reduce(P(bsum), P(db), nch, 1, 1, Cout, Cout, 0, 0, 1, 0, 1, be.stream())
    assert err(dw, conv.weight.grad) <= 5e-5 * max(1.0, conv.weight.grad.abs().max().item())
    assert err(db, conv.bias.grad) <= 5e-5 * max(1.0, conv.bias.grad.abs().max().item())


@pytest.fixture
def single_pass():
    """Run the pre-split cases with hi-only planes: the NS = 1 instantiations ("bf16" mode: one MFMA per product)."""
    global SPX_SINGLE
    SPX_SINGLE = True
    yield
    SPX_SINGLE = False


@pytest.mark.parametrize("case", SPX_CASES)
def test_conv_presplit_single_pass(be, single_pass, case):
    _spx_case(be, *case)


@pytest.mark.parametrize("case", [(1, 64, 96, 16, 3, 1, 1), (2, 96, 40, 16, 3, 1, 1)])
def test_conv_presplit_halo_single_pass(be, single_pass, case):
    test_conv_presplit_halo(be, case)


@pytest.mark.parametrize("case", [(1, 136, 72, 16, 3, 1, 1), (2, 64, 136, 16, 3, 1, 1)])
def test_wgrad_presplit_row_of_taps_single_pass(be, single_pass, case):
    test_wgrad_presplit_row_of_taps(be, case)


def test_conv_presplit_rowhalo_emu_single_pass(single_pass):
    test_conv_presplit_rowhalo_emu((2, 96, 40, 16, 3, 1, 1))


@pytest.mark.gpu
@pytest.mark.parametrize("case", SPX_CASES_GPU)
def test_conv_presplit_large_single_pass(single_pass, case):
    test_conv_presplit_large(case)


@pytest.mark.parametrize("B,n,heads", [(1, 32, 4), (2, 70, 4), (1, 300, 2)])
def test_linattn_bwd_kv_fused(be, B, n, heads):
    """The one-pass k / v backward of linear attention (P recomputed, dP and dv on the fp32 matrix cores) against torch:
    ragged pixel counts (not a multiple of the 32-pixel tile), 2 and 4 heads."""
    torch.manual_seed(n)
    HD = heads * 32
    qkv = torch.randn(B, n, 3 * HD)
    k, v = qkv[..., HD:2 * HD], qkv[..., 2 * HD:]
    kmax = k.max(1).values                                              # [B, HD]
    ksum = torch.exp(k - kmax[:, None]).sum(1)
    Pn = torch.exp(k - kmax[:, None]) / ksum[:, None]                   # softmax over n
    dctx = torch.randn(B, heads, 32, 32) * 0.3
    ctx = torch.randn(B, heads, 32, 32)
    rvec = (dctx * ctx).sum(-1).reshape(B, HD)
    Ph, vh = Pn.view(B, n, heads, 32), v.reshape(B, n, heads, 32)
    dP = torch.einsum("bnhe,bhde->bnhd", vh, dctx)
    dk_ref = (Ph * (dP - rvec.view(B, 1, heads, 32))).reshape(B, n, HD)
    dv_ref = torch.einsum("bnhd,bhde->bnhe", Ph, dctx).reshape(B, n, HD)
    dqkv = be.zeros(B, n, 3 * HD)
    be.L.cdf_linattn_bwd_kv(P(be.to(qkv)), 3 * HD, HD, P(be.to(dctx)), P(be.to(rvec)), P(be.to(kmax)), P(be.to(ksum)), P(dqkv), 3 * HD, HD, B, n, heads,
                            be.stream())
    out = dqkv.cpu()
    assert (out[..., :HD] == 0).all()                                   # the q block is not this kernel's
    assert err(out[..., HD:2 * HD], dk_ref) <= 2e-5 * max(1.0, dk_ref.abs().max().item())
    assert err(out[..., 2 * HD:], dv_ref) <= 2e-5 * max(1.0, dv_ref.abs().max().item())


# ---------------------------------------------------------------------------------------------
# specialised GEMM epilogues (csrc/cdf_epilogue.h, round 6) == the generic run-time-selected form, bit for bit
# ---------------------------------------------------------------------------------------------
EPI_SPECS = ["plain", "res", "gelu_pre_planes", "gelu_planes", "mulgelu_planes", "acc",
             "bf_plain", "bf_res", "bf_gelu_pre", "bf_gelu", "bf_mulgelu"]


def _epi_case(be, spec, B, Cin, Cout, H, sbias=False):
    """One 3 x 3 stride-1 pre-split GEMM with the operand list of one EpiSpec, run with cdf_gemm_tuning.epilogue = 1 (specialised
    straight-line form) and = 0 (generic): every output tensor must be identical."""
    bf = spec.startswith("bf_")
    torch.manual_seed(3)
    k = 3
    x = torch.randn(B, H, H, Cin)
    w = torch.randn(Cout, Cin, k, k) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout)
    sb = torch.randn(B, Cout) if sbias else None
    other = torch.randn(B, H, H, Cout)                       # residual / GELU' source / previous contents of y
    zero = be.zeros(16)
    wd_ = be.to(w)
    ldk = (Cin + 31) // 32 * 32
    whi = torch.empty(9, Cout, ldk, dtype=torch.int16, device=be.device)
    wlo = None if bf else torch.empty_like(whi)
    be.L.cdf_pack_weight_bf16(P(wd_), P(whi), P(wlo), 9, Cout, Cin, ldk, 1, Cin * 9, 9, be.stream())
    xs = _split(be, be.to(x), bf)
    plan = cd.conv_fwd(H, H, k, k, 1, 1, 1, 1, 1)
    as_bf = lambda t: (t.bfloat16().view(torch.int16))       # a bf16 tensor as its raw plane
    biasd, sbd = be.to(bias), (be.to(sb) if sbias else None)
    otherd = be.to(as_bf(other) if bf else other)

    def run(epilogue):
        be.tune.set(epilogue=epilogue)
        y = be.to(other.clone()) if spec == "acc" else be.zeros(B, H, H, Cout)
        pre = torch.zeros(B, H, H, Cout, dtype=torch.int16 if bf else torch.float32, device=be.device)
        yh = torch.zeros(B, H, H, Cout, dtype=torch.int16, device=be.device)
        yl = torch.zeros_like(yh)
        be._keep += [pre, yh, yl]
        a = dict(y=0, bias=P(biasd), res=0, pre=0, mul=0, act=0, mul_mode=0, acc=0, io=0, yh=0, yl=0)
        if spec in ("plain", "res", "acc"):
            a["y"] = P(y)
        if spec in ("res", "bf_res"):
            a["res"] = P(otherd)
        if "gelu" in spec and "mul" not in spec:
            a["act"] = 1
        if "pre" in spec:
            a["pre"] = P(pre)
        if "mulgelu" in spec:
            a["mul"], a["mul_mode"], a["bias"] = P(otherd), 1, 0
        if spec == "acc":
            a["acc"], a["bias"] = 1, 0
        if "planes" in spec or bf:
            a["yh"] = P(yh)
            a["yl"] = 0 if bf else P(yl)
        if bf:
            a["io"] = {"bf_res": 1, "bf_gelu_pre": 2, "bf_mulgelu": 4}.get(spec, 0)
        be.L.cdf_conv_gemm_bf16x_io(P(xs[0]), P(xs[1]), xs[0].shape[-1], P(zero), P(whi), P(wlo), ldk, a["y"], Cout, B, H, H, Cin, H, H, Cout, H, H, 1, 1,
                                    1, plan.desc, a["bias"], P(sbd), Cout if sbias else 0, a["res"], Cout, a["pre"], Cout, a["mul"], Cout, a["act"],
                                    a["mul_mode"], a["acc"], a["io"], a["yh"], a["yl"], Cout, 0, 0, be.tune.ptr, be.stream())
        return [t.cpu().clone() for t in (y, pre, yh, yl)]

    try:
        fast, generic = run(1), run(0)
    finally:
        be.tune.set(epilogue=1)
    for name, f, g in zip(("y", "pre", "y_hi", "y_lo"), fast, generic):
        assert torch.equal(f, g), (spec, name, (f.float() - g.float()).abs().max().item())
    assert any(t.abs().sum() > 0 for t in fast), "the launch wrote nothing"


@pytest.mark.parametrize("spec", EPI_SPECS)
def test_specialised_epilogue_equals_generic(be, spec):
    # 3 x 3, 64 -> 64 channels, 16 x 16 pixels, 2 images: whole 128-pixel tiles and whole 64-wide N tiles -> the halo kernel's specialised form
    _epi_case(be, spec, 2, 64, 64, 16)


@pytest.mark.parametrize("spec", ["res", "gelu_pre_planes", "bf_res"])
def test_specialised_epilogue_with_per_sample_bias(be, spec):
    # one image per 128-row tile is what the per-sample bias needs (two tiles per image here)
    _epi_case(be, spec, 2, 64, 64, 16, sbias=True)


@pytest.mark.parametrize("spec", ["res", "mulgelu_planes", "bf_gelu_pre"])
def test_specialised_epilogue_rowhalo_emu(spec):
    """... through the resident row-halo kernel's two-pass form (operand of pass 1 refilled inside pass 0), at a simulator-sized width."""
    from conftest import Backend
    be = Backend("emu")
    be.tune.set(halo=47 | 64)
    _epi_case(be, spec, 2, 64, 64, 16)
    _epi_case(be, spec, 1, 128, 128, 16)


def _lnbwd_case(be, B, Cmid, C, H, tune=None):
    """cdf_conv_gemm_bf16x_lnbwd (data gradient of the 3 x 3 convolution behind a channel LayerNorm, with the LayerNorm backward in the
    epilogue) against the two launches it replaces: the same GEMM writing dhn, then cdf_layernorm_c_bwd."""
    torch.manual_seed(5)
    dy = torch.randn(B, H, H, Cmid)
    w = torch.randn(C, Cmid, 3, 3) / math.sqrt(Cmid * 9)     # (already in data-gradient orientation: rows = the LayerNorm's channels)
    h = torch.randn(B, H, H, C) * 1.5 + 0.3
    g = torch.randn(C)
    M = B * H * H
    mean = h.mean(-1).reshape(M)
    rstd = (h.var(-1, unbiased=False) + 1e-5).rsqrt().reshape(M)
    zero = be.zeros(16)
    ldk = (Cmid + 31) // 32 * 32
    whi = torch.empty(9, C, ldk, dtype=torch.int16, device=be.device)
    wlo = torch.empty_like(whi)
    be.L.cdf_pack_weight_bf16(P(be.to(w)), P(whi), P(wlo), 9, C, Cmid, ldk, 1, Cmid * 9, 9, be.stream())
    xs = _split(be, be.to(dy), False)
    plan = cd.conv_fwd(H, H, 3, 3, 1, 1, 1, 1, 1)
    assert be.L.cdf_conv_gemm_bf16x_lnbwd_ok(B, H, H, Cmid, C, plan.nphase, plan.desc[2])
    hd, gd, md, rd = be.to(h), be.to(g), be.to(mean), be.to(rstd)
    old = {k: be.tune.get(k) for k in (tune or {})}
    be.tune.set(**(tune or {}))
    try:
        # reference pair
        dhn = be.zeros(B, H, H, C)
        be.L.cdf_conv_gemm_bf16x(P(xs[0]), P(xs[1]), xs[0].shape[-1], P(zero), P(whi), P(wlo), ldk, P(dhn), C, B, H, H, Cmid, H, H, C, H, H, 1, 1, 1,
                                 plan.desc, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, be.tune.ptr, be.stream())
        dh0, dg0, db0 = be.zeros(B, H, H, C), be.zeros(C), be.zeros(C)
        part0 = be.zeros(be.L.cdf_layernorm_blocks(M, C) * 2 * C)
        be.L.cdf_layernorm_c_bwd(P(dhn), C, P(hd), C, P(gd), P(md), P(rd), P(dh0), C, 0, 0, P(dg0), P(db0), P(part0), M, C, 0, 0, be.stream())
        # fused; parameter gradients ACCUMULATE (the reference pair above wrote into zeros: start from a known non-zero value)
        dh1, dg1, db1 = be.zeros(B, H, H, C), be.to(torch.full((C,), 2.0)), be.to(torch.full((C,), -3.0))
        part1 = be.zeros(M // 64 * 2 * C)
        be.L.cdf_conv_gemm_bf16x_lnbwd(P(xs[0]), P(xs[1]), xs[0].shape[-1], P(zero), P(whi), P(wlo), ldk, B, H, H, Cmid, C, plan.desc, P(hd), C,
                                       P(md), P(rd), P(gd), P(dh1), C, P(dg1), P(db1), P(part1), be.tune.ptr, be.stream())
    finally:
        be.tune.set(**old)
    dh0, dh1 = dh0.cpu(), dh1.cpu()
    scale = dh0.abs().max().item()
    assert scale > 0
    # same GEMM sums (possibly another tile shape => another summation order), same per-pixel expressions: fp32 rounding apart
    assert (dh0 - dh1).abs().max().item() <= 2e-5 * scale, (dh0 - dh1).abs().max().item() / scale
    for a0, a1, off in ((dg0, dg1, 2.0), (db0, db1, -3.0)):
        a0, a1 = a0.cpu(), a1.cpu() - off
        assert (a0 - a1).abs().max().item() <= 2e-5 * max(1.0, a0.abs().max().item()) * math.sqrt(M / 256), (a0 - a1).abs().max().item()


@pytest.mark.parametrize("shape", [(2, 128, 64, 16), (1, 256, 128, 16), (4, 128, 64, 16)])
def test_conv_dgrad_with_layernorm_backward_epilogue(be, shape):
    _lnbwd_case(be, *shape)


def test_conv_dgrad_with_layernorm_backward_epilogue_generic_kernels(be):
    # halo = 0: the generic (non-resident) kernels' tiles -- 64 / 128 rows at these sizes
    _lnbwd_case(be, 2, 128, 64, 16, tune=dict(halo=0))
    _lnbwd_case(be, 1, 128, 128, 16, tune=dict(halo=0))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 128, 64, 128), (4, 256, 128, 64), (32, 128, 64, 128), (2, 256, 128, 32)])
def test_conv_dgrad_with_layernorm_backward_epilogue_large(shape):
    from conftest import Backend
    _lnbwd_case(Backend("hip"), *shape)


@pytest.mark.gpu
@pytest.mark.parametrize("spec", EPI_SPECS)
@pytest.mark.parametrize("shape", [(4, 64, 128, 128), (4, 128, 64, 128), (4, 128, 256, 64), (8, 512, 256, 16)])
def test_specialised_epilogue_equals_generic_large(spec, shape):
    from conftest import Backend
    _epi_case(Backend("hip"), spec, *shape)

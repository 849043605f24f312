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
    # io_bf16 = 2: bf16 x / res, fp32 y (the data gradient handed to the image-side block's fp32 tensors): the fp32 kernel's result exactly
    y32, yref2 = be.empty(B, H, H, Cp), be.empty(B, H, H, Cp)
    be.L.cdf_dwconv7(P(xf), Cp, P(wp), Cp, 0, 0, 0, P(yref2), Cp, B, H, H, Cp, 1, 0, P(rf), Cp, be.stream())
    be.L.cdf_dwconv7_io(P(xb), Cp, P(wp), Cp, 0, 0, 0, P(y32), Cp, B, H, H, Cp, 1, 0, P(rb), Cp, 2, be.stream())
    assert torch.equal(y32.cpu(), yref2.cpu())
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


# ---------------------------------------------------------------------------------------------------------------------------------------
# network level: the bf16-storage engine against the fp32 oracle, within the mode's stated tolerance (runtime.BF16_TOLERANCE -- the SAME
# numbers as with fp32 tensors), on the simulator and on the MI355X
# ---------------------------------------------------------------------------------------------------------------------------------------
from test_modules import mbe, quiet  # noqa: E402,F401


@pytest.fixture
def bf16_mode():
    from colddiff import runtime as rt
    saved = rt.precision
    rt.set_precision("bf16")
    rt.bump_weights_epoch()
    yield
    rt.set_precision(saved)
    rt.bump_weights_epoch()


def _grad_worst(net, ref):
    gmax = max(g.abs().max().item() for g in ref.values())
    worst, name = 0.0, None
    for n, p in net.named_parameters():
        r = ref[n]
        v = (p.grad.cpu() - r).abs().max().item() / max(r.abs().max().item(), 1e-2 * gmax)
        if v > worst:
            worst, name = v, n
    return worst, name


@pytest.mark.parametrize("dim,mults,size", [(8, (1, 2), 16), (16, (1, 2, 4), 16)])
def test_unet_bf16_storage_vs_oracle(mbe, bf16_mode, monkeypatch, dim, mults, size):
    """Unet forward + backward with every tensor between kernels in bf16: the stream really is bf16 (checked on the blocks' outputs and on
    what they save), output and every gradient within BF16_TOLERANCE of the fp32 oracle."""
    from colddiff import bf16store as BFS, ops
    from colddiff.runtime import BF16_TOLERANCE
    from deblurring_diffusion_pytorch import Unet
    from oracle import cold_oracle as O
    assert BFS.enabled()
    torch.manual_seed(9)
    net = quiet(Unet, dim=dim, dim_mults=mults, channels=3)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    x, t = torch.rand(2, 3, size, size) * 2 - 1, torch.tensor([1, 40])
    gy = torch.randn(2, 3, size, size) / 1000
    ps = {k: v.clone().requires_grad_() for k, v in sd.items()}
    yr = O.unet_forward(ps, x, t)
    yr.backward(gy)
    net = net.to(mbe.device)
    seen = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, n=n: seen.__setitem__(n, (i[0].dtype, o.dtype)))
             for n, m in net.named_modules() if type(m).__name__ in ("ConvNextBlock", "Residual")]
    saved_dtypes = set()
    with torch.autograd.graph.saved_tensors_hooks(lambda t_: (saved_dtypes.add((t_.dtype, t_.dim())), t_)[1], lambda t_: t_):
        y = net(mbe.to(x), mbe.to(t))
    for h in hooks:
        h.remove()
    y.backward(mbe.to(gy))
    assert seen["downs.0.0"] == (torch.float32, torch.float32)                    # the image-side block: 4-channel fp32 tensors
    assert seen["downs.0.1"] == (torch.float32, torch.bfloat16)                   # ... whose output enters the stream inside the next block
    assert all(v == (torch.bfloat16, torch.bfloat16) for k, v in seen.items() if k not in ("downs.0.0", "downs.0.1")), seen
    assert (torch.bfloat16, 4) in saved_dtypes
    e = (y.cpu() - yr.detach()).abs().max().item()
    worst, name = _grad_worst(net, {k: v.grad for k, v in ps.items()})
    print("bf16 storage", dim, mults, ": forward max-abs error", e, "worst gradient error / scale", worst, name)
    assert e <= BF16_TOLERANCE["forward_max_abs"] and worst <= BF16_TOLERANCE["grad_rel_of_tensor_max"]
    # the round 2-4 form of the mode (fp32 tensors, bf16 GEMM operands) is still there for A/B
    monkeypatch.setenv("COLDDIFF_BF16_STORAGE", "0")
    assert not BFS.enabled()
    with torch.no_grad():
        y0 = net(mbe.to(x), mbe.to(t))
    assert (y0.cpu() - yr.detach()).abs().max().item() <= BF16_TOLERANCE["forward_max_abs"]


def test_trainer_and_sampler_on_the_bf16_stream(mbe, bf16_mode, tmp_path):
    """Two optimizer steps (fused accumulation, Adam on the fp32 master weights) and a sampler call with the bf16 stream: the loss follows
    the fp32 oracle's within the mode's loss tolerance and the sampler's output stays within its forward tolerance per call."""
    from colddiff.runtime import BF16_TOLERANCE
    from denoising_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    from oracle import cold_oracle as O
    torch.manual_seed(0)
    net = quiet(Unet, dim=8, dim_mults=(1, 2), channels=3).to(mbe.device)
    diff = GaussianDiffusion(net, image_size=16, channels=3, timesteps=10, sampling_routine="x0_step_down").to(mbe.device)
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    tr = Trainer(diff, None, image_size=16, train_batch_size=2, train_lr=2e-5, train_num_steps=2, gradient_accumulate_every=2,
                 dataset="synthetic", results_folder=str(tmp_path / "res"))
    g = torch.Generator().manual_seed(1)
    batches = [[(torch.rand(2, 3, 16, 16, generator=g) * 2 - 1, torch.randn(2, 3, 16, 16, generator=g), torch.randint(0, 10, (2,), generator=g))
                for _ in range(2)] for _ in range(2)]
    ca, cb = O.cosine_tables(10)
    otr = O.OracleTrainer(sd0, lambda p, x, e, t: O.loss_fn(x, O.unet_forward(p, O.noise_q_sample(x, e, t, ca, cb), t)), lr=2e-5, accumulate=2)
    for s in range(2):
        it = iter(batches[s])

        def micro(it=it):
            x, e, t = (mbe.to(v) for v in next(it))
            return tr.core.prepare(x, e, t=t)
        tr._prepare_micro = micro
        loss = tr.train_step()
        tr.step += 1
        lo = otr.train_step(batches[s])
        assert abs(loss.item() - lo) <= 5 * BF16_TOLERANCE["loss_rel"] * abs(lo), (loss.item(), lo)       # (a dim-8 net at 16 x 16: few terms)
    noise = torch.randn(2, 3, 16, 16)
    with torch.no_grad():
        _, direct, _ = quiet(diff.gen_sample, batch_size=2, img=mbe.to(noise))
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        rdirect = O.unet_forward(sd, noise, torch.full((2,), 9))
    assert (direct.cpu() - rdirect).abs().max().item() <= BF16_TOLERANCE["forward_max_abs"]


def test_stored_gelu_derivative_is_bit_identical(mbe, monkeypatch):
    """conv1's epilogue can store GELU'(v) (out of the erf / exp evaluation GELU needs anyway) in place of the pre-activation, and conv2's
    data gradient then multiplies by the stored value instead of evaluating erf / exp per element again: the same function of the same
    fp32 input, so every gradient must come out bit for bit the same (parity-grade bf16x3 arithmetic, pre-split path: dim >= 64)."""
    from colddiff import functions as F_
    from colddiff import runtime as rt
    from colddiff.unet import ConvNextBlock, anchor
    assert rt.precision == "bf16x3"
    torch.manual_seed(4)
    blk = ConvNextBlock(64, 64, time_emb_dim=16).to(mbe.device)
    x = mbe.to(torch.randn(2, 8, 8, 64))
    gt = mbe.to(torch.randn(2, 16))
    dy = mbe.to(torch.randn(2, 8, 8, 64))
    outs = []
    for flag in (False, True):
        monkeypatch.setattr(F_, "_PRE_GRAD", flag)
        for p in blk.parameters():
            p.grad = None
        xi = x.clone().requires_grad_()
        y = blk(xi, gt)
        y.backward(dy)
        outs.append([y.detach().clone(), xi.grad.clone()] + [p.grad.clone() for p in blk.parameters()])
    assert all(torch.equal(a, b) for a, b in zip(*outs))


@pytest.mark.parametrize("B,n,heads", [(2, 70, 4), (1, 300, 2)])
def test_linattn_bwd_kv_planes(be, B, n, heads):
    """cdf_linattn_bwd_kv_planes: dk | dv as bf16 hi / lo operand planes must be cdf_split_bf16 of what cdf_linattn_bwd_kv writes in fp32,
    bit for bit (hi-only planes: the rounding of it)."""
    torch.manual_seed(n)
    HD = heads * 32
    kv = be.to(torch.randn(B, n, 2 * HD))
    k = kv[..., :HD].cpu()
    kmax = k.max(1).values
    ksum = torch.exp(k - kmax[:, None]).sum(1)
    dctx, rvec = be.to(torch.randn(B, heads, 32, 32) * 0.3), be.to(torch.randn(B, HD))
    km, ks = be.to(kmax), be.to(ksum)
    ref = be.zeros(B, n, 2 * HD)
    be.L.cdf_linattn_bwd_kv(P(kv), 2 * HD, 0, P(dctx), P(rvec), P(km), P(ks), P(ref), 2 * HD, 0, B, n, heads, be.stream())
    hi, lo, hi1 = (torch.zeros(B, n, 2 * HD, dtype=torch.int16, device=be.device) for _ in range(3))
    be.L.cdf_linattn_bwd_kv_planes(P(kv), 2 * HD, 0, P(dctx), P(rvec), P(km), P(ks), P(hi), P(lo), 2 * HD, 0, B, n, heads, be.stream())
    be.L.cdf_linattn_bwd_kv_planes(P(kv), 2 * HD, 0, P(dctx), P(rvec), P(km), P(ks), P(hi1), 0, 2 * HD, 0, B, n, heads, be.stream())
    rh, rl = (torch.zeros(B, n, 2 * HD, dtype=torch.int16, device=be.device) for _ in range(2))
    be.L.cdf_split_bf16(P(ref), 2 * HD, P(rh), P(rl), 2 * HD, B * n, 2 * HD, be.stream())
    assert torch.equal(hi.cpu(), rh.cpu()) and torch.equal(lo.cpu(), rl.cpu()) and torch.equal(hi1.cpu(), rh.cpu())
    assert ref.abs().max().item() > 0.01


def test_conv_gemm_io_reads_and_writes_the_bf16_stream(be):
    """cdf_conv_gemm_io, the exact-fp32 GEMM at the attention block's boundary: a batched (per image) launch whose residual operand is a
    bf16 tensor and whose result goes out as a bf16 plane only -- against the fp32 launch on the widened residual, rounded."""
    from colddiff import convdesc as cd
    torch.manual_seed(6)
    B, n, dim = 3, 80, 64
    xn, Nb = be.to(torch.randn(B, n, dim)), be.to(torch.randn(B, dim, dim) / 8)
    bias, res = be.to(torch.randn(dim)), torch.randn(B, n, dim + 8)
    desc = cd.conv_fwd(1, n, 1, 1, 1, 0, 0, 0, 0).desc
    yref = be.empty(B, n, dim)
    be.L.cdf_conv_gemm(P(xn), dim, P(Nb), dim, P(yref), dim, 1, 1, n, dim, 1, n, dim, 1, n, 1, 1, 1, desc, P(bias), 0, 0, P(be.to(bf(res).float())), dim + 8,
                       0, 0, 0, 0, 0, 0, 0, 0, B, n * dim, dim * dim, n * dim, 1, 0, 0, 0, be.stream())
    y = torch.zeros(B, n, dim + 8, dtype=torch.bfloat16, device=be.device)          # (a pitched destination: a slice of a concat buffer)
    be.L.cdf_conv_gemm_io(P(xn), dim, P(Nb), dim, 0, dim + 8, 1, 1, n, dim, 1, n, dim, 1, n, 1, 1, 1, desc, P(bias), 0, 0, P(be.to(bf(res))), dim + 8,
                          0, 0, 0, 0, 0, 0, 0, 0, B, n * dim, dim * dim, n * (dim + 8), 1, 0, 0, 0, 1, P(y), dim + 8, be.stream())
    assert torch.equal(y[..., :dim].cpu(), rbf(yref)) and (y[..., dim:].cpu() == 0).all()


@pytest.mark.parametrize("kw", [dict(channels=1), dict(with_time_emb=False), dict(residual=True), dict(dim_mults=(1, 2, 4, 8))])
def test_unet_variants_on_the_bf16_stream(mbe, bf16_mode, kw):
    """The constructor variants the reference's scripts use (one-channel MNIST net, no time embedding, residual output, four levels) on the
    bf16 stream: forward within the mode's tolerance of the fp32 oracle, backward runs (gradients finite, every parameter reached)."""
    from colddiff.runtime import BF16_TOLERANCE
    from deblurring_diffusion_pytorch import Unet
    from oracle import cold_oracle as O
    torch.manual_seed(13)
    args = dict(dim=8, dim_mults=(1, 2), channels=3)
    args.update(kw)
    net = quiet(Unet, **args)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    C = args["channels"]
    x, t = torch.rand(2, C, 16, 16) * 2 - 1, torch.tensor([0, 7])
    with torch.no_grad():
        yr = O.unet_forward(sd, x, t, residual=args.get("residual", False))
    net = net.to(mbe.device)
    y = net(mbe.to(x), mbe.to(t))
    assert (y.cpu() - yr).abs().max().item() <= BF16_TOLERANCE["forward_max_abs"]
    y.backward(mbe.to(torch.randn(2, C, 16, 16) / 100))
    for n, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n

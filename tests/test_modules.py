"""Product parity: the drop-in packages (Unet / Model / GaussianDiffusion / Trainer on the HIP kernels)
against the reference-generated golden vectors and the CPU oracle.

Backends (fixture `mbe`):  emu = kernels on the host SIMT simulator, CPU tensors (not gpu)
                           hip = libcolddiff_hip.so on cuda:0 (@pytest.mark.gpu)
Tolerances are fp32: 1e-4 max-abs on outputs (north_star), relative 1e-3 on gradients whose true
value is not ~0; q_sample / mask / area-pixelate / schedules are compared exactly where the
reference arithmetic is reproduced op for op.
"""
import contextlib
import io
import os

import pytest
import torch

from oracle import cold_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return torch.load(os.path.join(GOLD, name), weights_only=False)


class MBE:
    def __init__(self, kind):
        self.kind = kind
        self.device = torch.device("cuda:0" if kind == "hip" else "cpu")

    def to(self, t):
        return t.to(self.device)


@pytest.fixture(params=[pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)])
def mbe(request):
    from colddiff import runtime
    if request.param == "emu":
        from emu_util import install_emu
        install_emu()
    else:
        runtime._lib_override = None
    yield MBE(request.param)
    runtime._lib_override = None


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def grad_check(named_params, ref_grads, tol=1e-3):
    gmax = max(v.abs().max().item() for v in ref_grads.values())
    for name, p in named_params:
        r = ref_grads[name]
        e = (p.grad.detach().cpu() - r).abs().max().item()
        assert e <= tol * max(r.abs().max().item(), 1e-3 * gmax), (name, e, r.abs().max().item())


def test_unet_golden(mbe):
    from deblurring_diffusion_pytorch import Unet
    g = load("unet_dim8.pt")
    net = quiet(Unet, **g["cfg"])
    assert list(net.state_dict().keys()) == list(g["sd"].keys())
    net.load_state_dict(g["sd"])
    net = net.to(mbe.device)
    y = net(mbe.to(g["x"]), mbe.to(g["t"]))
    assert (y.cpu() - g["y"]).abs().max() <= 1e-4
    y.backward(mbe.to(g["gy"]))
    grad_check(net.named_parameters(), g["grads"])


def test_gradient_planes_ride_on_the_gradient_tensors(mbe, monkeypatch):
    """Round 6: the depthwise data-gradient kernel and the attention block's LayerNorm backward write their result as bf16 hi / lo planes too
    and attach them to the gradient tensor; the consuming block takes them instead of launching cdf_split_bf16.  Same planes => every
    parameter gradient and the input gradient must be BIT-identical with the mechanism on and off, the planes must really be taken (fewer
    split launches), and a gradient the engine summed in place (two consumers) must not be served stale planes."""
    from deblurring_diffusion_pytorch import Unet
    from colddiff import ops
    torch.manual_seed(3)
    net = quiet(Unet, dim=64, dim_mults=(1, 2, 4), channels=3).to(mbe.device)       # (attention blocks in the folded AND the plain form)
    x = mbe.to(torch.randn(2, 3, 16, 16))
    t = mbe.to(torch.tensor([3, 7]))
    gy = mbe.to(torch.randn(2, 3, 16, 16))
    calls = []
    real_split, real_take = ops.split_bf16, ops.grad_planes
    monkeypatch.setattr(ops, "split_bf16", lambda v: (calls.append("split"), real_split(v))[1])
    monkeypatch.setattr(ops, "grad_planes", lambda v: (lambda r: (calls.append("taken") if r is not None else None, r)[1])(real_take(v)))

    def run(on):
        monkeypatch.setattr(ops, "GRAD_PLANES", on)
        del calls[:]
        net.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        net(xi, t).backward(gy)
        return [xi.grad.clone()] + [p.grad.clone() for p in net.parameters()], calls.count("split"), calls.count("taken")

    g_on, splits_on, taken_on = run(True)
    g_off, splits_off, taken_off = run(False)
    assert taken_off == 0 and taken_on >= 4, (taken_on, taken_off)
    assert splits_on == splits_off - taken_on, (splits_on, splits_off, taken_on)
    assert all(torch.equal(a, b) for a, b in zip(g_on, g_off))
    # a tensor whose contents changed after the planes were attached (the engine's in-place gradient sum) is not served
    v = mbe.to(torch.randn(1, 4, 4, 64))
    ops.attach_planes(v, real_split(v))
    assert real_take(v) is not None
    v.add_(1.0)
    assert real_take(v) is None


@pytest.mark.parametrize("dim,H", [(16, 12), (160, 8), (64, 16)])
def test_linear_attention_block_all_forms(mbe, dim, H):
    """Residual(PreNorm(LinearAttention)) in its three arithmetic forms -- plain (per-head products + to_out conv), to_out folded into a
    per-image matrix, q folded in as well (k|v-only projection; only where dim <= heads*32) -- against a torch restatement of
    deblurring_diffusion_pytorch.py:83-89,111-131,167-187: output and every gradient."""
    from colddiff import functions as F_, unet as D
    from einops import rearrange
    torch.manual_seed(dim)
    blk = D.Residual(D.PreNorm(dim, D.LinearAttention(dim)))
    with torch.no_grad():
        blk.fn.norm.g.add_(0.3 * torch.randn_like(blk.fn.norm.g))
        blk.fn.norm.b.add_(0.3 * torch.randn_like(blk.fn.norm.b))
    B = 2
    x0 = torch.randn(B, dim, H, H)
    gy = torch.randn(B, dim, H, H)
    # torch reference
    xr = x0.clone().requires_grad_(True)
    n, att = blk.fn.norm, blk.fn.fn
    var, mean = torch.var(xr, dim=1, unbiased=False, keepdim=True), torch.mean(xr, dim=1, keepdim=True)
    xn = (xr - mean) / (var + n.eps).sqrt() * n.g.detach() + n.b.detach()
    wq, wo, bo = att.to_qkv.weight.detach().clone().requires_grad_(True), att.to_out.weight.detach().clone().requires_grad_(True), \
        att.to_out.bias.detach().clone().requires_grad_(True)
    qkv = torch.nn.functional.conv2d(xn, wq).chunk(3, dim=1)
    q, k, v = [rearrange(t_, "b (h c) x y -> b h c (x y)", h=att.heads) for t_ in qkv]
    q = q * att.scale
    k = k.softmax(dim=-1)
    cx = torch.einsum("b h d n, b h e n -> b h d e", k, v)
    out = rearrange(torch.einsum("b h d e, b h d n -> b h e n", cx, q), "b h c (x y) -> b (h c) x y", x=H, y=H)
    yr = torch.nn.functional.conv2d(out, wo, bo) + xr
    yr.backward(gy)
    ref = {"x": xr.grad, "wq": wq.grad, "wo": wo.grad, "bo": bo.grad}
    blk = blk.to(mbe.device)
    saved = (F_._ATTN_FUSED, F_._ATTN_QFOLD)
    forms = [(0, False), (2, False)] + ([(2, True)] if dim <= att.heads * 32 else [])
    try:
        for fused, qfold in forms:
            F_._ATTN_FUSED, F_._ATTN_QFOLD = fused, qfold
            for p in blk.parameters():
                p.grad = None
            xd = mbe.to(x0.permute(0, 2, 3, 1).contiguous()).requires_grad_(True)
            y = blk(xd)
            y.backward(mbe.to(gy.permute(0, 2, 3, 1).contiguous()))
            tag = (dim, fused, qfold)
            assert (y.detach().cpu().permute(0, 3, 1, 2) - yr.detach()).abs().max() <= 1e-4, tag
            got = {"x": xd.grad.cpu().permute(0, 3, 1, 2), "wq": att.to_qkv.weight.grad.cpu(), "wo": att.to_out.weight.grad.cpu(),
                   "bo": att.to_out.bias.grad.cpu()}
            for kname, r in ref.items():
                assert (got[kname] - r).abs().max() <= 1e-3 * max(1.0, r.abs().max().item()), (tag, kname)
    finally:
        F_._ATTN_FUSED, F_._ATTN_QFOLD = saved


@pytest.mark.parametrize("fixture", ["model_ch32.pt", "model_noconv.pt"])       # the latter: resamp_with_conv=False (avg-pool down, bare nearest up)
def test_model_golden(mbe, fixture):
    from deblurring_diffusion_pytorch import Model
    g = load(fixture)
    net = Model(**g["cfg"])
    assert list(net.state_dict().keys()) == list(g["sd"].keys())
    net.load_state_dict(g["sd"])
    net = net.to(mbe.device)
    y = net(mbe.to(g["x"]), mbe.to(g["t"]))
    assert (y.cpu() - g["y"]).abs().max() <= 1e-4
    y.backward(mbe.to(g["gy"]))
    grad_check(net.named_parameters(), g["grads"], tol=2e-3)


def _net(mbe, sd):
    from deblurring_diffusion_pytorch import Unet
    net = quiet(Unet, dim=8, dim_mults=(1, 2), channels=3)
    net.load_state_dict(sd)
    return net.to(mbe.device)


def test_deblurring_golden(mbe):
    from deblurring_diffusion_pytorch import GaussianDiffusion
    g = load("diffusion.pt")
    net = _net(mbe, g["deblur/net_sd"])
    for key, c in g.items():
        if not key.startswith("deblur/") or key == "deblur/net_sd":
            continue
        _, routine, sampling = key.split("/")
        d = GaussianDiffusion(net, image_size=16, device_of_kernel="cuda", channels=3, timesteps=c["T"], kernel_std=c["std"],
                              kernel_size=c["ks"], blur_routine=routine, sampling_routine=sampling).to(mbe.device)
        for m, w in zip(d.gaussian_kernels, c["kernels"]):          # same generator as the reference's (torchgeometry) kernels
            assert torch.equal(m.weight.detach().cpu(), w)
        with torch.no_grad():
            q = d.q_sample(mbe.to(c["x"]), mbe.to(c["t"]))
            assert (q.cpu() - c["q"]).abs().max() <= 2e-6, key
            xt, direct, img = d.sample(batch_size=3, img=mbe.to(c["x"]))
        assert (xt.cpu() - c["xt"]).abs().max() <= 2e-6 and (img.cpu() - c["img"]).abs().max() <= 1e-4, key


def test_denoising_golden(mbe):
    from denoising_diffusion_pytorch import GaussianDiffusion
    g = load("diffusion.pt")
    net = _net(mbe, g["deblur/net_sd"])
    for sampling in ("x0_step_down", "ddim"):
        c = g[f"denoise/{sampling}"]
        d = GaussianDiffusion(net, image_size=16, channels=3, timesteps=c["T"], sampling_routine=sampling).to(mbe.device)
        assert torch.equal(d.sqrt_alphas_cumprod.cpu(), c["ca"]) and torch.equal(d.sqrt_one_minus_alphas_cumprod.cpu(), c["cb"])
        with torch.no_grad():
            assert torch.equal(d.q_sample(mbe.to(c["x"]), mbe.to(c["eps"]), mbe.to(c["t"])).cpu(), c["q"])
            assert (d.gen_sample(batch_size=3, img=mbe.to(c["eps"]))[2].cpu() - c["gen"]).abs().max() <= 1e-4
            assert (d.sample(batch_size=3, img=mbe.to(c["eps"]))[2].cpu() - c["sample"]).abs().max() <= 1e-4


def test_resolution_golden(mbe):
    from resolution_diffusion_pytorch import GaussianDiffusion
    g = load("diffusion.pt")
    net = _net(mbe, g["deblur/net_sd"])
    for key, c in g.items():
        if not key.startswith("resolution/"):
            continue
        routine = key.split("/")[1]
        d = GaussianDiffusion(net, image_size=16, device_of_kernel="cuda", channels=3, timesteps=c["T"], resolution_routine=routine,
                              sampling_routine="x0_step_down")
        with torch.no_grad():
            q = d.q_sample(mbe.to(c["x"]), mbe.to(c["t"]))
            exact = "_area" in routine                      # avg-pool down + nearest-exact up: bit-exact indexing and sums
            assert (q.cpu() - c["q"]).abs().max() <= (0.0 if exact else 1e-5), key
            xt, _, img = d.sample(batch_size=3, img=mbe.to(c["x"]))
        assert (xt.cpu() - c["xt"]).abs().max() <= (0.0 if exact else 1e-5) and (img.cpu() - c["img"]).abs().max() <= 1e-4, key


def _lists_close(a, b, tol):
    assert len(a) == len(b)
    for u, v in zip(a, b):
        assert (u.cpu() - v).abs().max() <= tol


def test_deblurring_sampler_variants_golden(mbe):
    """forward_and_backward(_2), sample_from_blur, all_sample (DEBLUR:610-925) against the reference's own outputs."""
    from deblurring_diffusion_pytorch import GaussianDiffusion
    g = load("variants.pt")
    net = _net(mbe, load("diffusion.pt")["deblur/net_sd"])
    for key, c in g.items():
        if not key.startswith("deblur/"):
            continue
        _, routine, sampling = key.split("/")
        d = GaussianDiffusion(net, image_size=16, device_of_kernel="cuda", channels=3, timesteps=c["T"], kernel_std=c["std"],
                              kernel_size=c["ks"], blur_routine=routine, sampling_routine=sampling).to(mbe.device)
        x = mbe.to(c["x"])
        F1, B1, i1 = d.forward_and_backward(batch_size=2, img=x)
        _lists_close(F1, c["fab"][0], 2e-6), _lists_close(B1, c["fab"][1], 1e-4), _lists_close([i1], [c["fab"][2]], 1e-4)
        F2, Ba, Bb, ia, ib = d.forward_and_backward_2(batch_size=2, img=x)
        _lists_close(F2, c["fab2"][0], 2e-6), _lists_close(Ba, c["fab2"][1], 1e-4), _lists_close(Bb, c["fab2"][2], 1e-4)
        _lists_close([ia, ib], list(c["fab2"][3:]), 1e-4)
        _lists_close(list(d.sample_from_blur(batch_size=2, img=mbe.to(c["half"]), start=2)), list(c["from_blur"]), 1e-4)
        X0, Xt = d.all_sample(batch_size=2, img=x, times=3)
        _lists_close(X0, c["all_sample"][0], 1e-4), _lists_close(Xt, c["all_sample"][1], 1e-4)


def test_denoising_forward_and_backward(mbe):
    """DENOISE:438-479; its single randn_like draw is replayed from the same seed for the oracle."""
    from denoising_diffusion_pytorch import GaussianDiffusion
    sd = load("diffusion.pt")["deblur/net_sd"]
    c = load("variants.pt")["denoise/fab"]
    net = _net(mbe, sd)
    d = GaussianDiffusion(net, image_size=16, channels=3, timesteps=c["T"], sampling_routine="x0_step_down").to(mbe.device)
    x = mbe.to(c["x"])
    torch.manual_seed(11)
    noise = torch.randn_like(x)
    torch.manual_seed(11)
    F1, B1, i1 = d.forward_and_backward(batch_size=2, img=x)
    ca, cb = O.cosine_tables(c["T"])
    rF, rB, ri = O.noise_forward_and_backward(lambda im, st: O.unet_forward(sd, im, st), c["x"], noise.cpu(), c["T"], ca, cb)
    _lists_close(F1, rF, 1e-6), _lists_close(B1, rB, 1e-4), _lists_close([i1], [ri], 1e-4)


def test_resolution_sampler_variants_golden(mbe):
    """all_sample, forward_and_backward, gen_sample(times=) (RESOL:460-617) against the reference's own outputs."""
    from resolution_diffusion_pytorch import GaussianDiffusion
    g = load("variants.pt")
    net = _net(mbe, load("diffusion.pt")["deblur/net_sd"])
    for key, c in g.items():
        if not key.startswith("resolution/Incremental"):
            continue
        _, routine, sampling = key.split("/")
        d = GaussianDiffusion(net, image_size=16, device_of_kernel="cuda", channels=3, timesteps=c["T"], resolution_routine=routine,
                              sampling_routine=sampling)
        x = mbe.to(c["x"])
        X0, Xt = d.all_sample(batch_size=2, img=x)
        _lists_close(X0, c["all_sample"][0], 1e-4), _lists_close(Xt, c["all_sample"][1], 1e-4)
        F1, B1, i1 = d.forward_and_backward(batch_size=2, img=x)
        _lists_close(F1, c["fab"][0], 0.0 if "_area" in routine else 1e-5)
        _lists_close(B1, c["fab"][1], 1e-4), _lists_close([i1], [c["fab"][2]], 1e-4)
        _lists_close(list(d.gen_sample(batch_size=2, img=x, times=2)), list(c["gen_times2"]), 1e-4)


def test_resolution_train_routines(mbe):
    """RESOL:655-760.  Deterministic routines against the reference's losses and gradients; the ones that draw random
    numbers against the oracle fed with the same draw (replayed from the seed on the product's device)."""
    from resolution_diffusion_pytorch import GaussianDiffusion
    g = load("variants.pt")
    sd = load("diffusion.pt")["deblur/net_sd"]
    sizes = O.pixelate_sizes("Incremental_factor_2", 3, 16)
    for tr in ("Final", "Step", "Final_small_noise", "Final_random_mean", "Final_random_mean_and_actual", "Gradient_norm"):
        for loss_type in ("l1", "l2"):
            if mbe.kind == "emu" and loss_type == "l2" and tr != "Final":     # keep the CPU suite short: every routine once + one l2
                continue
            c = g[f"resolution/train/{'Final' if tr == 'Gradient_norm' else tr}/{loss_type}"]
            net = _net(mbe, sd)
            d = GaussianDiffusion(net, image_size=16, device_of_kernel="cuda", channels=3, timesteps=3, loss_type=loss_type,
                                  resolution_routine="Incremental_factor_2", train_routine=tr)
            x, t = mbe.to(c["x"]), mbe.to(c["t"])
            torch.manual_seed(5)
            noise = torch.randn_like(x)
            torch.manual_seed(5)
            new_mean = torch.randn_like(torch.mean(x, [2, 3]))
            torch.manual_seed(5)
            loss = d.p_losses(x, t)
            loss.backward()
            if tr in ("Final", "Step"):
                ref_loss, ref_grads = c["loss"], c["grads"]
            else:
                ps = {k: (v.clone().requires_grad_() if v.is_floating_point() else v) for k, v in sd.items()}
                ref_loss = O.pixelate_p_losses(lambda im, st: O.unet_forward(ps, im, st), c["x"], c["t"], sizes, "bicubic", tr, loss_type,
                                               noise=noise.cpu(), new_mean=new_mean.cpu())
                ref_loss.backward()
                ref_grads = {k: v.grad for k, v in ps.items() if v.is_floating_point() and v.grad is not None}
                ref_loss = ref_loss.detach()
            assert abs(loss.item() - ref_loss.item()) <= 1e-5 * max(1.0, abs(ref_loss.item())), (tr, loss_type)
            if ref_grads is not None:
                grad_check([(k, p) for k, p in net.named_parameters() if k in ref_grads], ref_grads)
    d = GaussianDiffusion(net, image_size=16, device_of_kernel="cuda", channels=3, timesteps=3, resolution_routine="Incremental_factor_2",
                          train_routine="Step")
    with pytest.raises(RuntimeError):                     # every t = 0: the reference stacks an empty list (RESOL:641)
        d.p_losses(x, torch.zeros_like(t))


def test_demixing_golden(mbe):
    """demixing_diffusion_pytorch: q_sample, sample, gen_sample, forward_and_backward, all_sample, loss + grads."""
    from demixing_diffusion_pytorch import GaussianDiffusion
    c = load("mixing.pt")["demix"]
    net = _net(mbe, load("diffusion.pt")["deblur/net_sd"])
    d = GaussianDiffusion(net, image_size=16, channels=3, timesteps=c["T"]).to(mbe.device)
    x1, x2, t = mbe.to(c["x1"]), mbe.to(c["x2"]), mbe.to(c["t"])
    with torch.no_grad():
        assert torch.equal(d.q_sample(x1, x2, t).cpu(), c["q"])
        _lists_close(list(d.gen_sample(batch_size=2, img=x2, noise_level=0)), list(c["gen"]), 1e-4)
        _lists_close(list(d.sample(batch_size=2, img=x2)), list(c["sample"]), 1e-4)
        F1, B1, i1 = d.forward_and_backward(batch_size=2, img1=x1, img2=x2)
        _lists_close(F1, c["fab"][0], 1e-6), _lists_close(B1, c["fab"][1], 1e-4), _lists_close([i1], [c["fab"][2]], 1e-4)
        X0, Xt = d.all_sample(batch_size=2, img=x2)
        _lists_close(X0, c["all_sample"][0], 1e-4), _lists_close(Xt, c["all_sample"][1], 1e-4)
    loss = d.p_losses(x1, x2, t)
    loss.backward()
    assert abs(loss.item() - c["loss"].item()) <= 1e-5
    grad_check(net.named_parameters(), c["grads"])


def test_defading_generation_golden(mbe):
    """The defading-generation package (per-pixel mask blend into a solid colour)."""
    import importlib
    import sys
    pkg_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cold-diffusion-models_amd", "defading_generation")
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "defading_diffusion_pytorch" or k.startswith("defading_diffusion_pytorch.")}
    sys.path.insert(0, pkg_dir)
    try:
        GaussianDiffusion = importlib.import_module("defading_diffusion_pytorch").GaussianDiffusion
    finally:
        sys.path.remove(pkg_dir)
        for k in [k for k in sys.modules if k == "defading_diffusion_pytorch" or k.startswith("defading_diffusion_pytorch.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    g = load("mixing.pt")
    for key in ("defgen/0", "defgen/1"):
        c = g[key]
        net = _net(mbe, load("diffusion.pt")["deblur/net_sd"])
        d = GaussianDiffusion(net, image_size=16, channels=3, timesteps=c["T"], reverse=key.endswith("1"), kernel_std=c["kernel_std"],
                              initial_mask=c["initial_mask"]).to(mbe.device)
        assert torch.equal(d.alphas.cpu(), c["alphas"]) and torch.equal(d.one_minus_alphas.cpu(), c["one_minus"])
        x1, x2, t = mbe.to(c["x1"]), mbe.to(c["x2"]), mbe.to(c["t"])
        with torch.no_grad():
            assert torch.equal(d.q_sample(x1, x2, t).cpu(), c["q"])               # same two products and one sum per pixel
            _lists_close(list(d.sample(batch_size=2, img=x2)), list(c["sample"]), 1e-4)
            _lists_close(list(d.gen_sample(batch_size=2, img=x2, noise_level=0)), list(c["gen"]), 1e-4)
            F1, B1, i1 = d.forward_and_backward(batch_size=2, img1=x1, img2=x2)
            _lists_close(F1, c["fab"][0], 0.0), _lists_close(B1, c["fab"][1], 1e-4), _lists_close([i1], [c["fab"][2]], 1e-4)
            X0, Xt = d.all_sample(batch_size=2, img=x2)
            _lists_close(X0, c["all_sample"][0], 1e-4), _lists_close(Xt, c["all_sample"][1], 1e-4)
        loss = d.p_losses(x1, x2, t)
        loss.backward()
        assert abs(loss.item() - c["loss"].item()) <= 1e-5
        if c["grads"] is not None:
            grad_check(net.named_parameters(), c["grads"])


def test_trainer_train_loop_with_milestone(tmp_path):
    """(simulator only: host logic) Trainer.train() end to end (DEBLUR:1183-1235): optimizer steps, EMA, and at the milestone step the Algorithm-2 sample of
    the EMA model, the four PNG grids and the checkpoint, for a one-image package (deblurring) and the defading one (whose
    sampler takes `faded_recon_sample=`)."""
    from deblurring_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    import defading_diffusion_pytorch as DF
    from colddiff import runtime
    from emu_util import install_emu
    install_emu()
    mbe = MBE("emu")
    torch.manual_seed(0)
    try:
        _train_loop_cases(mbe, tmp_path, GaussianDiffusion, Trainer, Unet, DF)
    finally:
        runtime._lib_override = None


def _train_loop_cases(mbe, tmp_path, GaussianDiffusion, Trainer, Unet, DF):
    for name, make in (("deblur", lambda net: GaussianDiffusion(net, image_size=16, device_of_kernel='cuda', channels=3, timesteps=3, kernel_std=0.5,
                                                                  kernel_size=3, blur_routine='Incremental', sampling_routine='x0_step_down')),
                       ("defade", lambda net: DF.GaussianDiffusion(net, image_size=16, device_of_kernel='cuda', channels=3, timesteps=3,
                                                                   kernel_std=0.6, initial_mask=1, fade_routine='Incremental',
                                                                   sampling_routine='x0_step_down'))):
        net = quiet(Unet, dim=8, dim_mults=(1, 2), channels=3).to(mbe.device)
        d = make(net).to(mbe.device)
        res = tmp_path / name
        tr = quiet(Trainer, d, None, dataset='synthetic', image_size=16, train_batch_size=2, train_num_steps=3, save_and_sample_every=2,
                   results_folder=str(res))
        tr.quiet = True
        quiet(tr.train)
        assert tr.step == 3
        for f in ("sample-og-1.png", "sample-recon-1.png", "sample-direct_recons-1.png", "sample-xt-1.png", "model.pt"):
            assert (res / f).exists(), (name, f)
        ck = torch.load(str(res / "model.pt"), map_location="cpu", weights_only=False)
        assert ck["step"] == 2 and set(ck) == {"step", "model", "ema"}


def test_two_image_trainers(mbe, tmp_path):
    """Trainer variants of the forward(x1, x2) packages: demixing draws the second image from a second dataset (DEMIX:724-726),
    defading generation from uniform random colours (DEFGEN:769-773); one optimizer step each on synthetic data."""
    from demixing_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    from colddiff.diffusion import DefadeGenDiffusion
    from colddiff.trainer import DefadeGenTrainer
    torch.manual_seed(0)
    net = quiet(Unet, dim=8, dim_mults=(1, 2), channels=3).to(mbe.device)
    d = GaussianDiffusion(net, image_size=16, channels=3, timesteps=4).to(mbe.device)
    tr = quiet(Trainer, d, None, None, dataset='synthetic', image_size=16, train_batch_size=2, results_folder=str(tmp_path / "a"))
    a, b = tr._next_batch(), tr._second(None)
    assert a.shape == b.shape == (2, 3, 16, 16) and not torch.equal(a, b)           # two independent image streams
    w0 = tr.arena.data.clone()
    assert torch.isfinite(tr.train_step()).item() and not torch.equal(tr.arena.data, w0)
    net2 = quiet(Unet, dim=8, dim_mults=(1, 2), channels=3).to(mbe.device)
    d2 = DefadeGenDiffusion(net2, image_size=16, channels=3, timesteps=4, kernel_std=0.3, initial_mask=2).to(mbe.device)
    tr2 = quiet(DefadeGenTrainer, d2, None, dataset='synthetic', image_size=16, train_batch_size=2, results_folder=str(tmp_path / "b"))
    c = tr2._second(tr2._next_batch())
    assert c.shape == (2, 3, 16, 16) and float(c.min()) >= -0.5 and float(c.max()) < 0.5
    assert torch.equal(c, c[:, :, :1, :1].expand_as(c))                              # one colour per (sample, channel)
    assert torch.isfinite(tr2.train_step()).item()


def test_time_of_one_row_broadcasts_over_the_image_batch(mbe):
    """DEBLUR:160 adds the time condition by torch broadcasting, so a [1] time against B images is legal upstream (the GMM scripts call
    all_sample(1, imgs), DENOISE:1203); this engine indexes its per-sample tables by image row -- the [1] time must be expanded, not read
    past its end -- and any other mismatch must fail like the broadcast would."""
    from deblurring_diffusion_pytorch import Model, Unet
    torch.manual_seed(2)
    x = mbe.to(torch.randn(3, 3, 16, 16))
    for net in (quiet(Unet, dim=8, dim_mults=(1, 2), channels=3), Model(resolution=16, in_channels=3, out_ch=3, ch=32, ch_mult=(1, 2), num_res_blocks=1,
                                                                        attn_resolutions=(8,), dropout=0.0)):
        net = net.to(mbe.device).eval()
        with torch.no_grad():
            t1 = mbe.to(torch.tensor([2.0]))
            assert torch.equal(net(x, t1), net(x, t1.expand(3).contiguous()))
            with pytest.raises(RuntimeError, match="must match the size"):
                net(x, mbe.to(torch.tensor([1.0, 2.0])))


def test_step_vector_of_one_row_broadcasts_in_the_degradation_ops(mbe):
    """... and the same for the schedule lookups (`extract(a, t, x_shape)` upstream): all_sample(1, imgs) hands [1]-row steps to B-image launches."""
    from colddiff import degrade as D
    torch.manual_seed(4)
    x0, eps = mbe.to(torch.randn(3, 3, 8, 8)), mbe.to(torch.randn(3, 3, 8, 8))
    ca, cb = mbe.to(torch.rand(5)), mbe.to(torch.rand(5))
    t1 = mbe.to(torch.tensor([3]))
    assert torch.equal(D.noise_qsample(x0, eps, ca, cb, t1), D.noise_qsample(x0, eps, ca, cb, t1.expand(3).contiguous()))
    al = mbe.to(torch.rand(5, 1, 8, 8))
    assert torch.equal(D.blend_qsample(x0, eps, al, 1 - al, t1), D.blend_qsample(x0, eps, al, 1 - al, t1.expand(3).contiguous()))
    with pytest.raises(RuntimeError, match="must match the size"):
        D.noise_qsample(x0, eps, ca, cb, mbe.to(torch.tensor([1, 2])))


def test_generation_scripts_of_the_two_image_packages(mbe, tmp_path):
    """The generation / evaluation scripts the denoising, demixing and defading-generation Trainers share (colddiff.evaluate.GenEvalMixin;
    DENOISE:821-854, 1091-1395 and the whitespace-identical copies): every saved image is the sampler's output for the seeds the script
    draws -- replayed here from the same RNG state --, folders and file names as upstream."""
    import os
    from PIL import Image
    from denoising_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    from colddiff.diffusion import DefadeGenDiffusion
    from colddiff.trainer import DefadeGenTrainer
    S = 16
    folder = tmp_path / "imgs"
    folder.mkdir()
    g = torch.Generator().manual_seed(5)
    for i in range(6):
        Image.fromarray((torch.rand(S, S, 3, generator=g) * 255).to(torch.uint8).numpy()).save(folder / f"{i}.png")
    torch.manual_seed(0)
    net = quiet(Unet, dim=8, dim_mults=(1, 2), channels=3).to(mbe.device)
    d = GaussianDiffusion(net, image_size=S, channels=3, timesteps=3).to(mbe.device)
    tr = quiet(Trainer, d, str(folder), image_size=S, train_batch_size=2, results_folder=str(tmp_path / "den"), device_data=False, num_workers=0)
    saved = []
    tr._save = lambda img, name, nrow=6: saved.append((os.path.relpath(name, tmp_path), img.detach().cpu().clone()))
    # --- sample_and_save_for_fid: N(0, 1) seeds through gen_sample (DENOISE:835-843)
    torch.manual_seed(21)
    assert quiet(tr.sample_and_save_for_fid, num_samples=4, bs=2) == 4
    torch.manual_seed(21)
    want = []
    for _ in range(2):
        seeds = torch.randn(2, 3, S, S).to(mbe.device)
        want.append(quiet(tr.ema_core.gen_sample, batch_size=2, img=seeds)[2].cpu())
    want = torch.cat(want)
    assert [n for n, _ in saved] == [f"den_out/sample-x0-{i}.png" for i in range(4)]
    assert all(torch.equal(im[0], want[i]) for i, (_, im) in enumerate(saved))
    # --- sample_from_data_save: dataset rows start < idx <= end through all_sample, last x0 saved (DENOISE:1362-1395)
    del saved[:]
    assert quiet(tr.sample_from_data_save, start=0, end=3) == 3
    rows = torch.stack([tr._dataset_item(i) for i in (1, 2, 3)])
    x0s = quiet(tr.ema_core.all_sample, batch_size=3, img=rows, times=None)[0][-1].cpu()        # (X1_0s of (X1_0s, X2_0s, X_ts))
    assert [n for n, _ in saved] == [f"den/sample-x0-{i}.png" for i in range(3)] and all(torch.equal(im[0], x0s[i]) for i, (_, im) in enumerate(saved))
    # --- the GMM scripts: fit on the resampled dataset vectors, sample, blow up, all_sample (DENOISE:1161-1213, 1215-1286)
    del saved[:]
    assert quiet(tr.sample_as_a_vector_gmm_and_save, start=-1, end=5, siz=4, clusters=2, n_sample=4, num_samples=2) == 4
    assert [n for n, _ in saved] == [f"den_4_2/sample-x0-{i}.png" for i in range(4)] and all(im.shape == (1, 3, S, S) and torch.isfinite(im).all() for _, im in saved)
    feats = []

    class CaptureGMM:
        def __init__(self, **kw):
            self.kw = kw

        def fit(self, x):
            feats.append(x.detach().cpu().clone())

        def sample(self, num_datapoints):
            return torch.linspace(-1, 1, num_datapoints * 3 * 16).reshape(num_datapoints, -1)

    del saved[:]
    assert quiet(tr.sample_as_a_vector_pytorch_gmm_and_save, CaptureGMM, start=-1, end=5, siz=4, clusters=2, n_sample=2, num_samples=2) == 2
    import torch.nn.functional as F
    ref_feats = torch.stack([F.interpolate(tr._dataset_item(i).unsqueeze(0), size=4, mode='bilinear').flatten(1)[0] for i in range(6)]).cpu()
    assert torch.equal(feats[0], ref_feats)
    assert [n for n, _ in saved][:3] == ["den_4_2/sample-x0-0.png", "den_gmm_4_2/sample-0.png", "den_gmm_blur_4_2/sample-blur-0.png"]
    X0, Xt = quiet(tr.sample_as_a_vector_gmm, start=-1, end=5, siz=4, clusters=2, num_samples=2)
    assert len(X0) >= 1 and X0[-1].shape == (2, 3, S, S)
    # --- defading generation: one random colour per image as the seed (DEFGEN:880-891)
    net2 = quiet(Unet, dim=8, dim_mults=(1, 2), channels=3).to(mbe.device)
    d2 = DefadeGenDiffusion(net2, image_size=S, channels=3, timesteps=3, kernel_std=0.3, initial_mask=2).to(mbe.device)
    tr2 = quiet(DefadeGenTrainer, d2, None, dataset='synthetic', image_size=S, train_batch_size=2, results_folder=str(tmp_path / "gen"))
    saved2 = []
    tr2._save = lambda img, name, nrow=6: saved2.append(img.detach().cpu().clone())
    torch.manual_seed(4)
    assert quiet(tr2.sample_and_save_for_fid, noise=0, num_samples=2) == 2
    torch.manual_seed(4)
    seeds = (torch.rand((2, 3)) - 0.5)[:, :, None, None].expand(2, 3, S, S).to(mbe.device).contiguous()
    want2 = quiet(tr2.ema_core.gen_sample, batch_size=2, img=seeds, noise_level=0)[2].cpu()
    assert all(torch.equal(im[0], want2[i]) for i, im in enumerate(saved2))


def test_defading_trainer_test_methods(mbe, tmp_path):
    """The defading package's own Trainer test methods (DEFADE:814-940, 1146-1244: `all_sample` / `sample` take `faded_recon_sample=`
    there): what they save is what the samplers return for the batches they draw."""
    import os
    from defading_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    torch.manual_seed(0)
    net = quiet(Unet, dim=8, dim_mults=(1, 2), channels=3).to(mbe.device)
    d = GaussianDiffusion(net, image_size=16, device_of_kernel="cuda", channels=3, timesteps=3, kernel_std=0.6, initial_mask=1,
                          fade_routine="Incremental", sampling_routine="x0_step_down").to(mbe.device)
    tr = quiet(Trainer, d, None, dataset='synthetic', image_size=16, train_batch_size=2, results_folder=str(tmp_path / "r"))
    saved = {}
    tr._save = lambda img, name, nrow=6: saved.__setitem__(os.path.relpath(name, tmp_path), img.detach().cpu().clone())
    x0, xt = quiet(tr.test_from_data, "a")
    assert len(x0) >= 1 and torch.equal(saved["r/sample-0-a-x0.png"], x0[0].cpu()) and "r/og-a.png" in saved
    x0, xt = quiet(tr.test_from_random, "b")
    assert float(saved["r/og-b.png"].abs().max()) <= 0.9 + 1e-6 and torch.equal(saved["r/sample-0-b-xt.png"], xt[0].cpu())
    x0, xt = quiet(tr.test_with_mixup, "c")
    assert torch.allclose(saved["r/og-c.png"], (saved["r/og1-c.png"] + saved["r/og2-c.png"]) / 2) and torch.equal(saved["r/sample-0-c-x0.png"], x0[0].cpu())
    xt, direct, img = quiet(tr.controlled_direct_reconstruct, "d")
    assert torch.equal(saved["r/sample-recon-d.png"], img.cpu()) and torch.equal(saved["r/sample-xt-d.png"], xt.cpu()) and os.path.exists(tmp_path / "r" / "model.pt")


def test_defading_golden(mbe):
    from defading_diffusion_pytorch import GaussianDiffusion
    g = load("diffusion.pt")
    net = _net(mbe, g["deblur/net_sd"])
    for sampling in ("default", "x0_step_down"):
        c = g[f"defade/{sampling}"]
        d = GaussianDiffusion(net, image_size=16, device_of_kernel="cuda", channels=3, timesteps=c["T"], kernel_std=0.6, initial_mask=1,
                              fade_routine="Incremental", sampling_routine=sampling)
        assert torch.equal(d.fade_kernels, c["masks"])
        with torch.no_grad():
            assert torch.equal(d.q_sample(mbe.to(c["x"]), mbe.to(c["t"])).cpu(), c["q"])      # sequential products: bit-exact
            xt, _, img = d.sample(batch_size=3, faded_recon_sample=mbe.to(c["x"]))
        assert torch.equal(xt.cpu(), c["xt"]) and (img.cpu() - c["img"]).abs().max() <= 1e-4
    # Random_Incremental: per-sample crop offsets, checked against the oracle with the same offsets
    torch.manual_seed(3)
    d = GaussianDiffusion(net, image_size=16, device_of_kernel="cuda", channels=3, timesteps=4, kernel_std=0.5, initial_mask=1,
                          fade_routine="Random_Incremental", discrete=True)
    x, t = torch.rand(3, 3, 16, 16) * 2 - 1, torch.tensor([3, 0, 2])
    torch.manual_seed(11)
    q = d.q_sample(mbe.to(x), mbe.to(t)).cpu()
    torch.manual_seed(11)
    rx = torch.randint(0, 17, (3,), device=mbe.device).cpu()
    ry = torch.randint(0, 17, (3,), device=mbe.device).cpu()
    assert torch.equal(q, O.fade_q_sample(x, t, d.fade_kernels.cpu(), rx, ry, discrete=True))


def test_random_incremental_fade_128_q_sample_golden(mbe):
    """Defading 'Random_Incremental' (+- discrete) at 128 x 128, README schedule (README.md:125-126): q_sample bit-exact against the
    reference-generated tests/golden/fullsize.pt with the reference's crop offsets replayed (DEFADE:496-535).  (The six-step sampler
    of the same fixture runs on the MI355X: tests/test_gpu_fullsize.py.)"""
    from defading_diffusion_pytorch import GaussianDiffusion
    from test_oracle import masks_from_g1
    g = load("fullsize.pt")
    ref_masks = masks_from_g1(g["defade128/g1d"])
    for key, c in g.items():
        if key.endswith("g1d"):
            continue
        d = GaussianDiffusion(torch.nn.Identity(), image_size=128, device_of_kernel="cuda", channels=3, timesteps=c["T"], kernel_std=c["kernel_std"],
                              initial_mask=c["initial_mask"], fade_routine="Random_Incremental", discrete=key.endswith("/1"))
        assert (d.fade_kernels.cpu() - ref_masks).abs().max() <= 2.4e-7   # this host's exp may differ from the reference host's by an ulp
        d.fade_kernels = ref_masks.clone()                    # the masks are data from here on (as the blur kernels are state_dict entries)
        d._offsets = lambda b, dev, c=c: (c["rand_x"].to(dev), c["rand_y"].to(dev))
        x = mbe.to(c["levels"].float() / 255 * 2 - 1)
        with torch.no_grad():
            assert torch.equal(d.q_sample(x, mbe.to(c["t"])).cpu(), c["q"]), key


def test_extras_golden_discrete_individual_and_random_fades(mbe):
    """Reference-generated vectors (tests/golden/make_golden.py: extra_cases) for the branches that round 1 only compared with the
    oracle: deblurring `discrete=True` (DEBLUR:413-415, 441-444, 937-940, 954-958), blur_routine 'Individual_Incremental'
    (DEBLUR:380-383, 402-403, 427-428), defading 'Constant' / 'Random_Incremental' (+discrete) with the reference's own seeded crop
    offsets replayed (DEFADE:359-368, 501-516)."""
    from deblurring_diffusion_pytorch import GaussianDiffusion
    g = load("extras.pt")
    net = _net(mbe, load("diffusion.pt")["deblur/net_sd"])
    for key, c in g.items():
        if not key.startswith("deblur/"):
            continue
        _, routine, sampling = key.split("/")
        d = GaussianDiffusion(net, image_size=16, device_of_kernel="cuda", channels=3, timesteps=c["T"], kernel_std=c["std"],
                              kernel_size=c["ks"], blur_routine=routine, sampling_routine=sampling, discrete=c["discrete"]).to(mbe.device)
        for m, w in zip(d.gaussian_kernels, c["kernels"]):
            assert torch.equal(m.weight.detach().cpu(), w), key
        with torch.no_grad():
            if routine != "Individual_Incremental":          # (its kernel sizes differ per step: q_sample's conv stack is still uniform upstream)
                q = d.q_sample(mbe.to(c["x"]), mbe.to(c["t"])).cpu()
                if c["discrete"]:
                    # int() truncation to 8-bit levels: a 1-ulp difference before the floor may move a pixel by one level
                    lv = ((q - c["q"]).abs() * 127.5).round()
                    assert lv.max() <= 1 and (lv > 0).float().mean() <= 2e-3, (key, lv.max(), (lv > 0).float().mean())
                else:
                    assert (q - c["q"]).abs().max() <= 2e-6, key
            xt, direct, img = quiet(d.sample, batch_size=3, img=mbe.to(c["x"]))
        assert (xt.cpu() - c["xt"]).abs().max() <= 2e-6, key
        assert (direct.cpu() - c["direct"]).abs().max() <= 1e-4 and (img.cpu() - c["img"]).abs().max() <= 1e-4, key
    from defading_diffusion_pytorch import GaussianDiffusion as Defade
    for key, c in g.items():
        if not key.startswith("defade/"):
            continue
        _, routine, discrete, sampling = key.split("/")
        d = Defade(net, image_size=16, device_of_kernel="cuda", channels=3, timesteps=c["T"], kernel_std=0.5, initial_mask=1,
                   fade_routine=routine, sampling_routine=sampling, discrete=bool(int(discrete)))
        assert torch.equal(d.fade_kernels, c["masks"]), key
        if "Random" in routine:                               # the reference's draws (CPU generator) replayed on any device
            d._offsets = lambda b, dev, c=c: (c["rand_x"].to(dev), c["rand_y"].to(dev))
        with torch.no_grad():
            assert torch.equal(d.q_sample(mbe.to(c["x"]), mbe.to(c["t"])).cpu(), c["q"]), key          # bit-exact (products + truncation)
            xt, direct, img = d.sample(batch_size=3, faded_recon_sample=mbe.to(c["x"]))
        assert torch.equal(xt.cpu(), c["xt"]), key
        assert (direct.cpu() - c["direct"]).abs().max() <= 1e-4 and (img.cpu() - c["img"]).abs().max() <= 1e-4, key
    from resolution_diffusion_pytorch import GaussianDiffusion as Resol
    for key, c in g.items():
        if not key.startswith("resolution/"):
            continue
        d = Resol(net, image_size=16, device_of_kernel="cuda", channels=3, timesteps=c["T"], resolution_routine=key.split("/")[1],
                  sampling_routine="x0_step_down")
        with torch.no_grad():
            assert (d.q_sample(mbe.to(c["x"]), mbe.to(c["t"])).cpu() - c["q"]).abs().max() <= 1e-5, key
            xt, _, img = quiet(d.sample, batch_size=3, img=mbe.to(c["x"]))
        assert (xt.cpu() - c["xt"]).abs().max() <= 1e-5 and (img.cpu() - c["img"]).abs().max() <= 1e-4, key


def test_gaussian_taps_follow_torchgeometry_fp32_exp():
    """torchgeometry 0.1.2 image/gaussian.py evaluates exp IN fp32 on the fp32-rounded exponent (torch.exp(torch.tensor(.))), tap by
    tap.  Independent restatement with numpy float32 scalars + the SURVEY 8(c) probe values; fp64-exp-then-round (round 1) differs
    in 2-8 taps per kernel for sigma = 0.35 / 0.84 / 7."""
    import numpy as np
    from colddiff import degrade as D
    for k, s in ((11, 7.0), (11, 0.35), (11, 0.84), (15, 1.0), (5, 0.2), (3, 0.4)):
        e = np.array([np.exp(np.float32(-(x - k // 2) ** 2 / float(2 * s ** 2))) for x in range(k)], dtype=np.float32)
        g1 = torch.from_numpy(e) / torch.from_numpy(e).sum()
        ref = torch.matmul(g1.unsqueeze(-1), g1.unsqueeze(-1).t())
        got = D.gaussian_kernel2d((k, k), (s, s))
        assert got.dtype == torch.float32 and (got - ref).abs().max() <= 4e-7 * ref.max(), (k, s)       # numpy vs torch fp32 exp: a few ulp apart at most
        assert torch.equal(got, O.gaussian_kernel2d((k, k), (s, s)))
    k = D.gaussian_kernel2d((11, 11), (7.0, 7.0))
    assert abs(k[5, 5].item() - 0.0100549823) < 1e-9 and abs(k[0, 0].item() - 0.0060367407) < 1e-9
    assert abs(D.gaussian_kernel2d((15, 15), (1.0, 1.0))[7, 7].item() - 0.1591549516) < 2e-8


def test_trainer_matches_oracle(mbe, tmp_path):
    """3 optimizer steps (2 micro-steps each, fused Adam over the flat arena, EMA copy phase) and the
    checkpoint round trip, against the oracle's torch.optim.Adam loop."""
    from denoising_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    torch.manual_seed(0)
    net = quiet(Unet, dim=8, dim_mults=(1, 2), channels=3).to(mbe.device)
    diff = GaussianDiffusion(net, image_size=8, channels=3, timesteps=10, sampling_routine="x0_step_down").to(mbe.device)
    wrapped = torch.nn.DataParallel(diff, device_ids=[0]) if mbe.kind == "hip" else diff
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    tr = Trainer(wrapped, None, image_size=8, train_batch_size=2, train_lr=2e-5, train_num_steps=3, gradient_accumulate_every=2,
                 dataset="synthetic", results_folder=str(tmp_path / "res"))
    g = torch.Generator().manual_seed(1)
    batches = [[(torch.rand(2, 3, 8, 8, generator=g) * 2 - 1, torch.randn(2, 3, 8, 8, generator=g), torch.randint(0, 10, (2,), generator=g))
                for _ in range(2)] for _ in range(3)]
    ca, cb = O.cosine_tables(10)
    otr = O.OracleTrainer(sd0, lambda p, x, e, t: O.loss_fn(x, O.unet_forward(p, O.noise_q_sample(x, e, t, ca, cb), t)), lr=2e-5, accumulate=2)
    for s in range(3):
        it = iter(batches[s])
        tr._loss = lambda batch, it=it: tr.core.p_losses(*[mbe.to(v) for v in next(it)])
        loss = tr.train_step()
        tr.step += 1
        lo = otr.train_step(batches[s])
        assert abs(loss.item() - lo) <= 1e-5
        for k in sd0:
            assert (net.state_dict()[k].cpu() - otr.params[k].detach()).abs().max() <= 1e-6, k
    ema_sd = tr.ema_core.denoise_fn.state_dict()
    for k in sd0:      # step 0 EMA = copy of the weights after step 0; later steps (1, 2) do not touch it (update every 10)
        assert (ema_sd[k].cpu() - otr.ema[k]).abs().max() <= 1e-6
    tr.save()
    ck = torch.load(str(tmp_path / "res" / "model.pt"), map_location="cpu", weights_only=False)
    prefix = "module." if mbe.kind == "hip" else ""
    assert ck["step"] == 3 and f"{prefix}denoise_fn.time_mlp.1.weight" in ck["model"] and f"{prefix}alphas_cumprod" in ck["ema"]
    tr.load(str(tmp_path / "res" / "model.pt"))
    # the authors' checkpoints carry DataParallel's `module.` prefix (or not): both load into either kind of model
    flip = (lambda sd: {k[len("module."):]: v for k, v in sd.items()}) if prefix else (lambda sd: {"module." + k: v for k, v in sd.items()})
    torch.save({"step": 7, "model": flip(ck["model"]), "ema": flip(ck["ema"])}, str(tmp_path / "res" / "other.pt"))
    before = {k: v.clone() for k, v in tr.model.state_dict().items()}
    tr.load(str(tmp_path / "res" / "other.pt"))
    assert tr.step == 7 and all(torch.equal(v, tr.model.state_dict()[k]) for k, v in before.items())


def test_fused_accumulation_matches_oracle_micro_steps(mbe, tmp_path):
    """Trainer's fused gradient accumulation (the micro-batches of an optimizer step through the network as ONE batch) against the
    oracle's loop of separate (loss_i / 2).backward() micro-steps (DEBLUR:1188-1195): loss and every weight after each of 3 steps."""
    from denoising_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    torch.manual_seed(0)
    net = quiet(Unet, dim=8, dim_mults=(1, 2), channels=3).to(mbe.device)
    diff = GaussianDiffusion(net, image_size=8, channels=3, timesteps=10, sampling_routine="x0_step_down").to(mbe.device)
    sd0 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    tr = Trainer(diff, None, image_size=8, train_batch_size=2, train_lr=2e-5, train_num_steps=3, gradient_accumulate_every=2,
                 dataset="synthetic", results_folder=str(tmp_path / "res"))
    assert tr._can_fuse()
    g = torch.Generator().manual_seed(1)
    batches = [[(torch.rand(2, 3, 8, 8, generator=g) * 2 - 1, torch.randn(2, 3, 8, 8, generator=g), torch.randint(0, 10, (2,), generator=g))
                for _ in range(2)] for _ in range(3)]
    ca, cb = O.cosine_tables(10)
    otr = O.OracleTrainer(sd0, lambda p, x, e, t: O.loss_fn(x, O.unet_forward(p, O.noise_q_sample(x, e, t, ca, cb), t)), lr=2e-5, accumulate=2)
    for s in range(3):
        it = iter(batches[s])

        def micro(it=it):
            x, e, t = (mbe.to(v) for v in next(it))
            return tr.core.prepare(x, e, t=t)
        tr._prepare_micro = micro
        loss = tr.train_step()
        tr.step += 1
        lo = otr.train_step(batches[s])
        assert abs(loss.item() - lo) <= 1e-5
        for k in sd0:
            assert (net.state_dict()[k].cpu() - otr.params[k].detach()).abs().max() <= 1e-6, k


@pytest.mark.parametrize("package", ["denoising", "deblurring", "defading", "resolution"])
def test_fused_accumulation_same_draws_same_trajectory(mbe, tmp_path, monkeypatch, package):
    """Fused and unfused accumulation consume the data loader and the RNG in the same order (batch, second image, t, offsets per
    micro-batch), so from one seed they follow the same trajectory: losses and weights after 3 steps agree to fp32 summation order."""
    import importlib
    pkg = importlib.import_module(package + "_diffusion_pytorch")
    out = []
    for fuse in ("0", "1"):
        monkeypatch.setenv("COLDDIFF_FUSE_ACCUM", fuse)
        torch.manual_seed(3)
        net = quiet(pkg.Unet, dim=8, dim_mults=(1, 2), channels=3).to(mbe.device)
        kw = dict(image_size=16, channels=3, timesteps=6)
        if package == "deblurring":
            kw.update(device_of_kernel="cuda", kernel_size=3, kernel_std=0.5, blur_routine="Incremental")
        elif package == "defading":
            kw.update(device_of_kernel="cuda", kernel_std=0.5, fade_routine="Random_Incremental")
        elif package == "resolution":
            kw.update(device_of_kernel="cuda", timesteps=3, resolution_routine="Incremental_factor_2")
        d = pkg.GaussianDiffusion(net, **kw).to(mbe.device)
        tr = pkg.Trainer(d, None, image_size=16, train_batch_size=2, train_lr=1e-3, train_num_steps=3, gradient_accumulate_every=2,
                         dataset="synthetic", results_folder=str(tmp_path / ("res" + fuse)))
        assert tr._can_fuse() == (fuse == "1")
        torch.manual_seed(5)
        losses = []
        for _ in range(3):
            losses.append(tr.train_step().item())
            tr.step += 1
        out.append((losses, {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}))
    for a, b in zip(out[0][0], out[1][0]):
        assert abs(a - b) <= 2e-6 * max(1.0, abs(a)), (out[0][0], out[1][0])
    for k, v in out[0][1].items():
        # three Adam steps of lr 1e-3: an element whose gradient is at the rounding level may move by a fraction of lr either way
        assert (v - out[1][1][k]).abs().max() <= 2e-4, k
        assert ((v - out[1][1][k]).abs() > 2e-6).float().mean() <= 0.02, k


def test_fused_accumulation_falls_back_when_memory_runs_out(mbe, tmp_path):
    """FUSE_MAX_PIXELS counts pixels, not model width -- if the one fused pass does not fit where the reference's
    micro-step loop does, the step is rerun unfused on the SAME prepared micro-batches (same data / t / noise draws) and fusion stays
    off for this Trainer: loss and weights equal the plain loop's."""
    from denoising_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    out = []
    for oom in (False, True):
        torch.manual_seed(0)
        net = quiet(Unet, dim=8, dim_mults=(1, 2), channels=3).to(mbe.device)
        diff = GaussianDiffusion(net, image_size=8, channels=3, timesteps=10).to(mbe.device)
        tr = Trainer(diff, None, image_size=8, train_batch_size=2, train_lr=1e-3, train_num_steps=2, gradient_accumulate_every=2,
                     dataset="synthetic", results_folder=str(tmp_path / f"res{int(oom)}"))
        if oom:
            real = tr.core.loss_prepared

            def starved(prep):
                if prep[0].shape[0] > 2:                      # the 4-image fused batch "does not fit"
                    raise torch.OutOfMemoryError("simulated")
                return real(prep)
            tr.core.loss_prepared = starved
        else:
            tr._fuse_off = True                               # the plain loop, same draws
        torch.manual_seed(5)
        losses = []
        for _ in range(2):
            losses.append(quiet(tr.train_step).item())
            tr.step += 1
        assert tr._fuse_off and not tr._can_fuse()
        out.append((losses, tr.arena.data.clone()))
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])


def test_no_cpu_fallback():
    """Without the test-only simulator override the operators refuse CPU tensors."""
    from colddiff import runtime
    from deblurring_diffusion_pytorch import Unet
    assert runtime._lib_override is None
    net = quiet(Unet, dim=8, dim_mults=(1, 2), channels=3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.randn(1, 3, 8, 8), torch.tensor([1]))


def test_packed_cache_drops_dead_params(mbe):
    """A packed weight must not outlive its parameter: a new tensor can reuse both id() and the address."""
    import gc
    from colddiff import ops
    gc.collect()
    w = torch.nn.Parameter(mbe.to(torch.randn(8, 4, 3, 3)))
    n0 = len(ops._pack_cache)
    ops.packed(w, "conv_fwd")
    assert len(ops._pack_cache) == n0 + 1
    del w
    gc.collect()
    assert len(ops._pack_cache) == n0


def test_convnext_block_presplit_operands(mbe):
    """Wide ConvNeXt block: the 3x3 convs run on pre-split bf16 hi/lo planes (fwd, dgrad and wgrad).
    Forward and every gradient must still meet the fp32 parity bound against the oracle (DEBLUR:156-165)."""
    from colddiff.unet import ConvNextBlock
    from colddiff import functions as F_
    from oracle import cold_oracle as O
    assert F_.want_presplit(128, 128, 3)
    torch.manual_seed(3)
    blk = ConvNextBlock(128, 128, time_emb_dim=16, mult=1)
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.mul_(2.0)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in blk.state_dict().items()}
    blk = blk.to(mbe.device)
    x = torch.randn(2, 128, 32, 32)
    temb = torch.randn(2, 16)
    g = torch.randn(2, 128, 32, 32)
    xr = x.clone().requires_grad_(True)
    yr = O.convnext_block(sd, xr, temb)
    yr.backward(g)
    xd = mbe.to(x).requires_grad_(True)
    xn = F_.ToNHWC.apply(xd)
    y = F_.ToNCHW.apply(blk(xn, mbe.to(torch.nn.functional.gelu(temb))), 128, None)
    y.backward(mbe.to(g))
    tol = 1e-4
    assert (y.detach().cpu() - yr.detach()).abs().max().item() <= tol * yr.abs().max().item()
    assert (xd.grad.cpu() - xr.grad).abs().max().item() <= tol * xr.grad.abs().max().item()
    for n, p_ in blk.named_parameters():
        r = sd[n].grad
        assert (p_.grad.cpu() - r).abs().max().item() <= tol * max(r.abs().max().item(), 1e-3), n


@pytest.mark.parametrize("kind", ["conv", "convT"])
def test_resampling_conv_presplit(mbe, kind):
    """4x4 stride-2 down-sampling conv / transposed up-sampling conv (DEBLUR:125-130) on the pre-split LDS-DMA path:
    forward, data gradient and weight gradient against torch."""
    from colddiff import functions as F_
    from colddiff.unet import anchor
    assert F_.want_presplit(64, 64, 4)
    torch.manual_seed(5)
    mod = (torch.nn.Conv2d(64, 64, 4, 2, 1) if kind == "conv" else torch.nn.ConvTranspose2d(64, 64, 4, 2, 1))
    ref = (torch.nn.Conv2d(64, 64, 4, 2, 1) if kind == "conv" else torch.nn.ConvTranspose2d(64, 64, 4, 2, 1))
    ref.load_state_dict(mod.state_dict())
    mod = mod.to(mbe.device)
    x = torch.randn(2, 64, 12, 12)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    g = torch.randn_like(yr)
    yr.backward(g)
    xd = mbe.to(x).requires_grad_(True)
    xn = F_.ToNHWC.apply(xd)
    y = F_.ToNCHW.apply(F_.ConvFn.apply(anchor(xn), xn, mod, 64, kind, 2, (1, 1, 1, 1)), 64, None)
    y.backward(mbe.to(g))
    tol = 1e-4
    assert (y.detach().cpu() - yr.detach()).abs().max().item() <= tol * yr.abs().max().item()
    assert (xd.grad.cpu() - xr.grad).abs().max().item() <= tol * xr.grad.abs().max().item()
    for (n, p_), (_, r) in zip(mod.named_parameters(), ref.named_parameters()):
        assert (p_.grad.cpu() - r.grad).abs().max().item() <= tol * max(r.grad.abs().max().item(), 1e-3), n


def test_resnet_block_dropout_in_training(mbe, monkeypatch):
    """ResnetBlock in training mode with dropout 0.25 (MODEL2:94,124): the mask is the engine's own counter-based stream (not torch's
    Philox), so the check replays it -- cdf_dropout on ones with the block's seed -- into a torch restatement of the
    block: forward and every gradient.  Covers GroupNorm + SiLU + dropout + operand split fused in one pass, planes-only and not."""
    from colddiff.model2 import ResnetBlock
    from colddiff import functions as F_
    from colddiff import ops
    import torch.nn.functional as F
    torch.manual_seed(11)
    monkeypatch.setattr(F_, "_seed", lambda: 424242)
    for lean in (True, False):
        monkeypatch.setattr(F_, "_LEAN", lean)
        blk = ResnetBlock(in_channels=64, out_channels=128, dropout=0.25, temb_channels=32).train()
        sd = {k: v.detach().clone().requires_grad_(True) for k, v in blk.state_dict().items()}
        blk = blk.to(mbe.device)
        x, temb, g = torch.randn(8, 64, 16, 16), torch.randn(8, 32), torch.randn(8, 128, 16, 16)
        mask = ops.dropout(torch.ones(8, 16, 16, 128, device=mbe.device), 0.25, 424242).cpu().permute(0, 3, 1, 2)   # 0 or 1 / (1 - p)
        assert 0.6 < float((mask != 0).float().mean()) < 0.9
        sw = lambda v: v * torch.sigmoid(v)
        xr = x.clone().requires_grad_(True)
        h = F.conv2d(sw(F.group_norm(xr, 32, sd['norm1.weight'], sd['norm1.bias'], eps=1e-6)), sd['conv1.weight'], sd['conv1.bias'], padding=1)
        h = h + F.linear(sw(temb), sd['temb_proj.weight'], sd['temb_proj.bias'])[:, :, None, None]
        h = sw(F.group_norm(h, 32, sd['norm2.weight'], sd['norm2.bias'], eps=1e-6)) * mask
        yr = F.conv2d(xr, sd['nin_shortcut.weight'], sd['nin_shortcut.bias']) + F.conv2d(h, sd['conv2.weight'], sd['conv2.bias'], padding=1)
        yr.backward(g)
        xd = mbe.to(x).requires_grad_(True)
        y = F_.ToNCHW.apply(blk(F_.ToNHWC.apply(xd), mbe.to(sw(temb))), 128, None)
        y.backward(mbe.to(g))
        tol = 1e-4
        assert (y.detach().cpu() - yr.detach()).abs().max().item() <= tol * yr.abs().max().item()
        assert (xd.grad.cpu() - xr.grad).abs().max().item() <= tol * xr.grad.abs().max().item()
        for n, p_ in blk.named_parameters():
            r = sd[n].grad
            assert (p_.grad.cpu() - r).abs().max().item() <= tol * max(r.abs().max().item(), 1e-3), (lean, n)


def test_resnet_block_presplit_operands(mbe):
    """CIFAR `Model` ResnetBlock (MODEL2:114-133) with its 3x3 convs on pre-split bf16 planes: forward and every
    gradient against the oracle at the fp32 parity bound."""
    from colddiff.model2 import ResnetBlock
    from colddiff import functions as F_
    from oracle import cold_oracle as O
    assert F_.want_presplit(64, 128, 3) and F_.want_presplit(128, 128, 3)
    torch.manual_seed(7)
    blk = ResnetBlock(in_channels=64, out_channels=128, dropout=0.0, temb_channels=32)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in blk.state_dict().items()}
    blk = blk.to(mbe.device)
    x, temb, g = torch.randn(8, 64, 16, 16), torch.randn(8, 32), torch.randn(8, 128, 16, 16)
    xr = x.clone().requires_grad_(True)
    yr = O.resnet_block(sd, xr, temb)
    yr.backward(g)
    xd = mbe.to(x).requires_grad_(True)
    xn = F_.ToNHWC.apply(xd)
    sw = mbe.to(temb * torch.sigmoid(temb))
    y = F_.ToNCHW.apply(blk(xn, sw), 128, None)
    y.backward(mbe.to(g))
    tol = 1e-4
    assert (y.detach().cpu() - yr.detach()).abs().max().item() <= tol * yr.abs().max().item()
    assert (xd.grad.cpu() - xr.grad).abs().max().item() <= tol * xr.grad.abs().max().item()
    for n, p_ in blk.named_parameters():
        r = sd[n].grad
        assert (p_.grad.cpu() - r).abs().max().item() <= tol * max(r.abs().max().item(), 1e-3), n

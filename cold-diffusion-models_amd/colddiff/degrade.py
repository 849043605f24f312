"""Degradation operators D(x, t) and losses as autograd-free torch-facing calls into the HIP library.

Init-time constants (Gaussian taps, fade masks, cosine schedule) are generated here on the host
exactly as the reference generates them; everything applied to images runs in csrc/k_degrade.hip.
"""
import math

import torch

from . import runtime as rt
from .runtime import P

PAD_MODES = {"circular": 0, "reflect": 1}
PIX_MODES = {"area": 0, "bilinear": 1, "bicubic": 2}


# -- torchgeometry.image.get_gaussian_kernel2d (0.1.x) restated: deblurring_diffusion_pytorch.py:348-349 ------
def gaussian_1d(ksize, sigma):
    # torchgeometry 0.1.2 image/gaussian.py `gaussian()`: the exponent is rounded to fp32 by torch.tensor() and the
    # exponential is evaluated IN fp32 (torch.exp), one tap at a time; then normalised in fp32.
    g = torch.stack([torch.exp(torch.tensor(-(x - ksize // 2) ** 2 / float(2 * sigma ** 2))) for x in range(ksize)])
    return g / g.sum()


def gaussian_kernel2d(ksize, sigma):
    if isinstance(ksize, int):
        ksize = (ksize, ksize)
    if not isinstance(sigma, (tuple, list)):
        sigma = (sigma, sigma)
    kx, ky = gaussian_1d(ksize[0], sigma[0]), gaussian_1d(ksize[1], sigma[1])
    return torch.matmul(kx.unsqueeze(-1), ky.unsqueeze(-1).t())


def _img(x):
    rt.check(x)
    return x.contiguous().float()


def _steps(t, B):
    """The per-sample step table of a B-image launch.  Upstream indexes its schedules by broadcasting (`extract(a, t, x_shape)`,
    `t[:, None, None, None]`), so a [1] step vector against B images is legal there -- the GMM scripts call all_sample(1, imgs)
    (DENOISE:1203) -- while the kernels here read t[b] per image: expanded to B rows (int64, contiguous), any other mismatch fails as the
    broadcast would."""
    if t is None:
        return None
    if t.dim() == 0:
        t = t.reshape(1)
    if t.shape[0] != B:
        if t.shape[0] != 1:
            raise RuntimeError(f"The size of tensor a ({B}) must match the size of tensor b ({t.shape[0]}) at non-singleton dimension 0")
        t = t.expand(B)
    return t.to(torch.int64).contiguous()


def blur_chain(x, taps, k, pad_mode, t=None, step_lo=0, step_hi=0, img=None, want_prev=False, collapse_step=-1, quantise=False,
               taps1d=None):
    """Apply blur steps step_lo..hi(b) (hi = t[b] or step_hi) with the plane resident in LDS.
    Returns y (or the Alg.2 combination img - D_hi + D_{hi-1} when img is given) [, D_{hi-1}].
    taps1d ([T, C, 2, k], see separable_taps) selects the separable kernel: 2k instead of k*k FMAs per pixel."""
    x = _img(x)
    B, C, H, W = x.shape
    t = _steps(t, B)
    y = torch.empty_like(x)
    snap = torch.empty_like(x) if want_prev else None
    img = None if img is None else _img(img)     # keep the (possibly converted) tensor alive over the launch
    L = rt.lib()
    if taps1d is not None and k <= 64 and k // 2 < min(H, W) and L.cdf_blur_sep_lds_bytes(H, W) <= 160 * 1024:
        L.cdf_blur_chain_sep(P(x), P(y), P(snap), P(img), P(taps1d), P(t), B, C, H, W, k, step_lo, step_hi,
                             pad_mode, collapse_step, 1 if quantise else 0, rt.stream(x))
    else:
        L.cdf_blur_chain(P(x), P(y), P(snap), P(img), P(taps), P(t), B, C, H, W, k, step_lo, step_hi,
                         pad_mode, collapse_step, 1 if quantise else 0, rt.stream(x))
    return (y, snap) if want_prev else y


def separable_taps(taps):
    """[T, C, k, k] kernel stack -> [T, C, 2, k] (factor along y, factor along x) if EVERY kernel is an outer product of
    its row sums and column sums to fp32 rounding (true for the reference's Gaussians g (x) g, whose taps sum to 1:
    DEBLUR:363-389), else None.  The weights are state_dict data, so this is checked, not assumed."""
    w = taps.double()
    tot = w.sum((-1, -2), keepdim=True)
    if not bool((tot.abs() > 1e-12).all()):
        return None
    gy, gx = w.sum(-1), w.sum(-2) / tot.squeeze(-1)           # w ~ gy[:, None] * gx[None, :]
    resid = (w - gy.unsqueeze(-1) * gx.unsqueeze(-2)).abs().amax((-1, -2))
    if not bool((resid <= 1e-7 * w.abs().amax((-1, -2))).all()):
        return None
    return torch.stack([gy, gx], dim=-2).float().contiguous()


def blur_fits_lds(H, W, k):
    return W % 4 == 0 and rt.lib().cdf_blur_lds_bytes(H, W, k) <= 160 * 1024


def blur_step(x, taps_c, k, pad_mode):
    x = _img(x)
    B, C, H, W = x.shape
    y = torch.empty_like(x)
    rt.lib().cdf_blur_step(P(x), P(y), P(taps_c), B, C, H, W, k, pad_mode, rt.stream(x))
    return y


def plane_mean_(x):
    B, C, H, W = x.shape
    rt.lib().cdf_plane_mean(P(x), B * C, H * W, rt.stream(x))
    return x


def mask_chain(x, masks, t=None, step_lo=0, step_hi=0, img=None, off_y=None, off_x=None, quantise=False):
    x = _img(x)
    B, C, H, W = x.shape
    t = _steps(t, B)
    off_y, off_x = _steps(off_y, B), _steps(off_x, B)
    y = torch.empty_like(x)
    snap = torch.empty_like(x) if img is not None else None
    img = None if img is None else _img(img)
    rt.lib().cdf_mask_chain(P(x), P(y), P(snap), P(img), P(masks), P(t), P(off_y), P(off_x), B, C, H, W,
                            masks.shape[1], masks.shape[2], step_lo, step_hi, 1 if quantise else 0, rt.stream(x))
    return y


def pixelate_chain(x, sizes, mode, t=None, step_lo=0, step_hi=0, img=None):
    x = _img(x)
    B, C, H, W = x.shape
    t = _steps(t, B)
    assert H == W, "the resolution operator works on square images (as the reference asserts)"
    y = torch.empty_like(x)
    snap = torch.empty_like(x) if img is not None else None
    img = None if img is None else _img(img)
    rt.lib().cdf_pixelate_chain(P(x), P(y), P(snap), P(img), P(sizes), P(t), B, C, H, step_lo, step_hi,
                                mode, rt.stream(x))
    return y


def x0_step_down(img, d_t, d_tm1):
    img, d_t, d_tm1 = _img(img), _img(d_t), _img(d_tm1)
    out = torch.empty_like(img)
    rt.lib().cdf_x0_step_down(P(img), P(d_t), P(d_tm1), P(out), img.numel(), rt.stream(img))
    return out


def noise_qsample(x0, eps, ca, cb, t):
    x0, eps = _img(x0), _img(eps)
    t = _steps(t, x0.shape[0])
    out = torch.empty_like(x0)
    rt.lib().cdf_noise_qsample(P(x0), P(eps), P(ca), P(cb), P(t), P(out), x0.shape[0], x0[0].numel(), rt.stream(x0))
    return out


def noise_step(img, x1, noise, ca, cb, t, est_noise):
    img, x1 = _img(img), _img(x1)
    noise = None if noise is None else _img(noise)
    out = torch.empty_like(img)
    rt.lib().cdf_noise_step(P(img), P(x1), P(noise), P(ca), P(cb), int(t), 1 if est_noise else 0, P(out), img.numel(), rt.stream(img))
    return out


def blend_qsample(x1, x2, alphas, one_minus, t):
    """alphas[t[b]] * x1 + one_minus[t[b]] * x2 with per-pixel tables [T, 1, H, W] (or [T, H, W])."""
    x1, x2 = _img(x1), _img(x2)
    B, C, H, W = x1.shape
    t = _steps(t, B)
    out = torch.empty_like(x1)
    rt.lib().cdf_blend_qsample(P(x1), P(x2), P(alphas), P(one_minus), P(t), P(out), B, C, H * W, rt.stream(x1))
    return out


def blend_step(img, x1, x2, alphas, one_minus, t):
    """img - q(x1, x2, t-1) + q(x1, x2, t-2) (x1 itself when t = 1) with the second image held fixed."""
    img, x1, x2 = _img(img), _img(x1), _img(x2)
    out = torch.empty_like(img)
    rt.lib().cdf_blend_step(P(img), P(x1), P(x2), P(alphas), P(one_minus), int(t), P(out), img.shape[2] * img.shape[3], img.numel(),
                            rt.stream(img))
    return out


class _Loss(torch.autograd.Function):
    """mean |x_start - x_recon| (l1) or mean squared error (l2); gradient flows to x_recon only."""

    @staticmethod
    def forward(ctx, x_start, x_recon, l2):
        xs, xr = _img(x_start), _img(x_recon)
        out = torch.empty((1,), device=xs.device, dtype=torch.float32)
        part = torch.empty((1024,), device=xs.device, dtype=torch.float32)
        rt.lib().cdf_loss_fwd(P(xs), P(xr), P(out), P(part), xs.numel(), l2, rt.stream(xs))
        ctx.l2 = l2
        ctx.save_for_backward(xs, xr)
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        xs, xr = ctx.saved_tensors
        g = g.contiguous().float().view(1)
        gy = torch.empty_like(xr)
        rt.lib().cdf_loss_bwd(P(xs), P(xr), P(g), P(gy), xs.numel(), ctx.l2, rt.stream(xs))
        return None, gy, None


def loss(x_start, x_recon, loss_type):
    if loss_type == "l1":
        return _Loss.apply(x_start, x_recon, 0)
    if loss_type == "l2":
        return _Loss.apply(x_start, x_recon, 1)
    raise NotImplementedError()

"""The metric step after sampling: RMSE, SSIM and FID as the reference's evaluation code computes them
(deblurring_diffusion_pytorch.py:1677-1702 with Fid/fid_score.py:149-343 and pytorch_msssim.ssim).

* rmse / ssim run on the MI355X (cdf_loss_fwd, cdf_ssim_partial); there is no CPU fallback.
* FID = Frechet distance between the Gaussians fitted to InceptionV3 activations: the network is colddiff.inception.InceptionV3 (the
  reference's Fid/inception.py on the HIP kernels); its pretrained `pt_inception-2015-12-05` weights are a download upstream and are
  read from a local file here ($COLDDIFF_FID_WEIGHTS / torch hub cache).  Any other feature extractor
  `model(batch [B,3,H,W] in [0,1]) -> [B, dims]` (or a list of maps) can be passed as `model=`.
"""
import ctypes
import math

import numpy as np
import torch

from . import degrade as D
from . import runtime as rt
from .runtime import P


def rmse(a, b):
    """torch.sqrt(torch.mean((a - b) ** 2)) (DEBLUR:1678)."""
    return torch.sqrt(D.loss(rt.check(a).float().contiguous(), b.to(a.device).float().contiguous(), 'l2'))


def _gauss_window(size=11, sigma=1.5):
    """pytorch_msssim._fspecial_gauss_1d: exp(-x^2 / 2 sigma^2) on coords - size//2, normalised, fp32."""
    coords = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def ssim(X, Y, data_range=255, size_average=True, win_size=11, win_sigma=1.5, K=(0.01, 0.03)):
    """pytorch_msssim.ssim(X, Y, data_range, size_average) for [B, C, H, W] batches, one fused kernel."""
    assert X.shape == Y.shape and X.dim() == 4, "ssim takes two [B, C, H, W] batches of the same shape"
    assert win_size == 11, "the HIP kernel is specialised for the 11-tap window the reference uses"
    X = rt.check(X).float().contiguous()
    Y = Y.to(X.device).float().contiguous()
    B, C, H, W = X.shape
    L = rt.lib()
    tiles = L.cdf_ssim_tiles(H, W)
    partial = torch.empty((B * C, tiles), device=X.device, dtype=torch.float32)
    win = (ctypes.c_float * 11)(*_gauss_window(win_size, win_sigma).tolist())
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    L.cdf_ssim_partial(P(X), P(Y), P(partial), B * C, H, W, win, C1, C2, rt.stream(X))
    per_channel = partial.sum(1).view(B, C) / float((H - win_size + 1) * (W - win_size + 1))
    return per_channel.mean() if size_average else per_channel.mean(1)


# -- FID (Fid/fid_score.py) ------------------------------------------------------------------------------------------
def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """d^2 = |mu1 - mu2|^2 + Tr(C1 + C2 - 2 sqrt(C1 C2)), the numerically careful form of Fid/fid_score.py:149-200
    (singular product -> eps on the diagonals; a small imaginary part from sqrtm is dropped, a large one is an error)."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape, 'Training and test mean vectors have different lengths'
    assert sigma1.shape == sigma2.shape, 'Training and test covariances have different dimensions'
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        print('fid calculation produces singular product; adding %s to diagonal of cov estimates' % eps)
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError('Imaginary component {}'.format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


def get_activations(samples, model, batch_size=50, dims=2048, device='cuda:0'):
    """[N, dims] float64 activations of `model` over `samples` ([N, 3, H, W] in [0, 1]) (Fid/fid_score.py get_activations;
    like the reference, a trailing partial batch IS processed: drop_last=False)."""
    n = samples.shape[0]
    out = np.empty((n, dims))
    for s in range(0, n, batch_size):
        with torch.no_grad():
            pred = model(samples[s:s + batch_size].to(device))
        if isinstance(pred, (list, tuple)):
            pred = pred[0]
        if pred.dim() == 4:                                   # not yet pooled: global spatial average (dims != 2048 in the reference)
            pred = pred.mean((2, 3))
        out[s:s + pred.shape[0]] = pred.reshape(pred.shape[0], -1).double().cpu().numpy()
    return out


def calculate_activation_statistics(samples, model, batch_size=50, dims=2048, device='cuda:0'):
    act = get_activations(samples, model, batch_size, dims, device)
    return np.mean(act, axis=0), np.cov(act, rowvar=False)


def calculate_fid_given_samples(samples, batch_size=50, device='cuda:0', dims=2048, num_workers=1, model=None):
    """FID of two sample collections `samples = [A, B]` (Fid/fid_score.py:331-343).  `model`: a feature extractor to use instead of
    InceptionV3([BLOCK_INDEX_BY_DIM[dims]])."""
    if model is None:
        from .inception import InceptionV3
        model = InceptionV3([InceptionV3.BLOCK_INDEX_BY_DIM[dims]]).to(device)          # raises FileNotFoundError without the weight file
    m1, s1 = calculate_activation_statistics(samples[0], model, batch_size, dims, device)
    m2, s2 = calculate_activation_statistics(samples[1], model, batch_size, dims, device)
    return calculate_frechet_distance(m1, s1, m2, s2)

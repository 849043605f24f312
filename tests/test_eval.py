"""evaluation samplers around gen_sample / all_sample and the metric step (RMSE, SSIM, FID)."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_data import _write_images


def ssim_reference(X, Y, data_range=1.0, win_size=11, win_sigma=1.5, K=(0.01, 0.03)):
    """pytorch_msssim 0.2.x `ssim(X, Y, data_range, size_average=True)` restated with torch ops (the package is not installed;
    cited by DEBLUR:1569, 1679): _fspecial_gauss_1d, gaussian_filter = grouped conv along H then along W, _ssim."""
    coords = torch.arange(win_size, dtype=torch.float32) - win_size // 2
    g = torch.exp(-(coords ** 2) / (2 * win_sigma ** 2))
    g = (g / g.sum())
    C = X.shape[1]

    def filt(z):
        z = F.conv2d(z, g.view(1, 1, -1, 1).repeat(C, 1, 1, 1), groups=C)
        return F.conv2d(z, g.view(1, 1, 1, -1).repeat(C, 1, 1, 1), groups=C)

    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = filt(X), filt(Y)
    s1, s2, s12 = filt(X * X) - mu1 * mu1, filt(Y * Y) - mu2 * mu2, filt(X * Y) - mu1 * mu2
    cs = (2 * s12 + C2) / (s1 + s2 + C2)
    m = ((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * cs
    return m.flatten(2).mean(-1).mean()


@pytest.fixture(params=[pytest.param("emu"), pytest.param("hip", marks=pytest.mark.gpu)])
def dev(request):
    from colddiff import runtime
    if request.param == "emu":
        from emu_util import install_emu
        install_emu()
        yield torch.device("cpu")
    else:
        runtime._lib_override = None
        yield torch.device("cuda:0")
    runtime._lib_override = None


@pytest.mark.parametrize("shape", [(2, 3, 16, 16), (1, 1, 45, 70), (3, 3, 32, 64)])
def test_ssim_and_rmse(dev, shape):
    from colddiff import metrics
    torch.manual_seed(1)
    X = torch.rand(shape)
    Y = (X + 0.1 * torch.randn(shape)).clamp(0, 1)
    got = metrics.ssim(X.to(dev), Y.to(dev), data_range=1, size_average=True).cpu()
    ref = ssim_reference(X, Y, 1.0)
    assert abs(float(got) - float(ref)) <= 2e-5, (float(got), float(ref))
    assert abs(float(metrics.ssim(X.to(dev), X.to(dev), data_range=1)) - 1.0) <= 1e-6            # identical images
    r = metrics.rmse(X.to(dev), Y.to(dev)).cpu()
    assert abs(float(r) - float(torch.sqrt(torch.mean((X - Y) ** 2)))) <= 1e-6


def test_frechet_distance_and_fid_plumbing():
    from colddiff import metrics
    rng = np.random.RandomState(0)
    a = rng.randn(500, 6) @ rng.randn(6, 6) + 1.0
    b = rng.randn(500, 6) * 0.5 - 2.0
    m1, s1, m2, s2 = a.mean(0), np.cov(a, rowvar=False), b.mean(0), np.cov(b, rowvar=False)
    d = metrics.calculate_frechet_distance(m1, s1, m2, s2)
    # closed form through the symmetric square root: Tr sqrt(C1 C2) = Tr sqrt(C1^1/2 C2 C1^1/2)
    w, v = np.linalg.eigh(s1)
    r1 = (v * np.sqrt(w)) @ v.T
    ev = np.linalg.eigvalsh(r1 @ s2 @ r1)
    ref = ((m1 - m2) ** 2).sum() + np.trace(s1) + np.trace(s2) - 2 * np.sqrt(np.clip(ev, 0, None)).sum()
    assert abs(d - ref) <= 1e-6 * abs(ref)
    assert abs(metrics.calculate_frechet_distance(m1, s1, m1, s1)) <= 1e-6
    # the sample-collection entry point with a stand-in feature extractor (no Inception weights offline)
    feat = lambda x: torch.stack([x.mean((1, 2, 3)), x.std((1, 2, 3)), x[:, 0].mean((1, 2))], 1)
    A, B = torch.rand(40, 3, 8, 8), torch.rand(40, 3, 8, 8) * 0.5
    fid = metrics.calculate_fid_given_samples([A, B], batch_size=16, device='cpu', dims=3, model=feat)
    assert fid > 0 and abs(metrics.calculate_fid_given_samples([A, A], batch_size=16, device='cpu', dims=3, model=feat)) < 1e-8
    with pytest.raises(RuntimeError, match="feature extractor"):
        metrics.calculate_fid_given_samples([A, B], device='cpu')


def test_evaluation_samplers_on_a_folder(tmp_path):
    """The reference Trainer's evaluation methods end to end on a folder of PNGs (simulator kernels): GMM samplers, the
    degrade->restore metric sweep, test_from_data, save_training_data."""
    from colddiff import runtime
    from emu_util import install_emu
    install_emu()
    try:
        from deblurring_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
        folder = str(tmp_path / "imgs")
        _write_images(folder, 8)
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            net = Unet(dim=8, dim_mults=(1, 2), channels=3)
            d = GaussianDiffusion(net, image_size=16, device_of_kernel="cpu", channels=3, timesteps=3, kernel_size=3, kernel_std=0.5,
                                  sampling_routine="x0_step_down")
            tr = Trainer(d, folder, image_size=16, train_batch_size=4, train_num_steps=1, dataset="train", results_folder=str(tmp_path / "res"),
                         num_workers=0, device_data=True)
            res = tr.fid_distance_decrease_from_manifold(fid_func=None, start=-1, end=5, batch=4)
            n = tr.sample_as_a_mean_blur_torch_gmm_ablation(None, ch=3, clusters=2, noise=0.001, num_samples=4, bs=2)
            tr.sample_as_a_mean_blur_torch_gmm(None, clusters=2, num_samples=4, noise_levels=(0.001,), repeats=1)
            xt, dr, img = tr.sample_as_a_blur_torch_gmm(None, siz=2, ch=3, clusters=2, sample_at=1, num_samples=4)
            X0, Xt = tr.test_from_data("t")
            tr.save_training_data()
        assert set(res) == {f"{m}_{k}" for m in ("rmse", "ssim") for k in ("blur", "deblur", "direct_deblur")}
        assert all(np.isfinite(v) for v in res.values()) and 0 < res["ssim_blur"] <= 1 and res["rmse_blur"] > 0
        assert n == 4 and len(os.listdir(str(tmp_path / "res") + "_out")) == 4
        assert img.shape == (4, 3, 16, 16) and torch.isfinite(img).all()
        assert len(X0) == 4 and len(Xt) == 3
        names = os.listdir(str(tmp_path / "res"))
        assert "Gif-t-x0.gif" in names and "sample-recon-1-2-2.png" in names and "7.png" in names and "sample-xt-0.001-0-0.png" in names
    finally:
        runtime._lib_override = None

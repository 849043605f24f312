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
    with pytest.raises(FileNotFoundError, match="pt_inception-2015-12-05"):      # no weight file offline: an error, not a random network
        metrics.calculate_fid_given_samples([A, B], device='cpu')


def test_evaluation_samplers_on_a_folder(tmp_path, dev):
    """The reference Trainer's evaluation methods end to end on a folder of PNGs (emu: simulator kernels, hip: the MI355X with the
    device-side image cache): GMM samplers, the degrade->restore metric sweep, test_from_data, save_training_data."""
    from colddiff import runtime
    try:
        from deblurring_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
        folder = str(tmp_path / "imgs")
        _write_images(folder, 8)
        torch.manual_seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            net = Unet(dim=8, dim_mults=(1, 2), channels=3).to(dev)
            d = GaussianDiffusion(net, image_size=16, device_of_kernel=str(dev), channels=3, timesteps=3, kernel_size=3, kernel_std=0.5,
                                  sampling_routine="x0_step_down").to(dev)
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
        pass


def test_evaluation_samplers_vs_reference_golden(dev):
    """tests/golden/evaluation.pt holds what the UNMODIFIED reference Trainer methods computed on a fixed in-memory dataset
    (make_golden.py::evaluation_cases): the channel-mean matrix and the opt()-feature matrix handed to the GMM, the images
    sample_as_a_blur_torch_gmm saves for a fixed GMM sample, and the image sets / RMSE / SSIM of fid_distance_decrease_from_manifold.
    The same methods of this package's Trainer must reproduce them (emu: simulator kernels; hip: the MI355X)."""
    from deblurring_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    G = torch.load(os.path.join(os.path.dirname(__file__), "golden", "evaluation.pt"), weights_only=False)
    sd = torch.load(os.path.join(os.path.dirname(__file__), "golden", "diffusion.pt"), weights_only=False)["deblur/net_sd"]
    cfg, imgs = G["cfg"], G["images"]
    import tempfile
    tmp = tempfile.mkdtemp()
    with contextlib.redirect_stdout(io.StringIO()):
        net = Unet(dim=8, dim_mults=(1, 2), channels=3)
        net.load_state_dict(sd)
        d = GaussianDiffusion(net, image_size=cfg["image_size"], device_of_kernel=str(dev), channels=3, timesteps=cfg["T"],
                              kernel_std=cfg["kernel_std"], kernel_size=cfg["kernel_size"], blur_routine=cfg["blur_routine"],
                              sampling_routine=cfg["sampling_routine"]).to(dev)
        for m, w in zip(d.gaussian_kernels, G["kernels"]):
            assert torch.equal(m.weight.detach().cpu(), w)
        tr = Trainer(d, None, image_size=cfg["image_size"], train_batch_size=4, train_num_steps=1, dataset="synthetic",
                     results_folder=os.path.join(tmp, "res"))

    class ListDS(torch.utils.data.Dataset):
        def __len__(self):
            return imgs.shape[0]

        def __getitem__(self, i):
            return imgs[i]

    tr.ds = ListDS()
    # -- channel means (DEBLUR:1399-1405) ------------------------------------------------------------------------------
    cm = tr._channel_means(100).cpu()
    assert cm.shape == G["channel_means"].shape == (100, 3)
    assert (cm - G["channel_means"]).abs().max() <= 1e-6

    # -- sample_as_a_blur_torch_gmm with the reference's GMM sample replayed (DEBLUR:1514-1564) ---------------------------
    bg = G["blur_gmm"]
    fits = []

    class ReplayGMM:
        def __init__(self, **kw):
            assert kw["num_components"] == bg["clusters"] and kw["covariance_regularization"] == 0.0001 and kw["batch_size"] == 100

        def fit(self, x):
            fits.append(x.detach().cpu().clone())

        def sample(self, num_datapoints):
            return bg["og_x"][:num_datapoints].clone()

    ns = 48 if dev.type == "cuda" else 8                                   # (samples are independent: the simulator replays the first 8 of the 48)
    with contextlib.redirect_stdout(io.StringIO()):
        xt, direct, recon = tr.sample_as_a_blur_torch_gmm(ReplayGMM, siz=bg["siz"], ch=3, clusters=bg["clusters"], sample_at=bg["sample_at"],
                                                          num_samples=ns)
    assert fits[0].shape == bg["feats"].shape and (fits[0] - bg["feats"]).abs().max() <= 1e-5
    for name, got in (("xt", xt), ("direct_recons", direct), ("recon", recon)):
        want = bg["saved"][name][:ns] * 2 - 1                              # the reference saves (img + 1) / 2
        assert (got.cpu() - want).abs().max() <= 2e-4, name

    # -- fid_distance_decrease_from_manifold (DEBLUR:1567-1702) ---------------------------------------------------------------
    sw = G["sweep"]
    calls = []

    def fid_func(samples):
        calls.append([z.detach().cpu().clone() for z in samples])
        return float(len(calls))

    with contextlib.redirect_stdout(io.StringIO()):
        res = tr.fid_distance_decrease_from_manifold(fid_func=fid_func, start=sw["start"], end=sw["end"], batch=32)
    assert len(calls) == 3 and calls[0][0].shape == sw["sets"]["orig"].shape == (40, 3, 16, 16)
    for i, k in enumerate(("blur", "deblur", "direct_deblur")):
        assert (calls[i][0] - sw["sets"]["orig"]).abs().max() <= 1e-6
        assert (calls[i][1] - sw["sets"][k]).abs().max() <= 2e-4, k
        assert abs(res[f"rmse_{k}"] - float(sw["rmse"][k])) <= 2e-5, (k, res[f"rmse_{k}"], float(sw["rmse"][k]))
        assert abs(res[f"ssim_{k}"] - float(sw["ssim"][k])) <= 1e-4, (k, res[f"ssim_{k}"], float(sw["ssim"][k]))
        assert res[f"fid_{k}"] == float(i + 1)

"""Evaluation samplers of the reference's Trainer: host logic around the device samplers.

Mirrors deblurring_diffusion_pytorch.py:1238-1267 (test_from_data), 1391-1456 (sample_as_a_mean_blur_torch_gmm_ablation),
1459-1511 (sample_as_a_mean_blur_torch_gmm), 1514-1564 (sample_as_a_blur_torch_gmm), 1567-1702
(fid_distance_decrease_from_manifold) and 1712-1722 (save_training_data): same method names, arguments, file names.  What they
call on the diffusion model (all_sample / gen_sample / gen_sample_2 / opt / sample_from_blur) runs on the HIP kernels; the
Gaussian-mixture fit is CPU work upstream too (pycave there, scikit-learn's GaussianMixture here behind the same small API).
Image titles / cv2 montages (add_title, the paper_* figure methods) are figure code and stay out of scope.
"""
import os

import torch

from . import metrics
from . import runtime as rt


class GMM:
    """The part of pycave's GMM interface the reference uses (`torch_gmm(num_components=, trainer_params=, covariance_type=,
    convergence_tolerance=, batch_size=[, covariance_regularization=])`, `.fit(tensor)`, `.sample(num_datapoints=)`) on
    sklearn.mixture.GaussianMixture."""

    def __init__(self, num_components=10, trainer_params=None, covariance_type='full', convergence_tolerance=0.001, batch_size=None,
                 covariance_regularization=1e-6, random_state=0):
        from sklearn.mixture import GaussianMixture
        self.model = GaussianMixture(n_components=num_components, covariance_type=covariance_type, tol=convergence_tolerance,
                                     reg_covar=covariance_regularization, random_state=random_state)

    def fit(self, data):
        self.model.fit(data.detach().float().cpu().numpy())
        return self

    def sample(self, num_datapoints):
        x, _ = self.model.sample(num_datapoints)
        return torch.from_numpy(x).float()

    def get_params(self):
        return {"weights": self.model.weights_, "means": self.model.means_, "covariances": self.model.covariances_}


def _create_folder(path):
    os.makedirs(path, exist_ok=True)


class EvalMixin:
    """Evaluation methods of the reference Trainer; mixed into colddiff.trainer.Trainer."""

    # -- dataset access ---------------------------------------------------------------------------------------------
    def _dataset_batches(self, batch_size):
        """The dataset in order in batches of `batch_size` (drop_last=True), as `DataLoader(self.ds, shuffle=False, drop_last=True)`
        yields them (DEBLUR:1396-1397); from the device cache when the folder lives in HBM."""
        from .trainer import DeviceImageCache, DeviceLoader
        assert self.ds is not None, "this evaluation method needs an image folder (the Trainer was built on synthetic data)"
        batch_size = min(batch_size, len(self.ds))            # (upstream hard-codes 100 with drop_last: smaller folders would yield nothing)
        if isinstance(self.ds, DeviceImageCache):
            dl = DeviceLoader(self.ds, batch_size, shuffle=False)        # (the dataset's own chain: DataLoader(self.ds, ...))
            for _ in range(len(self.ds) // batch_size):
                yield next(dl)
            return
        from torch.utils import data
        for img in data.DataLoader(self.ds, batch_size=batch_size, shuffle=False, num_workers=0, drop_last=True):
            yield img.to(self.device)

    def _dataset_item(self, idx):
        from .trainer import DeviceImageCache
        """self.ds[idx] (DEBLUR:1578, 1717): the dataset's own transform chain -- a random crop where the dataset augments."""
        if isinstance(self.ds, DeviceImageCache):
            if self.ds.recipe.crop == 'center' and not self.ds.recipe.flip:
                return self.ds.item(idx)
            from .trainer import DeviceLoader
            dl = DeviceLoader(self.ds, 1, shuffle=False, seed=123457 + idx)
            dl.order, dl.pos = torch.tensor([idx], device=self.ds.data.device), 0
            return next(dl)[0]
        return self.ds[idx].to(self.device)

    def _save(self, img, name, nrow=6):
        from .trainer import save_image
        save_image((img + 1) * 0.5, name, nrow=nrow)

    def _fit_gmm(self, torch_gmm, feats, clusters, batch_size, **extra):
        torch_gmm = GMM if torch_gmm is None else torch_gmm
        model = torch_gmm(num_components=clusters, trainer_params=dict(gpus=1), covariance_type='full', convergence_tolerance=0.001,
                          batch_size=batch_size, **extra)
        model.fit(feats)
        return model

    def _channel_means(self, batch_size=100):
        return torch.cat([torch.mean(img, [2, 3]) for img in self._dataset_batches(batch_size)], dim=0)      # DEBLUR:1399-1405

    # -- DEBLUR:1238-1267 ---------------------------------------------------------------------------------------------
    def test_from_data(self, extra_path, s_times=None):
        og_img = self._next_batch()
        X_0s, X_ts = self.ema_core.all_sample(batch_size=self.batch_size, img=og_img, times=s_times)
        self._save(og_img, str(self.results_folder / f'og-{extra_path}.png'))
        frames_0, frames_t = [], []
        for i in range(len(X_0s)):
            p0, pt = str(self.results_folder / f'sample-{i}-{extra_path}-x0.png'), str(self.results_folder / f'sample-{i}-{extra_path}-xt.png')
            self._save(X_0s[i], p0)
            frames_0.append(p0)
            if i < len(X_ts):                 # all_sample returns one more x0 than x_t (DEBLUR:686); upstream indexes past the end here
                self._save(X_ts[i], pt)
                frames_t.append(pt)
        from PIL import Image
        for name, frames in ((f'Gif-{extra_path}-x0.gif', frames_0), (f'Gif-{extra_path}-xt.gif', frames_t)):
            ims = [Image.open(f).convert('RGB') for f in frames]
            if ims:
                ims[0].save(str(self.results_folder / name), save_all=True, append_images=ims[1:], duration=100, loop=0)
        return X_0s, X_ts

    # -- DEBLUR:1391-1456 ---------------------------------------------------------------------------------------------
    def sample_as_a_mean_blur_torch_gmm_ablation(self, torch_gmm=None, ch=3, clusters=10, noise=0, num_samples=6400, bs=64):
        model = self._fit_gmm(torch_gmm, self._channel_means(100), clusters, 100)
        og_x = model.sample(num_datapoints=num_samples).to(self.device).unsqueeze(2).unsqueeze(3)
        xt_folder, out_folder, dr_folder = f'{self.results_folder}_xt', f'{self.results_folder}_out', f'{self.results_folder}_dir_recons'
        for f in (xt_folder, out_folder, dr_folder):
            _create_folder(f)
        cnt, n = 0, self.image_size
        for j in range(num_samples // bs):
            og_img = og_x[j * bs: j * bs + bs].expand(bs, ch, n, n).float().contiguous()
            xt, direct_recons, all_images = self.ema_core.gen_sample(batch_size=bs, img=og_img, noise_level=noise)
            for i in range(all_images.shape[0]):
                self._save(all_images[i:i + 1], f'{out_folder}/sample-x0-{cnt}.png', nrow=1)
                self._save(xt[i:i + 1], f'{xt_folder}/sample-x0-{cnt}.png', nrow=1)
                self._save(direct_recons[i:i + 1], f'{dr_folder}/sample-x0-{cnt}.png', nrow=1)
                cnt += 1
        return cnt

    # -- DEBLUR:1459-1511 ---------------------------------------------------------------------------------------------
    def sample_as_a_mean_blur_torch_gmm(self, torch_gmm=None, start=0, end=1000, ch=3, clusters=10, num_samples=48,
                                        noise_levels=(0.001, 0.002, 0.003, 0.004), repeats=3):
        model = self._fit_gmm(torch_gmm, self._channel_means(100), clusters, 100)
        n = self.image_size
        og_x = model.sample(num_datapoints=num_samples).to(self.device).unsqueeze(2).unsqueeze(3).expand(num_samples, ch, n, n).float().contiguous()
        i = 0
        for noise in noise_levels:
            for j in range(repeats):
                xt, direct_recons, all_images = self.ema_core.gen_sample_2(batch_size=num_samples, img=og_x, noise_level=noise)
                for name, im in (('og', og_x), ('recon', all_images), ('direct_recons', direct_recons), ('xt', xt)):
                    self._save(im, str(self.results_folder / f'sample-{name}-{noise}-{i}-{j}.png'))

    # -- DEBLUR:1514-1564 ---------------------------------------------------------------------------------------------
    def sample_as_a_blur_torch_gmm(self, torch_gmm=None, siz=4, ch=3, clusters=10, sample_at=1, num_samples=48):
        import torch.nn.functional as F
        feats = []
        for img in self._dataset_batches(100):
            z = self.ema_core.opt(img, t=sample_at)
            feats.append(F.interpolate(z, size=siz, mode='bilinear').flatten(1))     # (tiny host-side resample of the GMM features, as upstream)
        model = self._fit_gmm(torch_gmm, torch.cat(feats, dim=0), clusters, 100, covariance_regularization=0.0001)
        og_x = model.sample(num_datapoints=num_samples).to(self.device).reshape(num_samples, ch, siz, siz)
        og_img = F.interpolate(og_x, size=self.image_size, mode='bilinear').float().contiguous()
        xt, direct_recons, all_images = self.ema_core.sample_from_blur(batch_size=num_samples, img=og_img, start=sample_at)
        for name, im in (('og', og_img), ('recon', all_images), ('direct_recons', direct_recons), ('xt', xt)):
            self._save(im, str(self.results_folder / f'sample-{name}-{sample_at}-{siz}-{clusters}.png'))
        return xt, direct_recons, all_images

    # -- DEBLUR:1567-1702 ---------------------------------------------------------------------------------------------
    def fid_distance_decrease_from_manifold(self, fid_func=None, start=0, end=1000, batch=32):
        """Degrade -> restore every dataset image in (start, end]; RMSE / SSIM (/ FID when fid_func is given) of the degraded, the
        sampled and the directly reconstructed images against the originals.  Returns the numbers the reference prints."""
        items = []
        for idx in range(len(self.ds)):
            if idx > start:
                items.append(self._dataset_item(idx))
            if end is not None and idx == end:
                break
        all_samples = torch.stack(items)
        orig, blurred, deblurred, direct = [], [], [], []
        rep3 = lambda z: z.repeat(1, 3 // z.shape[1], 1, 1)
        cnt = 0
        while cnt < all_samples.shape[0]:
            og_img = all_samples[cnt: cnt + batch].float()
            X_0s, X_ts = self.ema_core.all_sample(batch_size=og_img.shape[0], img=og_img, times=None)
            for dst, z in ((orig, og_img), (blurred, X_ts[0]), (deblurred, X_0s[-1]), (direct, X_0s[0])):
                dst.append((rep3(z.to(self.device)) + 1) * 0.5)
            cnt += og_img.shape[0]
        orig, blurred, deblurred, direct = (torch.cat(z, dim=0) for z in (orig, blurred, deblurred, direct))
        out = {}
        for name, z in (('blur', blurred), ('deblur', deblurred), ('direct_deblur', direct)):
            out[f'rmse_{name}'] = float(metrics.rmse(orig, z))
            out[f'ssim_{name}'] = float(metrics.ssim(orig, z, data_range=1, size_average=True))
            if fid_func is not None:
                out[f'fid_{name}'] = float(fid_func(samples=[orig, z]))
            print(f"The RMSE of {name} images with original image is {out[f'rmse_{name}']}")
            print(f"The SSIM of {name} images with original image is {out[f'ssim_{name}']}")
            if fid_func is not None:
                print(f"The FID of {name} images with original image is {out[f'fid_{name}']}")
        if fid_func is not None:
            print(f"Hence the improvement in FID using sampling is {out['fid_blur'] - out['fid_deblur']}")
            print(f"Hence the improvement in FID using direct sampling is {out['fid_blur'] - out['fid_direct_deblur']}")
        return out

    # -- DEBLUR:1712-1722 ---------------------------------------------------------------------------------------------
    def save_training_data(self):
        _create_folder(f'{self.results_folder}/')
        for idx in range(len(self.ds)):
            self._save(self._dataset_item(idx)[None], f'{self.results_folder}/{idx}.png', nrow=1)


class GenEvalMixin:
    """The generation / evaluation scripts the denoising, demixing and defading-generation Trainers share (whitespace-identical upstream
    except `sample_and_save_for_fid`): denoising_diffusion_pytorch.py:821-854, 1091-1395; demixing_diffusion_pytorch.py:806-836,
    1080-1384; defading-generation .../defading_diffusion_pytorch.py:868-904, 1148-1452.  Same names, arguments, folders and file names;
    the samplers they drive (`gen_sample`, `all_sample`) run on the HIP kernels, the Gaussian-mixture fit is scikit-learn on the host as
    upstream (`GaussianMixture(n_components, random_state=0)`), pycave's `torch_gmm` goes through `EvalMixin._fit_gmm`.  Upstream calls
    `self.ema_model.all_sample` without `.module` (an AttributeError under DataParallel): here `ema_core`.  The hard-coded sample counts
    (6400 / 10000 / 100 per round, batches of 128 / 1000) are the defaults of trailing keyword arguments; titles and GIFs are figure code."""

    def _seed_images(self, bs):
        """The starting images `sample_and_save_for_fid` samples from, one batch (per package)."""
        raise NotImplementedError

    def sample_and_save_for_fid(self, noise=0, num_samples=6400, bs=None):
        bs = self._fid_batch if bs is None else bs
        out_folder = f'{self.results_folder}_out'
        _create_folder(out_folder)
        cnt = 0
        for _ in range(int(num_samples / bs)):
            og_img = self._seed_images(bs)
            xt, direct_recons, all_images = self._gen(bs, og_img, noise)
            for i in range(all_images.shape[0]):
                self._save(all_images[i:i + 1], f'{out_folder}/sample-x0-{cnt}.png', nrow=1)
                cnt += 1
        return cnt

    def _gen(self, bs, og_img, noise):
        return self.ema_core.gen_sample(batch_size=bs, img=og_img, noise_level=noise)

    def _all_sample(self, bs, og_img):
        """(x0 estimates per step, x_t per step).  Upstream unpacks TWO values (`X_0s, X_ts = ...all_sample(...)`, copied from the deblurring
        package) from the THREE the denoising / demixing `all_sample` returns (X1_0s, X2_0s, X_ts: DENOISE:482-515) -- a ValueError there;
        the evident intent is the image estimates and the trajectory."""
        res = self.ema_core.all_sample(batch_size=bs, img=og_img, times=None)
        return res[0], res[-1]

    def _dataset_vectors(self, start, end, siz=None):
        """rows idx with start < idx <= end of the dataset (upstream's `if idx > start` / `if idx == end: break`), resampled to siz x siz
        (bilinear) and flattened when siz is given"""
        import torch.nn.functional as F
        rows = []
        for idx in range(len(self.ds)):
            img = self._dataset_item(idx).unsqueeze(0)
            if siz is not None:
                img = F.interpolate(img, size=siz, mode='bilinear').flatten(1)
            if idx > start:
                rows.append(img[0])
            if end is not None and idx == end:
                break
        return torch.stack(rows)

    def _sk_gmm(self, feats, clusters):
        from sklearn.mixture import GaussianMixture
        return GaussianMixture(n_components=clusters, random_state=0).fit(feats.detach().float().cpu().numpy())

    def _up(self, og_x, n, ch, siz):
        import torch.nn.functional as F
        og_x = torch.as_tensor(og_x).reshape(n, ch, siz, siz).to(self.device).float()
        return F.interpolate(og_x, size=self.image_size, mode='bilinear').contiguous()

    def sample_as_a_vector_gmm(self, start=0, end=1000, siz=64, ch=3, clusters=10, num_samples=100):
        gm = self._sk_gmm(self._dataset_vectors(start, end, siz), clusters)
        og_x, _ = gm.sample(n_samples=num_samples)
        og_img = self._up(og_x, num_samples, ch, siz)
        X_0s, X_ts = self._all_sample(1, og_img)
        extra_path = 'vec'
        self._save(og_img, str(self.results_folder / f'og-{start}-{end}-{siz}-{clusters}-{extra_path}.png'))
        for i in range(len(X_0s)):
            self._save(X_0s[i], str(self.results_folder / f'sample-{start}-{end}-{siz}-{clusters}-{i}-{extra_path}-x0.png'))
            if i < len(X_ts):
                self._save(X_ts[i], str(self.results_folder / f'sample-{start}-{end}-{siz}-{clusters}-{i}-{extra_path}-xt.png'))
        return X_0s, X_ts

    def sample_as_a_vector_gmm_and_save(self, start=0, end=1000, siz=64, ch=3, clusters=10, n_sample=10000, num_samples=10000):
        gm = self._sk_gmm(self._dataset_vectors(start, end, siz), clusters)
        folder = f'{self.results_folder}_{siz}_{clusters}/'
        _create_folder(folder)
        cnt = 0
        for _ in range(int(n_sample / num_samples)):
            og_x, _ = gm.sample(n_samples=num_samples)
            og_img = self._up(og_x, num_samples, ch, siz)
            X_0s, X_ts = self._all_sample(1, og_img)
            x0s = X_0s[-1]
            for i in range(x0s.shape[0]):
                self._save(x0s[i:i + 1], f'{folder}sample-x0-{cnt}.png', nrow=1)
                cnt += 1
        return cnt

    def _torch_gmm_rounds(self, torch_gmm, feats, siz, ch, clusters, n_sample, num_samples, from_blur):
        model = self._fit_gmm(torch_gmm, feats, clusters, 1000)
        f_x0, f_gmm, f_blur = (f'{self.results_folder}_{siz}_{clusters}/', f'{self.results_folder}_gmm_{siz}_{clusters}/',
                               f'{self.results_folder}_gmm_blur_{siz}_{clusters}/')
        for f in (f_x0, f_gmm, f_blur):
            _create_folder(f)
        cnt = 0
        for _ in range(int(n_sample / num_samples)):
            og_x = model.sample(num_datapoints=num_samples)
            og_img = self._up(og_x, num_samples, ch, siz)
            X_0s, X_ts = self._all_sample(og_img.shape[0], og_img)
            x0s, blurs = X_0s[-1], X_ts[0]
            for i in range(x0s.shape[0]):
                self._save(x0s[i:i + 1], f'{f_x0}sample-x0-{cnt}.png', nrow=1)
                self._save(og_img[i:i + 1], f'{f_gmm}sample-{cnt}.png', nrow=1)
                self._save(blurs[i:i + 1], f'{f_blur}sample-blur-{cnt}.png', nrow=1)
                cnt += 1
        return cnt

    def sample_as_a_vector_pytorch_gmm_and_save(self, torch_gmm, start=0, end=1000, siz=64, ch=3, clusters=10, n_sample=10000, num_samples=100):
        return self._torch_gmm_rounds(torch_gmm, self._dataset_vectors(start, end, siz), siz, ch, clusters, n_sample, num_samples, False)

    def sample_as_a_vector_from_blur_pytorch_gmm_and_save(self, torch_gmm, start=0, end=1000, siz=64, ch=3, clusters=10, n_sample=10000,
                                                          num_samples=100):
        # (upstream fits on the same resampled dataset vectors: the "from blur" variant differs from the one above only in its prints)
        return self._torch_gmm_rounds(torch_gmm, self._dataset_vectors(start, end, siz), siz, ch, clusters, n_sample, num_samples, True)

    def sample_from_data_save(self, start=0, end=1000, chunk=1000):
        all_samples = self._dataset_vectors(start, end)                      # [n, C, H, W]
        _create_folder(f'{self.results_folder}/')
        cnt = 0
        while cnt < all_samples.shape[0]:
            og_img = all_samples[cnt: cnt + chunk].to(self.device).float().contiguous()
            X_0s, X_ts = self._all_sample(og_img.shape[0], og_img)
            x0s = X_0s[-1]
            for i in range(x0s.shape[0]):
                self._save(x0s[i:i + 1], f'{self.results_folder}/sample-x0-{cnt}.png', nrow=1)
                cnt += 1
        return cnt

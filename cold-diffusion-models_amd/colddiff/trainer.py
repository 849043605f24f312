"""`Trainer` — the training loop of the cold-diffusion packages on the fused optimizer tail.

Follows deblurring_diffusion_pytorch.py:1057-1235 (and the denoising variant,
denoising_diffusion_pytorch.py:620-789): same constructor, `train() / save() / load() / step_ema()
/ reset_parameters()`, same checkpoint dict (`step`, `model`, `ema` state_dicts with the
DataParallel `module.` prefix when the model was wrapped), gradient accumulation, Adam(lr) defaults,
EMA copy-then-lerp schedule and milestone sampling.  Differences, all performance-only:
  * Adam / EMA / zero_grad are single kernel launches over a flat parameter arena (colddiff.flat)
  * multi-GPU = one process per GPU + RCCL all-reduce (colddiff.parallel) instead of DataParallel
  * the loss is logged without a host sync on every micro-step (the reference calls loss.item() twice)
"""
import copy
import glob
import os
from functools import partial
from pathlib import Path

import torch
from torch.utils import data

from . import flat, parallel
from . import runtime as rt


def cycle(dl, sampler=None):
    """Endless iterator over a DataLoader (DEBLUR:52-55); a DistributedSampler is re-seeded on every wrap-around so that the
    ranks' shards are re-shuffled each epoch."""
    epoch = 0
    while True:
        if sampler is not None and hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(epoch)
        for d in dl:
            yield d
        epoch += 1


def unwrap(model):
    """The reference scripts hand the Trainer a torch.nn.DataParallel wrapper; compute on its module."""
    if isinstance(model, (torch.nn.DataParallel, torch.nn.parallel.DistributedDataParallel)):
        return model.module
    return model


class EMA:
    def __init__(self, beta):
        self.beta = beta

    def update_model_average(self, ma_arena, model_arena):
        flat.ema_update(ma_arena, model_arena, self.beta)


# -- image folder datasets (DEBLUR:983-1026) with PIL only (torchvision is not a dependency) ----------
def _load_image(path, size, augment):
    from PIL import Image
    import numpy as np
    img = Image.open(path)
    big = int(size * 1.12)
    img = img.resize((big, big), Image.BILINEAR)
    if augment:
        ox, oy = (int(torch.randint(0, big - size + 1, (1,))) for _ in range(2))
    else:
        ox = oy = (big - size) // 2          # CenterCrop
    img = img.crop((ox, oy, ox + size, oy + size))
    if augment and torch.rand(1).item() < 0.5:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    arr = np.asarray(img, dtype=np.float32) / 255.0
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(arr).permute(2, 0, 1).contiguous() * 2 - 1


class Dataset(data.Dataset):
    augment = False

    def __init__(self, folder, image_size, exts=('jpg', 'jpeg', 'png')):
        super().__init__()
        self.folder, self.image_size = folder, image_size
        self.paths = [p for ext in exts for p in Path(f'{folder}').glob(f'**/*.{ext}')]

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, index):
        return _load_image(self.paths[index], self.image_size, self.augment)


class Dataset_Aug1(Dataset):
    augment = True


class SyntheticImages:
    """Endless 8-bit-quantised uniform images generated on the device (benchmarks, smoke tests)."""

    def __init__(self, batch_size, channels, image_size, device, seed=123457):
        self.shape = (batch_size, channels, image_size, image_size)
        self.device = device
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)

    def __iter__(self):
        return self

    def __next__(self):
        return torch.randint(0, 256, self.shape, generator=self.gen, device=self.device).float() / 255 * 2 - 1


def save_image(tensor, path, nrow=6):
    """Minimal torchvision.utils.save_image: [B,C,H,W] in [0,1] -> PNG grid."""
    from PIL import Image
    t = tensor.detach().float().clamp(0, 1).cpu()
    B, C, H, W = t.shape
    ncol = min(nrow, B)
    nr = (B + ncol - 1) // ncol
    pad = 2
    grid = torch.zeros(C, nr * (H + pad) + pad, ncol * (W + pad) + pad)
    for i in range(B):
        r, c = divmod(i, ncol)
        grid[:, pad + r * (H + pad): pad + r * (H + pad) + H, pad + c * (W + pad): pad + c * (W + pad) + W] = t[i]
    arr = (grid * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    Image.fromarray(arr[:, :, 0] if C == 1 else arr).save(str(path))


def _match_module_prefix(sd, target_keys):
    """The authors' checkpoints are state_dicts of DataParallel-wrapped models (`module.` on every key: DEBLUR:1140-1157 with
    celebA_128.py:102); the same file must load whether or not THIS model is wrapped (the reference's test scripts carry
    remove_data_parallel / adjust_data_parallel helpers for that, e.g. DEBLUR:1026-1055)."""
    has = len(sd) > 0 and all(k.startswith('module.') for k in sd)
    want = all(k.startswith('module.') for k in target_keys)
    if has and not want:
        return {k[len('module.'):]: v for k, v in sd.items()}
    if want and not has:
        return {'module.' + k: v for k, v in sd.items()}
    return sd


class Trainer(object):
    AUG_DATASETS = ('mnist', 'cifar10', 'flower', 'celebA', 'AFHQ', 'train')

    def __init__(self, diffusion_model, folder, *, ema_decay=0.995, image_size=128, train_batch_size=32, train_lr=2e-5,
                 train_num_steps=100000, gradient_accumulate_every=2, fp16=False, step_start_ema=2000, update_ema_every=10,
                 save_and_sample_every=1000, results_folder='./results', load_path=None, dataset=None, shuffle=True,
                 num_workers=8):
        super().__init__()
        assert not fp16, "apex fp16 is not supported (every reference script passes fp16=False)"
        self.model = diffusion_model
        self.ema = EMA(ema_decay)
        self.ema_model = copy.deepcopy(self.model)
        self.update_ema_every = update_ema_every
        self.step_start_ema = step_start_ema
        self.save_and_sample_every = save_and_sample_every
        self.batch_size = train_batch_size
        self.image_size = image_size
        self.gradient_accumulate_every = gradient_accumulate_every
        self.train_num_steps = train_num_steps
        self.core = unwrap(self.model)
        self.ema_core = unwrap(self.ema_model)
        # the denoising package feeds (image, fresh Gaussian noise) pairs (DENOISE:738-742)
        self.pair_noise = hasattr(self.core, 'sqrt_alphas_cumprod')
        self.device = next(self.core.parameters()).device

        self.ds, self.dl = self._make_loader(folder, dataset, shuffle, num_workers, seed=123457)

        # Adam(diffusion_model.parameters(), lr): one flat arena over every parameter of the model
        self.arena = flat.FlatArena(list(self.core.parameters()))
        self.ema_arena = flat.FlatArena(list(self.ema_core.parameters()))
        self.opt = flat.FusedAdam(self.arena, lr=train_lr)
        self.step = 0
        self.results_folder = Path(results_folder)
        self.results_folder.mkdir(exist_ok=True)
        self.fp16 = fp16
        self.quiet = False

        parallel.init_distributed()
        self.sync = None
        if parallel.world_size() > 1:
            self.sync = parallel.GradSync(self.arena)
            parallel.set_engine(self.sync)
            # every rank starts from rank 0's weights (the reference replicates GPU 0's module)
            torch.distributed.broadcast(self.arena.data, src=0)
            rt.bump_weights_epoch()
            parallel.decorrelate_rng()     # weights are equal now: from here on every rank draws its own t / noise / masks
        self.reset_parameters()
        if load_path is not None:
            self.load(load_path)

    def _make_loader(self, folder, dataset, shuffle, num_workers, seed):
        """(dataset, endless batch iterator): image folder as in the reference (DEBLUR:1094-1096), or synthetic images."""
        if dataset == 'synthetic' or folder is None:
            # (every rank draws its own shard: the seed is offset by the rank)
            return None, SyntheticImages(self.batch_size, self.core.channels, self.image_size, self.device, seed=seed + parallel.rank())
        aug = dataset in self.AUG_DATASETS
        print(dataset, "DA used" if aug else "")
        ds = (Dataset_Aug1 if aug else Dataset)(folder, self.image_size)
        sampler = None
        if parallel.world_size() > 1:
            sampler = data.distributed.DistributedSampler(ds, shuffle=shuffle)
        dl = cycle(data.DataLoader(ds, batch_size=self.batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                                   pin_memory=self.device.type == 'cuda', num_workers=num_workers, drop_last=True), sampler)
        return ds, dl

    # -- EMA / checkpoint --------------------------------------------------------------------------------
    def reset_parameters(self):
        self.ema_model.load_state_dict(self.model.state_dict())
        rt.bump_weights_epoch()

    def step_ema(self):
        if self.step < self.step_start_ema:
            self.reset_parameters()
            return
        self.ema.update_model_average(self.ema_arena, self.arena)

    def save(self, itrs=None):
        if parallel.rank() != 0:
            return
        ckpt = {'step': self.step, 'model': self.model.state_dict(), 'ema': self.ema_model.state_dict()}
        name = 'model.pt' if itrs is None else f'model_{itrs}.pt'
        torch.save(ckpt, str(self.results_folder / name))

    def load(self, load_path):
        print("Loading : ", load_path)
        ckpt = torch.load(load_path, map_location=self.device)
        self.step = ckpt['step']
        self.model.load_state_dict(_match_module_prefix(ckpt['model'], self.model.state_dict().keys()))
        self.ema_model.load_state_dict(_match_module_prefix(ckpt['ema'], self.ema_model.state_dict().keys()))
        rt.bump_weights_epoch()

    # -- the hot loop (DEBLUR:1183-1235) -----------------------------------------------------------------
    def _next_batch(self):
        d = next(self.dl)
        if isinstance(d, (list, tuple)):
            d = d[0]
        return d.to(self.device, non_blocking=True)

    def _second(self, batch):
        """The second argument of forward(x1, x2), None for the one-image packages; fresh Gaussian noise for the
        denoising package (DENOISE:738-742)."""
        return torch.randn_like(batch) if self.pair_noise else None

    def _sample_source(self):
        """The images a milestone samples from: a data batch, or for the two-image packages the second image
        (fresh noise drawn like a batch: DENOISE:759-762)."""
        og_img = self._next_batch()
        x2 = self._second(og_img)
        return og_img if x2 is None else x2

    def _loss(self, batch):
        x2 = self._second(batch)
        return self.core(batch) if x2 is None else self.core(batch, x2)

    def train_step(self):
        """One optimizer step = gradient_accumulate_every micro-steps + Adam (+ EMA); returns the
        mean micro-step loss as a 0-dim device tensor (no host sync)."""
        acc = self.gradient_accumulate_every
        scale = 1.0 / (acc * parallel.world_size())
        total = None
        for i in range(acc):
            if self.sync is not None:
                self.sync.begin()
            loss = torch.mean(self._loss(self._next_batch()))
            if self.sync is not None and i == acc - 1:
                self.sync.arm()
            (loss * scale).backward()
            total = loss.detach() if total is None else total + loss.detach()
        if self.sync is not None:
            self.sync.finish()
        self.opt.step()
        self.opt.zero_grad()
        if self.step % self.update_ema_every == 0:
            self.step_ema()
        return total / acc

    def train(self):
        acc_loss = 0
        while self.step < self.train_num_steps:
            u_loss = self.train_step()
            if self.step % 100 == 0 and parallel.rank() == 0 and not self.quiet:
                print(f'{self.step}: {u_loss.item()}')
            acc_loss = acc_loss + u_loss
            if self.step != 0 and self.step % self.save_and_sample_every == 0:
                self._milestone(acc_loss)
                acc_loss = 0
            self.step += 1
        print('training completed')

    def _milestone(self, acc_loss):
        milestone = self.step // self.save_and_sample_every
        if parallel.rank() == 0:                       # sampling and checkpointing stay on one GPU
            og_img = self._sample_source()
            if hasattr(self.ema_core, 'defade_fn'):
                xt, direct_recons, all_images = self.ema_core.sample(batch_size=self.batch_size, faded_recon_sample=og_img)
            else:
                xt, direct_recons, all_images = self.ema_core.sample(batch_size=self.batch_size, img=og_img)
            for name, im in (('og', og_img), ('recon', all_images), ('direct_recons', direct_recons), ('xt', xt)):
                save_image((im + 1) * 0.5, self.results_folder / f'sample-{name}-{milestone}.png', nrow=6)
            mean = float(acc_loss) / (self.save_and_sample_every + 1)
            print(f'Mean of last {self.step}: {mean}')
            self.save()
            if self.step % (self.save_and_sample_every * 100) == 0:
                self.save(self.step)
        if parallel.world_size() > 1:
            torch.distributed.barrier()


class DemixTrainer(Trainer):
    """Trainer of demixing_diffusion_pytorch (DEMIX:596-775): two image folders; every micro-step mixes a batch of the
    first into a batch of the second, and sampling starts from images of the second."""

    def __init__(self, diffusion_model, folder1, folder2, *, dataset=None, shuffle=True, num_workers=8, **kw):
        super().__init__(diffusion_model, folder1, dataset=dataset, shuffle=shuffle, num_workers=num_workers, **kw)
        self.pair_noise = False
        self.ds1, self.dl1 = self.ds, self.dl
        self.ds2, self.dl2 = self._make_loader(folder2, dataset, shuffle, num_workers, seed=7654321)

    def _second(self, batch=None):
        d = next(self.dl2)
        if isinstance(d, (list, tuple)):
            d = d[0]
        return d.to(self.device, non_blocking=True)

    def _sample_source(self):
        return self._second()          # DEMIX:744 draws from the second loader only


class DefadeGenTrainer(Trainer):
    """Trainer of the defading-generation package (DEFGEN:646-810): the second image is one uniform random colour per
    sample and channel, rand(B, 3) - 0.5 spread over the image (DEFGEN:769-773)."""

    def __init__(self, diffusion_model, folder, **kw):
        super().__init__(diffusion_model, folder, **kw)
        self.pair_noise = False

    def _second(self, batch):
        B, C, H, W = batch.shape
        c = torch.rand((B, C), device=batch.device) - 0.5
        return c[:, :, None, None].expand(B, C, H, W).contiguous()

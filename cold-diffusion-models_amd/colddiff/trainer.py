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
from .evaluate import EvalMixin, GenEvalMixin
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


# -- image folder datasets with PIL only (torchvision is not a dependency) -------------------------------------------
class Recipe:
    """One of the reference's `transforms.Compose` chains, as data.  Every chain is
    [deterministic resize] -> crop (random / centre, optionally after a zero border) -> [mirror] -> ToTensor -> t * 2 - 1.

    resize: 'sq112'  transforms.Resize((int(1.12 s), int(1.12 s)))                    (DEBLUR:990, 1011)
            'short'  transforms.Resize(s): shorter edge to s, aspect kept             (RESOL:824, DEFADE:564)
            'none'   the file's own size                                              (DEFADE:588: RandomCrop first; the Resize(s)
                                                                                       after an s x s crop returns its input)
    pad:    RandomCrop(s, padding=pad), constant fill 0                               (RESOL:825, DEFADE:588)
    crop:   'random' (RandomCrop) | 'center' (CenterCrop)
    rgb:    img.convert('RGB') before the chain                                       (DENOISE:565, 588)"""

    def __init__(self, name, resize, crop, flip, pad=0, rgb=False):
        self.name, self.resize, self.crop, self.flip, self.pad, self.rgb = name, resize, crop, flip, pad, rgb

    def with_rgb(self):
        return Recipe(self.name, self.resize, self.crop, self.flip, self.pad, True)

    def __repr__(self):
        return f"Recipe({self.name}: resize={self.resize} pad={self.pad} crop={self.crop} flip={self.flip} rgb={self.rgb})"


AUG1 = Recipe('Dataset_Aug1', 'sq112', 'random', True)                  # DEBLUR:983-1004
CENTER112 = Recipe('Dataset', 'sq112', 'center', False)                 # DEBLUR:1006-1026
AUG2 = Recipe('Dataset_Aug2', 'short', 'random', True, pad=4)           # RESOL:817-831
CIFAR_PAD = Recipe('DatasetCifar10', 'none', 'random', True, pad=4)     # DEFADE:579-599
CENTER_SHORT = Recipe('Dataset', 'short', 'center', False)              # DEFADE:557-576


def center_offset(full, size):
    """torchvision's CenterCrop: int(round((full - size) / 2.0)) with Python's round-half-to-even -- NOT (full - size) // 2:
    143 -> 128 starts at 8, 71 -> 64 at 4, 35 -> 32 and 31 -> 28 at 2."""
    return int(round((full - size) / 2.0))


def resized(img, size, recipe):
    """The deterministic head of the chain on a PIL image (convert, Resize)."""
    from PIL import Image
    if recipe.rgb:
        img = img.convert('RGB')
    if recipe.resize == 'sq112':
        big = int(size * 1.12)
        return img.resize((big, big), Image.BILINEAR)
    if recipe.resize == 'short':
        w, h = img.size
        short, long = (w, h) if w <= h else (h, w)
        if short == size:
            return img                                   # torchvision returns the image itself
        new_long = int(size * long / short)
        return img.resize((size, new_long) if w <= h else (new_long, size), Image.BILINEAR)
    return img


def _load_image(path, size, recipe):
    from PIL import Image
    import numpy as np
    img = resized(Image.open(path), size, recipe)
    arr = np.asarray(img)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    if recipe.pad:
        arr = np.pad(arr, ((recipe.pad, recipe.pad), (recipe.pad, recipe.pad), (0, 0)))
    H, W = arr.shape[:2]
    assert H >= size and W >= size, f"{path}: {W}x{H} after the resize is smaller than the {size}x{size} crop"
    if recipe.crop == 'random':
        if H == size and W == size:
            oy = ox = 0                                  # RandomCrop.get_params draws nothing in this case
        else:
            oy = int(torch.randint(0, H - size + 1, (1,)))          # i (top) first, then j (left)
            ox = int(torch.randint(0, W - size + 1, (1,)))
    else:
        oy, ox = center_offset(H, size), center_offset(W, size)
    arr = arr[oy:oy + size, ox:ox + size]
    if recipe.flip and torch.rand(1).item() < 0.5:
        arr = arr[:, ::-1]
    t = torch.from_numpy(np.array(arr)).permute(2, 0, 1).float().div(255)   # ToTensor
    return t * 2 - 1


class Dataset(data.Dataset):
    recipe = CENTER112

    def __init__(self, folder, image_size, exts=('jpg', 'jpeg', 'png'), recipe=None):
        super().__init__()
        self.folder, self.image_size = folder, image_size
        self.paths = [p for ext in exts for p in Path(f'{folder}').glob(f'**/*.{ext}')]
        if recipe is not None:
            self.recipe = recipe

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, index):
        return _load_image(self.paths[index], self.image_size, self.recipe)


class Dataset_Aug1(Dataset):
    recipe = AUG1


class Dataset_Aug2(Dataset):
    recipe = AUG2


class DatasetCifar10(Dataset):
    recipe = CIFAR_PAD


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


class Trainer(EvalMixin):
    # which transform chain a `dataset=` name selects (DEBLUR:1094-1114); the other packages override this table
    drop_last = True
    force_shuffle = False
    image_size_from_model = False

    @staticmethod
    def recipe_for(dataset):
        if dataset in ('mnist', 'cifar10', 'flower', 'celebA', 'AFHQ'):
            return AUG1
        if dataset == 'LSUN_train':                      # (upstream reads the LSUN lmdb through torchvision.datasets; a folder of its
            return Recipe('LSUN_train', 'sq112', 'random', False)     # images gets the same chain: no mirror, DEBLUR:1098-1108)
        return CENTER112

    def __init__(self, diffusion_model, folder, *, ema_decay=0.995, image_size=128, train_batch_size=32, train_lr=2e-5,
                 train_num_steps=100000, gradient_accumulate_every=2, fp16=False, step_start_ema=2000, update_ema_every=10,
                 save_and_sample_every=1000, results_folder='./results', load_path=None, dataset=None, shuffle=True,
                 num_workers=8, device_data=None):
        super().__init__()
        assert not fp16, "apex fp16 is not supported (every reference script passes fp16=False)"
        self.model = diffusion_model
        self.ema = EMA(ema_decay)
        self.ema_model = copy.deepcopy(self.model)
        self.update_ema_every = update_ema_every
        self.step_start_ema = step_start_ema
        self.save_and_sample_every = save_and_sample_every
        self.batch_size = train_batch_size
        self.gradient_accumulate_every = gradient_accumulate_every
        self.train_num_steps = train_num_steps
        self.core = unwrap(self.model)
        # RESOL:874 / DEFADE:681 take the sampling size from the model; the dataset still crops to the `image_size` argument
        self.image_size = self.core.image_size if self.image_size_from_model else image_size
        self.data_image_size = image_size
        self.ema_core = unwrap(self.ema_model)
        # the denoising package feeds (image, fresh Gaussian noise) pairs (DENOISE:738-742)
        self.pair_noise = hasattr(self.core, 'sqrt_alphas_cumprod')
        self.device = next(self.core.parameters()).device
        # device_data: keep the (1.12x-resized, uint8) image folder in HBM and crop / mirror / convert per batch in one kernel instead
        # of PIL DataLoader workers.  None = automatic: on for a HIP device unless COLDDIFF_DEVICE_DATA=0.
        if device_data is None:
            device_data = self.device.type == 'cuda' and os.environ.get("COLDDIFF_DEVICE_DATA", "1") != "0"
        self.device_data = bool(device_data)

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
            # ranks may arrive here minutes apart (each decoded its image folder into its device cache above): wait on the long-timeout
            # host group first, so that the broadcast's own watchdog only ever covers the broadcast
            parallel.milestone_barrier()
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
            return None, SyntheticImages(self.batch_size, self.core.channels, self.data_image_size, self.device, seed=seed + parallel.rank())
        recipe = self.recipe_for(dataset)
        self.recipe = recipe
        print(dataset, "DA used" if recipe.crop == 'random' else "")
        shuffle = True if self.force_shuffle else shuffle
        drop_last = self.drop_last or parallel.world_size() > 1      # (ranks must run the same number of equal micro-batches)
        if self.device_data:
            try:
                cache = DeviceImageCache(folder, self.data_image_size, self.device, recipe=recipe)
                return cache, DeviceLoader(cache, self.batch_size, shuffle=shuffle, seed=seed, rank=parallel.rank(),
                                           world=parallel.world_size(), drop_last=drop_last)
            except CacheUnfit as e:
                print(f"device image cache not used ({e}); falling back to the host DataLoader")
                self.device_data = False
        ds = Dataset(folder, self.data_image_size, recipe=recipe)
        sampler = None
        if parallel.world_size() > 1:
            sampler = data.distributed.DistributedSampler(ds, shuffle=shuffle)
        dl = cycle(data.DataLoader(ds, batch_size=self.batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                                   pin_memory=self.device.type == 'cuda', num_workers=num_workers, drop_last=drop_last), sampler)
        return ds, dl

    # -- EMA / checkpoint --------------------------------------------------------------------------------
    def reset_parameters(self):
        """ema_model.load_state_dict(model.state_dict()) (DEBLUR:1131-1132; every `update_ema_every`-th step until step_start_ema): the
        parameters live in two flat arenas of one layout, so they are ONE device copy instead of one per tensor (238 for the CelebA net);
        buffers go through load_state_dict's own path."""
        a, e = getattr(self, "arena", None), getattr(self, "ema_arena", None)
        if a is not None and e is not None and a.numel == e.numel and a.offsets == e.offsets and a.intact() and e.intact():
            e.data.copy_(a.data)
            bufs = dict(self.model.named_buffers())
            with torch.no_grad():
                for name, b in self.ema_model.named_buffers():
                    if name in bufs and b.data_ptr() != bufs[name].data_ptr():
                        b.copy_(bufs[name])
        else:
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

    # -- degradation prefetch (deblurring: q_sample is up to T sequential blur steps on B*C planes, independent of the network) ----
    # The trajectory is bit-identical to the non-prefetching loop BETWEEN milestones; a milestone's sample batch is drawn after the
    # micro-batches in the queue were already prefetched, i.e. one batch later in the data order than without prefetch for the plain
    # loop and `gradient_accumulate_every` batches later with fused accumulation.  ONE queue serves both loops (draw order = launch
    # order), so switching COLDDIFF_FUSE_ACCUM / falling back after an out-of-memory mid-run skips nothing and leaks nothing.
    _side = None
    _pending_q = None            # prefetched micro-batches [(prep, event)], oldest first
    _fuse_off = False            # set by the out-of-memory fallback of _fused_step

    def _can_prefetch(self):
        """Only for the plain loop on a HIP device: not when a test / subclass replaced _loss, not for two-image packages."""
        return (self.device.type == 'cuda' and hasattr(self.core, 'can_prepare_async') and '_loss' not in self.__dict__
                and type(self)._loss is Trainer._loss and not self.pair_noise and rt._lib_override is None
                and getattr(self.core, 'train_routine', None) == 'Final' and os.environ.get("COLDDIFF_PREFETCH", "1") != "0"
                and self.core.can_prepare_async())

    # -- fused gradient accumulation -------------------------------------------------------------------------------------------
    # The reference runs `gradient_accumulate_every` micro-batches one after the other, (loss_i / accumulate).backward() each
    # (DEBLUR:1188-1195) -- a memory measure for 16 GB cards.  With 288 GB of HBM the micro-batches of one optimizer step are
    # degraded one by one (same data order, same t / noise / offset draws in the same order) and then go through the network as ONE
    # batch: the loss of the concatenated batch is the mean of the micro-batch losses, so one backward pass yields the same
    # gradient sum, every GEMM sees twice the rows (the 16 x 16 layers fill the chip, the weight gradients need half the split-K
    # slabs), weights are read once per step instead of per micro-batch and the launch count halves.
    # CelebA-128 step (2 x 32 images): 58.4 -> 53.2 ms, 1096 -> 1202 img/s in one call (profiles/round4_fused_accumulation_ab.txt).
    FUSE_MAX_PIXELS = int(os.environ.get("COLDDIFF_FUSE_MAX_PIXELS", str(1 << 21)))     # 128 images of 128 x 128 (~90 GB of activations)

    def _can_fuse(self):
        """The plain loop only (not when a test / subclass replaced _loss), equal-sized micro-batches (the loaders that may end an
        epoch on a short batch do not fuse), a core whose training routine has the two-phase form."""
        acc = self.gradient_accumulate_every
        return (acc > 1 and not self._fuse_off and '_loss' not in self.__dict__ and type(self)._loss is Trainer._loss and os.environ.get("COLDDIFF_FUSE_ACCUM", "1") != "0"
                and hasattr(self.core, 'prepare') and self.core.fusable() and (self.drop_last or self.ds is None or parallel.world_size() > 1)
                and acc * self.batch_size * self.data_image_size ** 2 <= self.FUSE_MAX_PIXELS)

    def _prepare_micro(self):
        """One micro-batch in the reference's order of draws: the data batch, the second image / fresh noise (DENOISE:738-742), then
        forward()'s t (and whatever the degradation draws)."""
        batch = self._next_batch()
        x2 = self._second(batch)
        return self.core.prepare(batch) if x2 is None else self.core.prepare(batch, x2)

    def _take_prepared(self, keep):
        """The oldest prefetched micro-batch (launched now if the queue is empty), ordered before the compute stream's next kernel;
        afterwards the queue is topped up to `keep` entries."""
        if self._pending_q is None:
            self._pending_q = []
        if not self._pending_q:
            self._pending_q.append(self._launch_prepare())
        prep, ev = self._pending_q.pop(0)
        torch.cuda.current_stream().wait_event(ev)
        while len(self._pending_q) < keep:
            self._pending_q.append(self._launch_prepare())
        return prep

    def _fused_step(self, acc, prefetch):
        if self.sync is not None:
            self.sync.begin()
        if prefetch:
            # all of the NEXT step's micro-batches are degraded on the side stream under this step's forward / backward
            if self._pending_q is None:
                self._pending_q = []
            while len(self._pending_q) < acc:
                self._pending_q.append(self._launch_prepare())
            preps = [self._take_prepared(0) for _ in range(acc)]
            for _ in range(acc):
                self._pending_q.append(self._launch_prepare())
        else:
            preps = [self._prepare_micro() for _ in range(acc)]
        oom = False
        try:
            prep = tuple(torch.cat(parts) for parts in zip(*preps))
            loss = torch.mean(self.core.loss_prepared(prep))          # = the mean of the micro-batch losses (equal sizes)
            if self.sync is not None:
                self.sync.arm()
            (loss * (1.0 / parallel.world_size())).backward()
            return loss.detach()
        except torch.OutOfMemoryError:
            # FUSE_MAX_PIXELS counts pixels, not model width: a wide model / a shared device may not hold `acc` micro-batches of
            # activations where the reference's one-by-one loop fits.  Single rank only (on several ranks the others are already inside
            # their collectives).  Only a flag is set here: while this handler runs, the exception's traceback keeps every frame of the
            # aborted pass alive (Unet.forward's skip list, the nodes' locals, the partial graph with its saved activations), so the retry
            # happens AFTER the handler, when all of that has been released.
            if self.sync is not None:
                raise
            oom = True
        if oom:
            # drop what the aborted pass left, switch fusion off for this Trainer and run the SAME prepared micro-batches one after the
            # other -- the step's data, t and noise draws are unchanged
            import gc
            prep = loss = None
            self._fuse_off = True
            gc.collect()
            torch.cuda.empty_cache()
            self.opt.zero_grad()
            print(f"fused accumulation of {acc} micro-batches does not fit in device memory: running them one by one from here on")
            total = None
            for p_i in preps:
                l_i = torch.mean(self.core.loss_prepared(p_i))
                (l_i * (1.0 / acc)).backward()
                total = l_i.detach() if total is None else total + l_i.detach()
            return total / acc

    def _launch_prepare(self):
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        # every launch: whatever the main stream produced that the side stream reads (the epoch permutation a milestone's
        # _next_batch rebuilt, kernel-stack tensors) is ordered before it.  Issued BEFORE forward i is enqueued, so the prefetch
        # still overlaps with the forward / backward of micro-batch i.
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            prep = self.core.prepare(self._next_batch())
            ev = torch.cuda.Event()
            ev.record(self._side)
        for t in prep:
            t.record_stream(main)                            # produced on the side stream, consumed (and freed) on the main stream
        return prep, ev

    def _loss(self, batch):
        x2 = self._second(batch)
        return self.core(batch) if x2 is None else self.core(batch, x2)

    def train_step(self):
        """One optimizer step = gradient_accumulate_every micro-batches (one fused pass when `_can_fuse()`, else one forward /
        backward each: DEBLUR:1188-1195) + Adam (+ EMA); returns the mean micro-batch loss as a 0-dim device tensor (no host sync)."""
        acc = self.gradient_accumulate_every
        scale = 1.0 / (acc * parallel.world_size())
        total = None
        prefetch = self._can_prefetch()
        fused = self._can_fuse()
        if fused:
            total = self._fused_step(acc, prefetch) * acc
        for i in range(0 if fused else acc):
            if self.sync is not None:
                self.sync.begin()
            if prefetch:
                prep = self._take_prepared(1)                # micro-batch i+1 is degraded under the forward / backward of micro-batch i
                loss = torch.mean(self.core.loss_prepared(prep))
            else:
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
        parallel.milestone_barrier()


class DemixTrainer(GenEvalMixin, Trainer):
    """Trainer of demixing_diffusion_pytorch (DEMIX:596-775): two image folders; every micro-step mixes a batch of the
    first into a batch of the second, and sampling starts from images of the second."""

    @property
    def _fid_batch(self):                      # DEMIX:818: bs = self.batch_size, seeds = the next batch of the second folder
        return self.batch_size

    def _seed_images(self, bs):
        return self._second()[:bs]

    def __init__(self, diffusion_model, folder1, folder2, *, dataset=None, shuffle=True, num_workers=8, **kw):
        super().__init__(diffusion_model, folder1, dataset=dataset, shuffle=shuffle, num_workers=num_workers, **kw)
        self.pair_noise = False
        self.ds1, self.dl1 = self.ds, self.dl
        self.ds2, self.dl2 = self._make_loader(folder2, dataset, shuffle, num_workers, seed=7654321)

    @staticmethod
    def recipe_for(dataset):                   # DEMIX:636-643: 'train' augments; both datasets convert to RGB (DEMIX:545, 568)
        return (AUG1 if dataset == 'train' else CENTER112).with_rgb()

    def _second(self, batch=None):
        d = next(self.dl2)
        if isinstance(d, (list, tuple)):
            d = d[0]
        return d.to(self.device, non_blocking=True)

    def _sample_source(self):
        return self._second()          # DEMIX:744 draws from the second loader only


class DefadeGenTrainer(GenEvalMixin, Trainer):
    """Trainer of the defading-generation package (DEFGEN:646-810): the second image is one uniform random colour per
    sample and channel, rand(B, 3) - 0.5 spread over the image (DEFGEN:769-773)."""

    @property
    def _fid_batch(self):                      # DEFGEN:880-887: bs = self.batch_size, seeds = one random colour per image
        return self.batch_size

    def _seed_images(self, bs):
        n = self.image_size
        return (torch.rand((bs, 3)) - 0.5)[:, :, None, None].expand(bs, 3, n, n).to(self.device).contiguous()

    def __init__(self, diffusion_model, folder, **kw):
        super().__init__(diffusion_model, folder, **kw)
        self.pair_noise = False

    @staticmethod
    def recipe_for(dataset):                   # DEFGEN:682-687: 'train' augments; both datasets convert to RGB (DEFGEN:591, 614)
        return (AUG1 if dataset == 'train' else CENTER112).with_rgb()

    def _second(self, batch):
        B, C, H, W = batch.shape
        c = torch.rand((B, C), device=batch.device) - 0.5
        return c[:, :, None, None].expand(B, C, H, W).contiguous()


# -- device-side input pipeline -----------------------------------------------------------------------
class CacheUnfit(Exception):
    """The folder cannot live in the device cache (ragged image sizes after the recipe's resize, or larger than the memory
    budget): the Trainer falls back to the host Dataset + DataLoader."""


class DeviceImageCache:
    """An image folder decoded ONCE and kept in HBM as uint8 NHWC, already past the DETERMINISTIC head of the reference's
    transform chain (`convert('RGB')`, `transforms.Resize(...)` -- done with the same PIL call on the host at cache-build time by a
    thread pool).  What is random per sample (crop offset, mirror) or pure arithmetic (border, ToTensor, t*2-1) runs per batch in
    ONE kernel (cdf_augment_batch_pad): the training loop never waits for host image workers (the reference spawns 8-16 PIL
    processes, DEBLUR:1095, 1107, 1114).  CelebA (202 599 images) at S = 143: 12.4 GB of the MI355X's 288 GB.
    Raises CacheUnfit when the images are not all of one size after the resize (`Resize(s)` keeps the aspect ratio) or when the
    cache would not fit next to the model (budget: COLDDIFF_CACHE_FRACTION of the free HBM, default 0.5; pinned host staging buffer
    of the same size, at most half of the free host RAM)."""

    def __init__(self, folder, image_size, device, exts=('jpg', 'jpeg', 'png'), decode_threads=None, paths=None, recipe=None):
        import numpy as np
        from concurrent.futures import ThreadPoolExecutor
        from PIL import Image
        self.recipe = recipe = recipe or AUG1
        self.image_size = image_size
        self.paths = list(paths) if paths is not None else [p for ext in exts for p in Path(f'{folder}').glob(f'**/*.{ext}')]
        assert len(self.paths) > 0, f"no images under {folder}"

        def load(p):
            arr = np.asarray(resized(Image.open(p), image_size, recipe))
            return arr[:, :, None] if arr.ndim == 2 else arr

        first = load(self.paths[0])
        self.SH, self.SW, self.channels = first.shape
        self.S = self.SH                                            # (square caches: the 1.12x chains)
        n = len(self.paths)
        dev = torch.device(device)
        need = n * first.size
        if dev.type == 'cuda':
            free_hbm = torch.cuda.mem_get_info(dev)[0]
            frac = float(os.environ.get("COLDDIFF_CACHE_FRACTION", "0.5"))
            if need > frac * free_hbm:
                raise CacheUnfit(f"{n} images of {self.SH}x{self.SW}x{self.channels} = {need / 2**30:.1f} GiB exceed "
                                 f"{frac:.2f} of the free HBM ({free_hbm / 2**30:.1f} GiB)")
            try:
                import psutil
                if need > 0.5 * psutil.virtual_memory().available:
                    raise CacheUnfit(f"{need / 2**30:.1f} GiB of pinned staging exceed half of the free host memory")
            except ImportError:
                pass
        pad = recipe.pad
        if self.SH + 2 * pad < image_size or self.SW + 2 * pad < image_size:
            raise CacheUnfit(f"{self.paths[0]}: {self.SW}x{self.SH} is smaller than the {image_size}x{image_size} crop")
        host = torch.empty((n, self.SH, self.SW, self.channels), dtype=torch.uint8, pin_memory=dev.type == 'cuda')
        hv = host.numpy()
        hv[0] = first

        def fill(i):
            a = load(self.paths[i])
            if a.shape != first.shape:
                raise CacheUnfit(f"{self.paths[i]}: {a.shape} vs {first.shape} (ragged sizes / mixed image modes in one folder)")
            hv[i] = a

        with ThreadPoolExecutor(max_workers=decode_threads or min(32, os.cpu_count() or 1)) as ex:   # PIL releases the GIL while decoding / resizing
            list(ex.map(fill, range(1, n)))
        self.data = host.to(device, non_blocking=True)

    def __len__(self):
        return len(self.paths)

    def span(self):
        """(rows, columns) of valid crop offsets: the padded image minus the crop, plus one."""
        p = self.recipe.pad
        return self.SH + 2 * p - self.image_size + 1, self.SW + 2 * p - self.image_size + 1

    def center(self):
        p = self.recipe.pad
        return center_offset(self.SH + 2 * p, self.image_size), center_offset(self.SW + 2 * p, self.image_size)

    def batch(self, idx, oy, ox, flip):
        """[B, C, H, H] fp32 batch: images idx (int64 [B]) cropped at (oy, ox) of the bordered image, mirrored where flip != 0
        (int32 [B] each, on the device)."""
        rt.check(self.data)
        B, H = idx.numel(), self.image_size
        out = torch.empty((B, self.channels, H, H), device=self.data.device, dtype=torch.float32)
        rt.lib().cdf_augment_batch_pad(rt.P(self.data), len(self.paths), self.SH, self.SW, self.channels, self.recipe.pad, rt.P(idx), rt.P(oy),
                                       rt.P(ox), rt.P(flip), rt.P(out), B, H, H, rt.stream(self.data))
        return out

    def item(self, idx):
        """Image idx as the non-random chain yields it ([C, H, H]; centre crop, no mirror)."""
        dev = self.data.device
        i32 = lambda v: torch.tensor([v], dtype=torch.int32, device=dev)
        cy, cx = self.center()
        return self.batch(torch.tensor([idx], device=dev), i32(cy), i32(cx), i32(0))[0]


class DeviceLoader:
    """Endless batch iterator over a DeviceImageCache with the sampling semantics of the reference's loader: `shuffle=True`
    permutes the indices every epoch, `drop_last` as the package's DataLoader has it (DEBLUR:1095-1096 drops, RESOL:887 keeps the
    short batch), batch_size images per step; with several ranks every rank takes the DistributedSampler slice perm[rank::world] of
    the SAME epoch permutation (seed + epoch on a CPU generator).  Crop and mirror follow the cache's recipe: RandomCrop offsets
    (row first, then column; none drawn when the image IS the crop size, like RandomCrop.get_params) and RandomHorizontalFlip(0.5),
    or CenterCrop."""

    def __init__(self, cache, batch_size, shuffle=True, seed=123457, rank=0, world=1, drop_last=True, augment=None):
        self.cache, self.batch_size, self.shuffle, self.drop_last = cache, batch_size, shuffle, drop_last
        rec = cache.recipe
        self.random_crop = rec.crop == 'random' if augment is None else bool(augment)
        self.flip = rec.flip if augment is None else bool(augment)
        self.seed, self.rank, self.world = seed, rank, world
        self.epoch, self.pos, self.order = 0, 0, None
        dev = cache.data.device
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed + 7919 * rank)
        n = len(cache) // world if world > 1 else len(cache)
        assert n >= batch_size or not drop_last, f"{len(cache)} images over {world} ranks: fewer than one batch of {batch_size}"
        cy, cx = cache.center()
        self._cy = torch.full((batch_size,), cy, dtype=torch.int32, device=dev)
        self._cx = torch.full((batch_size,), cx, dtype=torch.int32, device=dev)
        self._noflip = torch.zeros((batch_size,), dtype=torch.int32, device=dev)

    def _new_epoch(self):
        n = len(self.cache)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            perm = torch.randperm(n, generator=g)
        else:
            perm = torch.arange(n)
        if self.world > 1:
            perm = perm[:n - n % self.world][self.rank::self.world]
        self.order = perm.to(self.cache.data.device)
        self.pos = 0
        self.epoch += 1

    def __iter__(self):
        return self

    def __next__(self):
        B = self.batch_size
        left = 0 if self.order is None else self.order.numel() - self.pos
        if left <= 0 or (left < B and self.drop_last):
            self._new_epoch()
            left = self.order.numel()
        B = min(B, left)                                                   # (short last batch when drop_last is off)
        idx = self.order[self.pos:self.pos + B].contiguous()
        self.pos += B
        dev = self.cache.data.device
        sy, sx = self.cache.span()
        if self.random_crop and (sy > 1 or sx > 1):
            oy = torch.randint(0, sy, (B,), generator=self.gen, device=dev, dtype=torch.int32)     # RandomCrop: i (top), then j (left)
            ox = torch.randint(0, sx, (B,), generator=self.gen, device=dev, dtype=torch.int32)
        elif self.random_crop:
            oy = ox = self._noflip[:B]
        else:
            oy, ox = self._cy[:B], self._cx[:B]
        if self.flip:
            flip = (torch.rand((B,), generator=self.gen, device=dev) < 0.5).to(torch.int32)      # RandomHorizontalFlip(p = 0.5)
        else:
            flip = self._noflip[:B]
        return self.cache.batch(idx, oy, ox, flip)


# -- the other packages' Trainers: same loop, their own `dataset=` tables and loader options ---------------------------------
class DenoiseTrainer(GenEvalMixin, Trainer):
    """denoising_diffusion_pytorch.py:620-789: `dataset == 'train'` augments, everything else is the centre crop; both datasets
    convert to RGB first (DENOISE:565, 588)."""
    _fid_batch = 128                           # DENOISE:835-839: bs = 128, seeds = N(0, 1) images (128 x 128 upstream: the model's size here)

    def _seed_images(self, bs):
        n = self.image_size
        return torch.randn(bs, 3, n, n).to(self.device)

    def _gen(self, bs, og_img, noise):         # (DENOISE:843: gen_sample has no noise_level in this package)
        return self.ema_core.gen_sample(batch_size=bs, img=og_img)

    @staticmethod
    def recipe_for(dataset):
        return (AUG1 if dataset == 'train' else CENTER112).with_rgb()


class ResolutionTrainer(Trainer):
    """resolution_diffusion_pytorch.py:837-900: 'cifar10' / 'celebA' -> Dataset_Aug1, 'flower' -> Dataset_Aug2 (Resize(s),
    RandomCrop(s, padding=4), mirror), else the centre crop; the DataLoader keeps the short last batch (no drop_last, RESOL:887);
    image_size comes from the model (RESOL:874)."""
    drop_last = False
    image_size_from_model = True

    @staticmethod
    def recipe_for(dataset):
        if dataset in ('cifar10', 'celebA'):
            return AUG1
        if dataset == 'flower':
            return AUG2
        return CENTER112

    def sample_as_a_mean_blur_torch_gmm_ablation(self, torch_gmm, siz=2, ch=3, clusters=10, sample_at=6, noise=0, num_samples=6400, bs=64):
        """RESOL:1117-1182 (this package's form of the method: the GMM is fitted on the `siz` x `siz` area-resampled degradations
        `opt(img, t=sample_at)` of the dataset, its samples are blown up nearest-exact and handed to `gen_sample`)."""
        import torch.nn.functional as F
        from .evaluate import _create_folder
        feats = [F.interpolate(self.ema_core.opt(img, t=sample_at), size=siz, mode='area').flatten(1) for img in self._dataset_batches(100)]
        model = self._fit_gmm(torch_gmm, torch.cat(feats, dim=0), clusters, 100)
        n = self.image_size
        og_x = F.interpolate(model.sample(num_datapoints=num_samples).to(self.device).reshape(num_samples, 3, siz, siz).float(), size=n,
                             mode='nearest-exact')
        xt_folder, out_folder, dr_folder = f'{self.results_folder}_xt', f'{self.results_folder}_out', f'{self.results_folder}_dir_recons'
        for f in (xt_folder, out_folder, dr_folder):
            _create_folder(f)
        cnt = 0
        for j in range(num_samples // bs):
            og_img = og_x[j * bs: j * bs + bs].expand(bs, ch, n, n).float().contiguous()
            xt, direct_recons, all_images = self.ema_core.gen_sample(batch_size=bs, img=og_img, noise_level=noise)
            for i in range(all_images.shape[0]):
                self._save(all_images[i:i + 1], f'{out_folder}/sample-x0-{cnt}.png', nrow=1)
                self._save(xt[i:i + 1], f'{xt_folder}/sample-x0-{cnt}.png', nrow=1)
                self._save(direct_recons[i:i + 1], f'{dr_folder}/sample-x0-{cnt}.png', nrow=1)
                cnt += 1
        return cnt


class DefadeTrainer(Trainer):
    """defading_diffusion_gaussian.py:651-705: 'cifar10' -> DatasetCifar10 (RandomCrop(s, padding=4) on the file's own size, mirror),
    'celebA' -> DatasetCelebA (= Dataset_Aug1), 'celebA_test' -> DatasetCelebATest (1.12x, centre), else Resize(s) + CenterCrop(s);
    always shuffled (DEFADE:695), no `shuffle=` argument upstream (accepted and ignored here)."""
    force_shuffle = True
    image_size_from_model = True

    def __init__(self, diffusion_model, folder, *, train_num_steps=700000, save_and_sample_every=10000, **kw):
        # this package's own defaults (DEFADE:661, 666); its scripts (cifar10_train.py, celebA_train.py) do not pass
        # save_and_sample_every, so a ported script must checkpoint / run the T-step sampler every 10000 steps, not every 1000
        super().__init__(diffusion_model, folder, train_num_steps=train_num_steps, save_and_sample_every=save_and_sample_every, **kw)

    @staticmethod
    def recipe_for(dataset):
        if dataset == 'cifar10':
            return CIFAR_PAD
        if dataset == 'celebA':
            return Recipe('DatasetCelebA', 'sq112', 'random', True)
        if dataset == 'celebA_test':
            return Recipe('DatasetCelebATest', 'sq112', 'center', False)
        return CENTER_SHORT

    # ---- this package's test methods (DEFADE:814-940, 1146-1244): `all_sample` / `sample` take `faded_recon_sample=` here -------------------
    def _frames(self, og_img, extra_path, x0_list, xt_list):
        """og-<extra>.png, sample-<i>-<extra>-x0 / -xt.png and the two GIFs (titles are figure code)"""
        from PIL import Image
        self._save(og_img, str(self.results_folder / f'og-{extra_path}.png'))
        frames_0, frames_t = [], []
        for i in range(len(x0_list)):
            p0, pt = str(self.results_folder / f'sample-{i}-{extra_path}-x0.png'), str(self.results_folder / f'sample-{i}-{extra_path}-xt.png')
            self._save(x0_list[i], p0)
            frames_0.append(p0)
            if i < len(xt_list):
                self._save(xt_list[i], pt)
                frames_t.append(pt)
        for name, frames in ((f'Gif-{extra_path}-x0.gif', frames_0), (f'Gif-{extra_path}-xt.gif', frames_t)):
            ims = [Image.open(f).convert('RGB') for f in frames if os.path.exists(f)]
            if ims:
                ims[0].save(str(self.results_folder / name), save_all=True, append_images=ims[1:], duration=100, loop=0)
        return x0_list, xt_list

    def test_from_data(self, extra_path, s_times=None):                # DEFADE:814-841
        og_img = self._next_batch()
        x0_list, xt_list = self.ema_core.all_sample(batch_size=self.batch_size, faded_recon_sample=og_img, times=s_times)
        return self._frames(og_img, extra_path, x0_list, xt_list)

    def test_with_mixup(self, extra_path):                             # DEFADE:843-883: the mean of two batches as the start
        og_img_1, og_img_2 = self._next_batch(), self._next_batch()
        og_img = (og_img_1 + og_img_2) / 2
        x0_list, xt_list = self.ema_core.all_sample(batch_size=self.batch_size, faded_recon_sample=og_img)
        self._save(og_img_1, str(self.results_folder / f'og1-{extra_path}.png'))
        self._save(og_img_2, str(self.results_folder / f'og2-{extra_path}.png'))
        return self._frames(og_img, extra_path, x0_list, xt_list)

    def test_from_random(self, extra_path):                            # DEFADE:885-920: a batch scaled by 0.9 as the start
        og_img = self._next_batch() * 0.9
        x0_list, xt_list = self.ema_core.all_sample(batch_size=self.batch_size, faded_recon_sample=og_img)
        return self._frames(og_img, extra_path, x0_list, xt_list)

    def controlled_direct_reconstruct(self, extra_path):               # DEFADE:922-940
        torch.manual_seed(0)
        og_img = self._next_batch()
        xt, direct_recons, all_images = self.ema_core.sample(batch_size=self.batch_size, faded_recon_sample=og_img)
        for name, im in (('og', og_img), ('recon', all_images), ('direct_recons', direct_recons), ('xt', xt)):
            self._save(im, str(self.results_folder / f'sample-{name}-{extra_path}.png'))
        self.save()
        return xt, direct_recons, all_images

    def test_from_data_save_results(self, batch_size=100, chunk=32):  # DEFADE:1146-1244: four folders of per-image PNGs
        from .evaluate import _create_folder
        all_samples = torch.cat(list(self._dataset_batches(batch_size)), dim=0)
        folders = [f'{self.results_folder}_{n}/' for n in ('orig', 'blur', 'deblur', 'd_deblur')]
        for f in folders:
            _create_folder(f)
        cnt = 0
        while cnt < all_samples.shape[0]:
            og_img = all_samples[cnt: cnt + chunk].to(self.device).float().contiguous()
            x0_list, xt_list = self.ema_core.all_sample(batch_size=og_img.shape[0], faded_recon_sample=og_img, times=None)
            sets = (og_img.cpu(), xt_list[0].cpu(), x0_list[-1].cpu(), x0_list[0].cpu())
            for i in range(og_img.shape[0]):
                for f, t in zip(folders, sets):
                    self._save(t[i:i + 1].repeat(1, 3 // t.shape[1], 1, 1), f'{f}{cnt + i}.png', nrow=1)
            cnt += og_img.shape[0]
        return cnt

"""Input pipeline: the device-side cache + crop/mirror/convert kernel against the reference's PIL transform
chain (Dataset_Aug1 / Dataset, deblurring_diffusion_pytorch.py:983-1026), and the host DataLoader path of the Trainer."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from emu_util import P


def _write_images(folder, n=7, seed=0):
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(folder, exist_ok=True)
    sizes = [(40, 52), (64, 64), (30, 30), (57, 33), (20, 48), (36, 36), (25, 31), (44, 44)]
    for i in range(n):
        w, h = sizes[i % len(sizes)]
        # smooth + noisy content so that the bilinear resize is not trivial
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(xx * 255 // w), (yy * 255 // h), ((xx + yy) * 255 // (w + h))], -1).astype(np.int32) + rng.randint(-40, 40, (h, w, 3))
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(folder, f"im{i:02d}.png"))


def _reference_item(path, size, oy=None, ox=None, flip=False, resize='sq112', pad=0, rgb=False):
    """The reference's transform chains written with PIL calls only (torchvision is not installed), independently of the package's
    own helpers: [convert('RGB')] -> Resize((S, S)) | Resize(s) | nothing -> [RandomCrop's zero border] -> crop -> [mirror] -> ToTensor
    -> t * 2 - 1.  oy/ox None = torchvision's CenterCrop."""
    from PIL import Image, ImageOps
    img = Image.open(path)
    if rgb:
        img = img.convert('RGB')
    if resize == 'sq112':
        S = int(size * 1.12)
        img = img.resize((S, S), Image.BILINEAR)
    elif resize == 'short':                                         # torchvision F.resize(img, int): shorter edge -> size
        w, h = img.size
        if not ((w <= h and w == size) or (h <= w and h == size)):
            if w < h:
                img = img.resize((size, int(size * h / w)), Image.BILINEAR)
            else:
                img = img.resize((int(size * w / h), size), Image.BILINEAR)
    if pad:
        img = ImageOps.expand(img, border=pad, fill=0)              # F.pad(img, 4, fill=0, 'constant')
    w, h = img.size
    if oy is None:
        oy, ox = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))     # torchvision CenterCrop
    img = img.crop((ox, oy, ox + size, oy + size))
    if flip:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    a = np.asarray(img).copy()
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(a).permute(2, 0, 1).float().div(255)      # ToTensor
    return t * 2 - 1


def test_augment_kernel(be):
    torch.manual_seed(0)
    N, S, C, H, B = 5, 18, 3, 16, 6
    cache = torch.randint(0, 256, (N, S, S, C), dtype=torch.uint8)
    idx = torch.tensor([4, 0, 2, 2, 1, 3])
    oy, ox = torch.tensor([0, 2, 1, 2, 0, 1], dtype=torch.int32), torch.tensor([2, 0, 1, 2, 1, 0], dtype=torch.int32)
    flip = torch.tensor([0, 1, 1, 0, 1, 0], dtype=torch.int32)
    out = be.empty(B, C, H, H)
    be.L.cdf_augment_batch(P(be.to(cache)), N, S, C, P(be.to(idx)), P(be.to(oy)), P(be.to(ox)), P(be.to(flip)), P(out), B, H, H, be.stream())
    for b in range(B):
        ref = cache[idx[b], oy[b]:oy[b] + H, ox[b]:ox[b] + H]
        if flip[b]:
            ref = ref.flip(1)
        ref = ref.permute(2, 0, 1).float().div(255) * 2 - 1
        assert torch.equal(out[b].cpu(), ref), b                                         # integer gather + ToTensor arithmetic: bit-exact


def test_augment_kernel_border_and_rectangular_cache(be):
    """cdf_augment_batch_pad: RandomCrop(s, padding=4) on a rectangular cache -- every corner of the zero border."""
    torch.manual_seed(1)
    N, SH, SW, C, H, pad = 3, 20, 27, 3, 16, 4
    cache = torch.randint(1, 256, (N, SH, SW, C), dtype=torch.uint8)
    padded = torch.nn.functional.pad(cache.permute(0, 3, 1, 2), (pad, pad, pad, pad)).permute(0, 2, 3, 1)
    oys, oxs = [0, SH + 2 * pad - H, 0, SH + 2 * pad - H, 5, 3], [0, 0, SW + 2 * pad - H, SW + 2 * pad - H, 7, 19]
    B = len(oys)
    idx = torch.tensor([0, 1, 2, 0, 1, 2])
    flip = torch.tensor([0, 1, 0, 1, 1, 0], dtype=torch.int32)
    out = be.empty(B, C, H, H)
    be.L.cdf_augment_batch_pad(P(be.to(cache)), N, SH, SW, C, pad, P(be.to(idx)), P(be.to(torch.tensor(oys, dtype=torch.int32))),
                               P(be.to(torch.tensor(oxs, dtype=torch.int32))), P(be.to(flip)), P(out), B, H, H, be.stream())
    for b in range(B):
        ref = padded[idx[b], oys[b]:oys[b] + H, oxs[b]:oxs[b] + H]
        if flip[b]:
            ref = ref.flip(1)
        ref = ref.permute(2, 0, 1).float().div(255) * 2 - 1
        assert torch.equal(out[b].cpu(), ref), b
    assert float(out[0, :, 0, 0].max()) == -1.0                                          # the border converts to -1


@pytest.fixture
def emu():
    from colddiff import runtime
    from emu_util import install_emu
    install_emu()
    yield
    runtime._lib_override = None


@pytest.mark.parametrize("size", [16, 28, 32, 64])
def test_device_pipeline_equals_reference_transforms(tmp_path, emu, size):
    """size 16: S - size = 1 (round and floor agree); 28 / 32: 3 -> torchvision starts the centre crop at 2, floor division at 1;
    64: 7 -> 4 vs 3 (the same odd margin class as 128: 143 - 128 = 15 -> 8 vs 7)."""
    from colddiff.trainer import Dataset, DeviceImageCache, DeviceLoader, CENTER112, center_offset
    folder = str(tmp_path / "imgs")
    _write_images(folder, 7)
    S = int(size * 1.12)
    cache = DeviceImageCache(folder, size, torch.device("cpu"), decode_threads=3, recipe=CENTER112)
    assert len(cache) == 7 and cache.S == S and cache.data.shape == (7, S, S, 3)
    assert center_offset(143, 128) == 8 and center_offset(71, 64) == 4 and center_offset(35, 32) == 2 and center_offset(31, 28) == 2
    # CenterCrop path (class Dataset): every item of one epoch, bit for bit, also against the host Dataset of the package
    dl = DeviceLoader(cache, batch_size=3, shuffle=False)
    host = Dataset(folder, size)
    assert [str(p) for p in host.paths] == [str(p) for p in cache.paths]
    got = torch.cat([next(dl), next(dl)])                                                  # 6 of 7 images (drop_last)
    for i in range(6):
        assert torch.equal(got[i], _reference_item(cache.paths[i], size)), i
        assert torch.equal(got[i], host[i]), i
        assert torch.equal(got[i], cache.item(i)), i
    assert dl.epoch == 1
    next(dl)                                                                               # 7th image alone is dropped: a new epoch starts
    assert dl.epoch == 2 and dl.pos == 3
    # RandomCrop + RandomHorizontalFlip path (class Dataset_Aug1): the kernel with explicit decisions
    idx = torch.tensor([6, 1, 3, 3])
    m = S - size
    oy, ox = torch.tensor([m, 0, 0, 1], dtype=torch.int32), torch.tensor([0, m, 0, 1], dtype=torch.int32)
    flip = torch.tensor([1, 0, 1, 0], dtype=torch.int32)
    out = cache.batch(idx, oy, ox, flip)
    for b in range(4):
        assert torch.equal(out[b], _reference_item(cache.paths[idx[b]], size, int(oy[b]), int(ox[b]), bool(flip[b]))), b
    # the augmenting loader draws offsets in range and both mirror states; a shuffled epoch is a permutation
    dl = DeviceLoader(cache, batch_size=7, augment=True, shuffle=True, seed=5)
    seen = set()
    for _ in range(6):
        x = next(dl)
        assert x.shape == (7, 3, size, size) and x.min() >= -1 and x.max() <= 1
        seen.add(tuple(sorted(dl.order.tolist())))
    assert seen == {tuple(range(7))}


def _write_square(folder, n, side, mode="RGB", seed=3):
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(folder, exist_ok=True)
    for i in range(n):
        a = rng.randint(0, 256, (side, side) if mode == "L" else (side, side, 3)).astype(np.uint8)
        Image.fromarray(a, mode).save(os.path.join(folder, f"s{i:02d}.png"))


def _replay(loader_seed, B, span, flip=True, device="cpu"):
    """The decisions a DeviceLoader with this seed draws for its first batch."""
    g = torch.Generator(device=device)
    g.manual_seed(loader_seed)
    oy = torch.randint(0, span[0], (B,), generator=g, dtype=torch.int32) if span[0] > 1 or span[1] > 1 else torch.zeros(B, dtype=torch.int32)
    ox = torch.randint(0, span[1], (B,), generator=g, dtype=torch.int32) if span[0] > 1 or span[1] > 1 else torch.zeros(B, dtype=torch.int32)
    fl = (torch.rand((B,), generator=g) < 0.5) if flip else torch.zeros(B, dtype=torch.bool)
    return oy, ox, fl


def test_per_package_dataset_chains(tmp_path, emu):
    """Each package's `dataset=` table selects ITS transform chain (VERDICT r2 missing #4): denoising converts to RGB and augments
    only for 'train' (DENOISE:544-589, 655-660); resolution 'flower' = Resize(s) + RandomCrop(s, padding=4) (RESOL:817-831, 878-885);
    defading 'cifar10' = RandomCrop(s, padding=4) on the file, default = Resize(s) + CenterCrop(s) (DEFADE:557-648, 685-692)."""
    from colddiff import trainer as T
    from colddiff.trainer import DeviceImageCache, DeviceLoader, Dataset
    # -- the tables ------------------------------------------------------------------------------------------------
    assert T.Trainer.recipe_for('celebA').name == 'Dataset_Aug1' and T.Trainer.recipe_for(None).crop == 'center'
    assert T.DenoiseTrainer.recipe_for('celebA').crop == 'center' and T.DenoiseTrainer.recipe_for('celebA').rgb
    assert T.DenoiseTrainer.recipe_for('train').crop == 'random' and T.DenoiseTrainer.recipe_for('train').rgb
    assert T.ResolutionTrainer.recipe_for('flower') is T.AUG2 and T.ResolutionTrainer.recipe_for('cifar10') is T.AUG1
    assert T.ResolutionTrainer.recipe_for('mnist').crop == 'center' and not T.ResolutionTrainer.drop_last
    assert T.DefadeTrainer.recipe_for('cifar10') is T.CIFAR_PAD and T.DefadeTrainer.recipe_for('mnist') is T.CENTER_SHORT
    assert T.DefadeTrainer.recipe_for('celebA').crop == 'random' and T.DefadeTrainer.recipe_for('celebA_test').crop == 'center'
    assert T.DemixTrainer.recipe_for('train').rgb and T.DefadeGenTrainer.recipe_for(None).rgb
    import denoising_diffusion_pytorch, resolution_diffusion_pytorch, defading_diffusion_pytorch
    assert denoising_diffusion_pytorch.Trainer is T.DenoiseTrainer and resolution_diffusion_pytorch.Trainer is T.ResolutionTrainer
    assert defading_diffusion_pytorch.Trainer is T.DefadeTrainer

    # -- DatasetCifar10: the files' own 20 x 20, 4 pixels of border, crop 20 --------------------------------------------
    size = 20
    f1 = str(tmp_path / "cifar")
    _write_square(f1, 5, size)
    cache = DeviceImageCache(f1, size, torch.device("cpu"), recipe=T.CIFAR_PAD)
    assert cache.data.shape == (5, 20, 20, 3) and cache.span() == (9, 9)
    dl = DeviceLoader(cache, batch_size=5, shuffle=False, seed=11)
    got = next(dl)
    oy, ox, fl = _replay(11, 5, (9, 9))
    assert oy.max() > 4 or ox.max() > 4                                          # some crop reaches into the border
    for b in range(5):
        assert torch.equal(got[b], _reference_item(cache.paths[b], size, int(oy[b]), int(ox[b]), bool(fl[b]), resize='none', pad=4)), b
    # host path of the same recipe: border value and range
    x = Dataset(f1, size, recipe=T.CIFAR_PAD)[0]
    assert x.shape == (3, size, size) and x.min() >= -1

    # -- Dataset_Aug2 on square 30 x 30 files: Resize(20) -> 20 x 20, border 4 ------------------------------------------
    f2 = str(tmp_path / "flower")
    _write_square(f2, 4, 30)
    cache = DeviceImageCache(f2, size, torch.device("cpu"), recipe=T.AUG2)
    assert cache.data.shape == (4, 20, 20, 3)
    dl = DeviceLoader(cache, batch_size=3, shuffle=False, seed=5, drop_last=False)
    got = next(dl)
    oy, ox, fl = _replay(5, 3, (9, 9))
    for b in range(3):
        assert torch.equal(got[b], _reference_item(cache.paths[b], size, int(oy[b]), int(ox[b]), bool(fl[b]), resize='short', pad=4)), b
    assert next(dl).shape[0] == 1                                                # RESOL:887 has no drop_last: the short batch is kept
    assert next(dl).shape[0] == 3 and dl.epoch == 2

    # -- defading plain Dataset: Resize(s) keeps the aspect ratio -> ragged folder falls back to the host path --------
    f3 = str(tmp_path / "ragged")
    _write_images(f3, 4)
    with pytest.raises(T.CacheUnfit):
        DeviceImageCache(f3, size, torch.device("cpu"), recipe=T.CENTER_SHORT)
    host = Dataset(f3, size, recipe=T.CENTER_SHORT)
    for i in range(4):
        assert torch.equal(host[i], _reference_item(host.paths[i], size, resize='short')), i
    # ... and a uniform folder is cached: 30 x 30 -> 20 x 20, centre crop = the whole image
    cache = DeviceImageCache(f2, size, torch.device("cpu"), recipe=T.CENTER_SHORT)
    assert torch.equal(cache.item(2), _reference_item(cache.paths[2], size, resize='short'))

    # -- denoising: greyscale files become 3 identical channels (convert('RGB')) ------------------------------------
    f4 = str(tmp_path / "grey")
    _write_square(f4, 3, 40, mode="L")
    rec = T.DenoiseTrainer.recipe_for('celebA')
    cache = DeviceImageCache(f4, 32, torch.device("cpu"), recipe=rec)
    assert cache.channels == 3
    x = cache.item(1)
    assert torch.equal(x, _reference_item(cache.paths[1], 32, rgb=True)) and torch.equal(x[0], x[2])
    assert DeviceImageCache(f4, 32, torch.device("cpu"), recipe=T.CENTER112).channels == 1       # deblurring keeps the file's mode


def test_device_loader_rank_shards():
    """DistributedSampler semantics: the ranks split ONE permutation of the epoch; no image is seen twice in an epoch."""
    from colddiff.trainer import DeviceLoader

    from colddiff.trainer import AUG1

    class FakeCache:
        S, image_size, recipe = 18, 16, AUG1
        data = torch.zeros(1)

        def __len__(self):
            return 11

        def center(self):
            return 1, 1

        def span(self):
            return 3, 3

    loaders = [DeviceLoader(FakeCache(), 2, seed=3, rank=r, world=2) for r in range(2)]
    for ld in loaders:
        ld._new_epoch()
    a, b = loaders[0].order.tolist(), loaders[1].order.tolist()
    assert len(a) == len(b) == 5 and not set(a) & set(b)
    assert loaders[0].gen.initial_seed() != loaders[1].gen.initial_seed()                 # ranks draw different crops / mirrors
    loaders[0]._new_epoch()
    assert loaders[0].order.tolist() != a                                                  # reshuffled every epoch (set_epoch)
    # nothing two-rank-shaped: 4 and 8 ranks (the node of SURVEY 8(e)) -- shards pairwise disjoint, equal-sized, covering all but
    # the n % world images DistributedSampler(drop_last) leaves out, per-rank augmentation streams all different
    FakeCache.__len__ = lambda self: 37
    for world in (4, 8):
        lds = [DeviceLoader(FakeCache(), 2, seed=3, rank=r, world=world) for r in range(world)]
        for ld in lds:
            ld._new_epoch()
        shards = [ld.order.tolist() for ld in lds]
        assert all(len(sh) == 37 // world for sh in shards)
        union = set().union(*shards)
        assert len(union) == world * (37 // world) and union <= set(range(37))
        assert len({ld.gen.initial_seed() for ld in lds}) == world


@pytest.mark.parametrize("device_data", [False, True])
def test_trainer_on_an_image_folder(tmp_path, emu, device_data):
    """Trainer(folder, dataset='celebA') end to end on a folder of PNGs: the host Dataset_Aug1 + DataLoader path (worker processes,
    cycle() wrap-around) and the device-side cache path."""
    from deblurring_diffusion_pytorch import GaussianDiffusion, Trainer, Unet
    folder = str(tmp_path / "imgs")
    _write_images(folder, 5)
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        net = Unet(dim=8, dim_mults=(1, 2), channels=3)
        d = GaussianDiffusion(net, image_size=16, device_of_kernel="cpu", channels=3, timesteps=3, kernel_size=3, kernel_std=0.5)
        tr = Trainer(d, folder, image_size=16, train_batch_size=2, train_lr=1e-3, train_num_steps=3, gradient_accumulate_every=2,
                     dataset="celebA", results_folder=str(tmp_path / "res"), num_workers=2, device_data=device_data)
    assert (tr.ds is not None) and len(tr.ds) == 5
    assert type(tr.dl).__name__ == ("DeviceLoader" if device_data else "generator")
    w0 = tr.arena.data.clone()
    losses = []
    for _ in range(3):                                                                     # 6 micro-batches of 2 from 5 images: wraps the epoch twice
        losses.append(float(tr.train_step()))
        tr.step += 1
    assert all(np.isfinite(losses)) and not torch.equal(tr.arena.data, w0)
    b = tr._next_batch()
    assert b.shape == (2, 3, 16, 16) and b.dtype == torch.float32 and -1 <= float(b.min()) and float(b.max()) <= 1

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


def _reference_item(path, size, oy=None, ox=None, flip=False):
    """The reference's transform chain written with PIL calls only (torchvision is not installed): Resize((S, S)) -> crop ->
    [mirror] -> ToTensor -> t * 2 - 1.  oy/ox None = CenterCrop."""
    from PIL import Image
    S = int(size * 1.12)
    img = Image.open(path).resize((S, S), Image.BILINEAR)
    if oy is None:
        oy = ox = int(round((S - size) / 2.0))                     # torchvision CenterCrop
    img = img.crop((ox, oy, ox + size, oy + size))
    if flip:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    t = torch.from_numpy(np.asarray(img).copy()).permute(2, 0, 1).float().div(255)     # ToTensor
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


@pytest.fixture
def emu():
    from colddiff import runtime
    from emu_util import install_emu
    install_emu()
    yield
    runtime._lib_override = None


def test_device_pipeline_equals_reference_transforms(tmp_path, emu):
    from colddiff.trainer import Dataset, DeviceImageCache, DeviceLoader
    folder = str(tmp_path / "imgs")
    _write_images(folder, 7)
    size = 16
    cache = DeviceImageCache(folder, size, torch.device("cpu"), decode_threads=3)
    assert len(cache) == 7 and cache.S == 17 and cache.data.shape == (7, 17, 17, 3)
    # CenterCrop path (class Dataset): every item of one epoch, bit for bit, also against the host Dataset of the package
    dl = DeviceLoader(cache, batch_size=3, augment=False, shuffle=False)
    host = Dataset(folder, size)
    assert [str(p) for p in host.paths] == [str(p) for p in cache.paths]
    got = torch.cat([next(dl), next(dl)])                                                  # 6 of 7 images (drop_last)
    for i in range(6):
        assert torch.equal(got[i], _reference_item(cache.paths[i], size)), i
        assert torch.equal(got[i], host[i]), i
    assert dl.epoch == 1
    next(dl)                                                                               # 7th image alone is dropped: a new epoch starts
    assert dl.epoch == 2 and dl.pos == 3
    # RandomCrop + RandomHorizontalFlip path (class Dataset_Aug1): the kernel with explicit decisions
    idx = torch.tensor([6, 1, 3, 3])
    oy, ox = torch.tensor([1, 0, 0, 1], dtype=torch.int32), torch.tensor([0, 1, 0, 1], dtype=torch.int32)
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


def test_device_loader_rank_shards():
    """DistributedSampler semantics: the ranks split ONE permutation of the epoch; no image is seen twice in an epoch."""
    from colddiff.trainer import DeviceLoader

    class FakeCache:
        S, image_size = 18, 16
        data = torch.zeros(1)

        def __len__(self):
            return 11

    loaders = [DeviceLoader(FakeCache(), 2, augment=True, seed=3, rank=r, world=2) for r in range(2)]
    for ld in loaders:
        ld._new_epoch()
    a, b = loaders[0].order.tolist(), loaders[1].order.tolist()
    assert len(a) == len(b) == 5 and not set(a) & set(b)
    assert loaders[0].gen.initial_seed() != loaders[1].gen.initial_seed()                 # ranks draw different crops / mirrors
    loaders[0]._new_epoch()
    assert loaders[0].order.tolist() != a                                                  # reshuffled every epoch (set_epoch)


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

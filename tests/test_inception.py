"""FID feature extractor (deblurring-diffusion-pytorch/Fid/inception.py, Fid/fid_score.py): the pooling /
resize kernels against ATen, every Inception block and the stem against the CPU restatement (oracle/inception_ref.py) with random
weights AND random BatchNorm statistics loaded through the pytorch-fid key layout, the whole 299 x 299 network on the MI355X, and
calculate_fid_given_samples end to end."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from emu_util import P
from oracle import inception_ref as R


def test_pool_and_resize_kernels(be):
    torch.manual_seed(0)
    B, H, W, C = 2, 9, 7, 8
    x = torch.randn(B, H, W, C)
    xd = be.to(x)
    xn = x.permute(0, 3, 1, 2)
    for k, s, p, mode, ref in ((3, 2, 0, 0, F.max_pool2d(xn, 3, 2)), (3, 1, 1, 0, F.max_pool2d(xn, 3, 1, 1)),
                               (3, 1, 1, 1, F.avg_pool2d(xn, 3, 1, 1, count_include_pad=False)), (3, 2, 0, 1, F.avg_pool2d(xn, 3, 2))):
        OH, OW = ref.shape[2:]
        wide = be.zeros(B, OH, OW, C + 4)                                     # written as a channel slice of a wider buffer
        be.L.cdf_pool2d(P(xd), C, P(wide) + 16, C + 4, B, H, W, C, k, s, p, mode, be.stream())
        got = wide.cpu()[..., 4:].permute(0, 3, 1, 2)
        if mode == 0:
            assert torch.equal(got, ref), (k, s, p)
        else:
            assert (got - ref).abs().max() <= 1e-6, (k, s, p)
        assert float(wide.cpu()[..., :4].abs().max()) == 0.0
    y = be.empty(B, C)
    be.L.cdf_global_avgpool(P(xd), C, P(y), C, B, H * W, C, be.stream())
    assert (y.cpu() - x.mean((1, 2))).abs().max() <= 1e-6
    img = torch.rand(2, 3, 11, 13)
    for (OH, OW) in ((29, 31), (11, 13), (5, 6)):
        out = be.zeros(2, OH, OW, 4)
        be.L.cdf_resize_bilinear_nhwc(P(be.to(img)), P(out), 4, 2, 3, 11, 13, OH, OW, 2.0, -1.0, be.stream())
        ref = 2 * F.interpolate(img, size=(OH, OW), mode='bilinear', align_corners=False) - 1
        assert (out.cpu()[..., :3].permute(0, 3, 1, 2) - ref).abs().max() <= 2e-6, (OH, OW)
        assert float(out.cpu()[..., 3].abs().max()) == 0.0


@pytest.fixture
def emu():
    from colddiff import runtime
    from emu_util import install_emu
    install_emu()
    yield
    runtime._lib_override = None


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def _check_block(mine, ref, x, tol=2e-4):
    mine.load_state_dict(ref.state_dict())
    with torch.no_grad():
        want = ref(x)
        got = mine.run(_nhwc(x)).permute(0, 3, 1, 2)
    assert got.shape == want.shape
    err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
    assert err <= tol, err
    return err


def test_inception_blocks_on_the_simulator(emu):
    """Stem layers (stride 2 without padding, 1x1, padded 3x3 on a 3-channel image) and the A / B / C / D blocks (5x5, 1x7 / 7x1
    convs, stride-2 reductions, both pool flavours, concat-free branch outputs) against the restatement, small spatial sizes."""
    from colddiff import inception as I
    torch.manual_seed(1)
    ref = R.randomise(R.FidInception3(), seed=3)
    x = torch.rand(1, 3, 15, 15) * 2 - 1
    x4 = torch.zeros(1, 15, 15, 4)
    x4[..., :3] = _nhwc(x)
    stem = [I.BasicConv2d(3, 32, 3, stride=2), I.BasicConv2d(32, 32, 3), I.BasicConv2d(32, 64, 3, padding=1)]
    h, hr = x4, x
    for m, r in zip(stem, (ref.Conv2d_1a_3x3, ref.Conv2d_2a_3x3, ref.Conv2d_2b_3x3)):
        m.load_state_dict(r.state_dict())
        with torch.no_grad():
            h, hr = m.run(h), r(hr)
        assert (h.permute(0, 3, 1, 2) - hr).abs().max() <= 2e-4 * max(1.0, hr.abs().max().item())
    _check_block(I.FIDInceptionA(192, 32), ref.Mixed_5b, torch.randn(1, 192, 5, 5))
    _check_block(I.InceptionB(288), ref.Mixed_6a, torch.randn(1, 288, 5, 5))
    _check_block(I.FIDInceptionC(768, 128), ref.Mixed_6b, torch.randn(1, 768, 4, 4))
    _check_block(I.InceptionD(768), ref.Mixed_7a, torch.randn(1, 768, 5, 5))
    _check_block(I.FIDInceptionE_1(1280), ref.Mixed_7b, torch.randn(1, 1280, 3, 3))
    _check_block(I.FIDInceptionE_2(2048), ref.Mixed_7c, torch.randn(1, 2048, 3, 3))


def test_fid_state_dict_layouts(emu):
    """The pytorch-fid file layout maps onto the wrapper's `blocks.{i}.{j}` tree; a wrapper truncated at block 1 loads the same file;
    a missing weight file is an error that names the download."""
    from colddiff.inception import InceptionV3
    ref = R.randomise(R.FidInception3(), seed=5)
    net = InceptionV3([3], weights=ref.state_dict())
    sd = net.state_dict()
    assert torch.equal(sd['blocks.0.0.conv.weight'], ref.Conv2d_1a_3x3.conv.weight)
    assert torch.equal(sd['blocks.2.3.branch3x3.bn.running_var'], ref.Mixed_6a.branch3x3.bn.running_var)
    assert torch.equal(sd['blocks.3.2.branch_pool.conv.weight'], ref.Mixed_7c.branch_pool.conv.weight)
    assert sum(p.numel() for p in net.parameters()) == 21785568           # = torchvision inception_v3 (27 161 264) - fc - AuxLogits
    small = InceptionV3([0, 1], weights=ref.state_dict())
    assert len(small.blocks) == 2 and small.last_needed_block == 1
    with pytest.raises(FileNotFoundError, match="pt_inception-2015-12-05"):
        InceptionV3([3], weights="/nonexistent/file.pth")
    assert InceptionV3.BLOCK_INDEX_BY_DIM == {64: 0, 192: 1, 768: 2, 2048: 3}


@pytest.mark.gpu
def test_inception_full_network_on_the_gpu():
    """The whole extractor as the metric step runs it: 64 x 64 images in (0, 1) -> resize 299 -> all four output blocks, against the
    CPU restatement.  Tolerance 2e-3 of each map's max (fp32-equivalent bf16x3 GEMMs through ~50 layers)."""
    from colddiff.inception import InceptionV3
    torch.manual_seed(2)
    ref = R.randomise(R.FidInception3(), seed=7)
    x = torch.rand(3, 3, 64, 64)
    net = InceptionV3([0, 1, 2, 3], weights=ref.state_dict()).to("cuda:0")
    with torch.no_grad():
        got = net(x.to("cuda:0"))
        want = R.features(ref, x, output_blocks=(0, 1, 2, 3))
    assert [tuple(g.shape) for g in got] == [tuple(w.shape) for w in want] and got[3].shape == (3, 2048, 1, 1)
    for i, (g, w) in enumerate(zip(got, want)):
        err = (g.cpu() - w).abs().max().item() / w.abs().max().item()
        print("block", i, tuple(w.shape), "rel err", err)
        assert err <= 2e-3, (i, err)
    # no resize / no normalisation path
    net2 = InceptionV3([2], resize_input=False, normalize_input=False, weights=ref.state_dict()).to("cuda:0")
    xs = torch.rand(2, 3, 91, 75)
    with torch.no_grad():
        g = net2(xs.to("cuda:0"))[0].cpu()
        w = R.features(ref, xs, output_blocks=(2,), resize_input=False, normalize_input=False)[0]
    assert (g - w).abs().max().item() <= 2e-3 * w.abs().max().item()


@pytest.mark.gpu
def test_fid_end_to_end_on_the_gpu():
    """calculate_fid_given_samples builds the extractor itself (weights from $COLDDIFF_FID_WEIGHTS) like Fid/fid_score.py:331-343;
    the value equals the Frechet distance of the restatement's features (dims=192: the covariance is well conditioned for 48 samples
    only at a low dimension)."""
    import os
    import tempfile
    from colddiff import metrics
    ref = R.randomise(R.FidInception3(), seed=11)
    torch.manual_seed(4)
    a, b = torch.rand(48, 3, 32, 32), (torch.rand(48, 3, 32, 32) * 0.8 + 0.1)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "pt_inception.pth")
        torch.save(ref.state_dict(), path)
        os.environ["COLDDIFF_FID_WEIGHTS"] = path
        try:
            fid = metrics.calculate_fid_given_samples([a, b], batch_size=16, device="cuda:0", dims=192)
        finally:
            del os.environ["COLDDIFF_FID_WEIGHTS"]
    with torch.no_grad():
        fa = R.features(ref, a, output_blocks=(1,))[0].mean((2, 3)).double().numpy()
        fb = R.features(ref, b, output_blocks=(1,))[0].mean((2, 3)).double().numpy()
    want = metrics.calculate_frechet_distance(fa.mean(0), np.cov(fa, rowvar=False), fb.mean(0), np.cov(fb, rowvar=False))
    print("fid", fid, "restatement", want)
    assert abs(fid - want) <= 2e-2 * abs(want) + 1e-3

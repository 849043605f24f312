"""`InceptionV3` — the FID feature extractor of the reference's metric step on the HIP kernels.

Mirrors deblurring-diffusion-pytorch/Fid/inception.py:16-328 (pytorch-fid's network: torchvision's Inception3 with the FID patches —
average pools that exclude the padding, a max pool in the last block, 1008-way logits dropped): same constructor
(`output_blocks, resize_input, normalize_input, requires_grad, use_fid_inception`), same `BLOCK_INDEX_BY_DIM`, `forward(inp [B,3,H,W]
in (0,1)) -> list of feature maps` (NCHW, pool3 as [B,2048,1,1]), and the same module tree (`blocks.{i}.{j}.<branch>.{conv,bn}`), so a
`state_dict` of the reference wrapper loads unchanged; `load_fid_state_dict` takes the pytorch-fid / torchvision key layout of
`pt_inception-2015-12-05-6726825d.pth` (`Conv2d_1a_3x3.conv.weight`, `Mixed_5b.branch1x1.bn.running_var`, ..., `fc.*` ignored).

torchvision is not a dependency: the architecture (InceptionA-E channel tables) is restated from torchvision.models.inception.
Inference only.  Every `BasicConv2d` (conv, bias-free -> BatchNorm(eps=1e-3, running statistics) -> ReLU) is ONE launch of the conv
GEMM kernels: the BatchNorm is folded into weight and bias once per weight version, ReLU is the epilogue (act = 3), and every branch
writes its channel slice of the block's output directly (no torch.cat).  Pools / global average / the 299x299 bilinear resize are
`k_pool.hip`.  The pretrained weights are a download upstream (FID_WEIGHTS_URL); offline they are read from a file
(`weights=` / $COLDDIFF_FID_WEIGHTS / the torch hub cache) and a missing file is an error, never a silent random network.
"""
import os
from functools import lru_cache

import torch
from torch import nn

from . import convdesc as cd
from . import ops
from . import runtime as rt
from .runtime import P

FID_WEIGHTS_URL = 'https://github.com/mseitzer/pytorch-fid/releases/download/fid_weights/pt_inception-2015-12-05-6726825d.pth'  # noqa: E501
FID_WEIGHTS_FILE = 'pt_inception-2015-12-05-6726825d.pth'
ACT_RELU = 3


MAX_TAPS = 16          # taps per gather-GEMM launch (CDF_MAX_TAPS in csrc/k_conv.hip)


@lru_cache(maxsize=None)
def _tap_parts(H, W, k, stride, pad):
    """conv_fwd's plan cut into launches of <= MAX_TAPS taps each (same output grid, disjoint tap subsets of the packed weight)."""
    OH, OW = (H + 2 * pad[0] - k[0]) // stride + 1, (W + 2 * pad[1] - k[1]) // stride + 1
    taps = [(ky - pad[0], kx - pad[1], ky * k[1] + kx) for ky in range(k[0]) for kx in range(k[1])]
    return tuple(cd.GemmPlan(H, W, OH, OW, OH, OW, 1, stride, [(0, 0, taps[i:i + MAX_TAPS])], len(taps)) for i in range(0, len(taps), MAX_TAPS))


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class BasicConv2d(nn.Module):
    """torchvision.models.inception.BasicConv2d: Conv2d(bias=False) -> BatchNorm2d(eps=0.001) -> ReLU; parameter container + one
    GEMM launch with the BatchNorm folded in."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(out_channels, eps=0.001)
        self.cin, self.cout = in_channels, out_channels
        self.k, self.stride, self.pad = _pair(kernel_size), stride, _pair(padding)
        self._folded = None

    def folded(self):
        """(weight', bias') with y = relu(conv(x, weight') + bias') == relu(bn(conv(x, weight))) in eval mode."""
        src = (self.conv.weight, self.bn.weight, self.bn.bias, self.bn.running_mean, self.bn.running_var)
        key = tuple((t.data_ptr(), t._version) for t in src)
        if self._folded is None or self._folded[0] != key:
            with torch.no_grad():
                s = self.bn.weight / torch.sqrt(self.bn.running_var + self.bn.eps)
                w = (self.conv.weight * s[:, None, None, None]).contiguous()
                b = (self.bn.bias - self.bn.running_mean * s).contiguous()
            self._folded = (key, w, b)
        return self._folded[1], self._folded[2]

    def out_hw(self, H, W):
        return ((H + 2 * self.pad[0] - self.k[0]) // self.stride + 1, (W + 2 * self.pad[1] - self.k[1]) // self.stride + 1)

    def run(self, x, out=None):
        """x: NHWC feature map [B,H,W,>=cin]; out: optional [B,OH,OW,cout] view (a channel slice of the block's output)."""
        B, H, W, _ = x.shape
        w, b = self.folded()
        plan = cd.conv_fwd(H, W, self.k[0], self.k[1], self.stride, self.pad[0], self.pad[1], self.pad[0], self.pad[1])
        K = self.cin * self.k[0] * self.k[1]
        sp = rt.precision != "f32" and K >= 64 and self.cout >= 64
        wp = ops.packed(w, "conv_fwd_sp" if sp else "conv_fwd")
        if plan.ntaps_w <= MAX_TAPS:
            return ops.conv_gemm(plan, x, self.cin, wp, self.cout, y=out, bias=b, act=ACT_RELU)
        # the 5 x 5 layers (25 taps; the gather-GEMM takes 16 per launch): two launches over disjoint tap sets accumulate the
        # pre-activation, ReLU follows (3 of the 94 conv layers)
        y = None
        for part in _tap_parts(H, W, self.k, self.stride, self.pad):
            y = ops.conv_gemm(part, x, self.cin, wp, self.cout, y=y, bias=b if y is None else None, accumulate=0 if y is None else 1)
        if out is None:
            out = y
        rows = y.numel() // y.shape[-1]
        rt.lib().cdf_act_fwd(P(y), ops.ld_of(y), P(out), ops.ld_of(out), rows, self.cout, ACT_RELU, rt.stream(y))
        return out


def pool(x, k, stride, pad, mode, out=None):
    """mode 'max' | 'avg' (count_include_pad=False) on an NHWC map."""
    B, H, W, C = x.shape
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = out if out is not None else torch.empty((B, OH, OW, C), device=x.device, dtype=torch.float32)
    rt.lib().cdf_pool2d(P(rt.check(x)), ops.ld_of(x), P(y), ops.ld_of(y), B, H, W, C, k, stride, pad, 0 if mode == 'max' else 1, rt.stream(x))
    return y


class MaxPool2d(nn.Module):
    """nn.MaxPool2d(kernel_size=3, stride=2) of the stem (inception.py:91, 100)."""

    def __init__(self, kernel_size=3, stride=2):
        super().__init__()
        self.k, self.s = kernel_size, stride

    def run(self, x):
        return pool(x, self.k, self.s, 0, 'max')


class AdaptiveAvgPool2d(nn.Module):
    """nn.AdaptiveAvgPool2d((1, 1)) (inception.py:122)."""

    def run(self, x):
        B, H, W, C = x.shape
        y = torch.empty((B, 1, 1, C), device=x.device, dtype=torch.float32)
        rt.lib().cdf_global_avgpool(P(rt.check(x)), ops.ld_of(x), P(y), C, B, H * W, C, rt.stream(x))
        return y


class _Block(nn.Module):
    """An Inception block: branches (lists of layers) whose outputs are concatenated along the channels, in order."""

    def _cat_out(self, x, widths, OH, OW):
        B = x.shape[0]
        out = torch.empty((B, OH, OW, sum(widths)), device=x.device, dtype=torch.float32)
        views, o = [], 0
        for w in widths:
            views.append(out[..., o:o + w])
            o += w
        return out, views


class FIDInceptionA(_Block):                                   # torchvision InceptionA + the FID pool patch (inception.py:196-220)
    def __init__(self, in_channels, pool_features):
        super().__init__()
        self.branch1x1 = BasicConv2d(in_channels, 64, 1)
        self.branch5x5_1 = BasicConv2d(in_channels, 48, 1)
        self.branch5x5_2 = BasicConv2d(48, 64, 5, padding=2)
        self.branch3x3dbl_1 = BasicConv2d(in_channels, 64, 1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, 3, padding=1)
        self.branch_pool = BasicConv2d(in_channels, pool_features, 1)
        self.widths = (64, 64, 96, pool_features)

    def run(self, x):
        _, H, W, _ = x.shape
        out, v = self._cat_out(x, self.widths, H, W)
        self.branch1x1.run(x, v[0])
        self.branch5x5_2.run(self.branch5x5_1.run(x), v[1])
        self.branch3x3dbl_3.run(self.branch3x3dbl_2.run(self.branch3x3dbl_1.run(x)), v[2])
        self.branch_pool.run(pool(x, 3, 1, 1, 'avg'), v[3])
        return out


class InceptionB(_Block):                                      # torchvision InceptionB (Mixed_6a)
    def __init__(self, in_channels):
        super().__init__()
        self.branch3x3 = BasicConv2d(in_channels, 384, 3, stride=2)
        self.branch3x3dbl_1 = BasicConv2d(in_channels, 64, 1)
        self.branch3x3dbl_2 = BasicConv2d(64, 96, 3, padding=1)
        self.branch3x3dbl_3 = BasicConv2d(96, 96, 3, stride=2)
        self.widths = (384, 96, in_channels)

    def run(self, x):
        _, H, W, _ = x.shape
        OH, OW = self.branch3x3.out_hw(H, W)
        out, v = self._cat_out(x, self.widths, OH, OW)
        self.branch3x3.run(x, v[0])
        self.branch3x3dbl_3.run(self.branch3x3dbl_2.run(self.branch3x3dbl_1.run(x)), v[1])
        pool(x, 3, 2, 0, 'max', v[2])
        return out


class FIDInceptionC(_Block):                                   # torchvision InceptionC + the FID pool patch (inception.py:223-251)
    def __init__(self, in_channels, channels_7x7):
        super().__init__()
        c7 = channels_7x7
        self.branch1x1 = BasicConv2d(in_channels, 192, 1)
        self.branch7x7_1 = BasicConv2d(in_channels, c7, 1)
        self.branch7x7_2 = BasicConv2d(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7_3 = BasicConv2d(c7, 192, (7, 1), padding=(3, 0))
        self.branch7x7dbl_1 = BasicConv2d(in_channels, c7, 1)
        self.branch7x7dbl_2 = BasicConv2d(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_3 = BasicConv2d(c7, c7, (1, 7), padding=(0, 3))
        self.branch7x7dbl_4 = BasicConv2d(c7, c7, (7, 1), padding=(3, 0))
        self.branch7x7dbl_5 = BasicConv2d(c7, 192, (1, 7), padding=(0, 3))
        self.branch_pool = BasicConv2d(in_channels, 192, 1)
        self.widths = (192, 192, 192, 192)

    def run(self, x):
        _, H, W, _ = x.shape
        out, v = self._cat_out(x, self.widths, H, W)
        self.branch1x1.run(x, v[0])
        self.branch7x7_3.run(self.branch7x7_2.run(self.branch7x7_1.run(x)), v[1])
        t = self.branch7x7dbl_1.run(x)
        for m in (self.branch7x7dbl_2, self.branch7x7dbl_3, self.branch7x7dbl_4):
            t = m.run(t)
        self.branch7x7dbl_5.run(t, v[2])
        self.branch_pool.run(pool(x, 3, 1, 1, 'avg'), v[3])
        return out


class InceptionD(_Block):                                      # torchvision InceptionD (Mixed_7a)
    def __init__(self, in_channels):
        super().__init__()
        self.branch3x3_1 = BasicConv2d(in_channels, 192, 1)
        self.branch3x3_2 = BasicConv2d(192, 320, 3, stride=2)
        self.branch7x7x3_1 = BasicConv2d(in_channels, 192, 1)
        self.branch7x7x3_2 = BasicConv2d(192, 192, (1, 7), padding=(0, 3))
        self.branch7x7x3_3 = BasicConv2d(192, 192, (7, 1), padding=(3, 0))
        self.branch7x7x3_4 = BasicConv2d(192, 192, 3, stride=2)
        self.widths = (320, 192, in_channels)

    def run(self, x):
        _, H, W, _ = x.shape
        OH, OW = self.branch3x3_2.out_hw(H, W)
        out, v = self._cat_out(x, self.widths, OH, OW)
        self.branch3x3_2.run(self.branch3x3_1.run(x), v[0])
        t = self.branch7x7x3_1.run(x)
        for m in (self.branch7x7x3_2, self.branch7x7x3_3):
            t = m.run(t)
        self.branch7x7x3_4.run(t, v[1])
        pool(x, 3, 2, 0, 'max', v[2])
        return out


class _FIDInceptionE(_Block):                                  # torchvision InceptionE; pool_mode is the FID patch (inception.py:254-328)
    pool_mode = 'avg'

    def __init__(self, in_channels):
        super().__init__()
        self.branch1x1 = BasicConv2d(in_channels, 320, 1)
        self.branch3x3_1 = BasicConv2d(in_channels, 384, 1)
        self.branch3x3_2a = BasicConv2d(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3_2b = BasicConv2d(384, 384, (3, 1), padding=(1, 0))
        self.branch3x3dbl_1 = BasicConv2d(in_channels, 448, 1)
        self.branch3x3dbl_2 = BasicConv2d(448, 384, 3, padding=1)
        self.branch3x3dbl_3a = BasicConv2d(384, 384, (1, 3), padding=(0, 1))
        self.branch3x3dbl_3b = BasicConv2d(384, 384, (3, 1), padding=(1, 0))
        self.branch_pool = BasicConv2d(in_channels, 192, 1)
        self.widths = (320, 384, 384, 384, 384, 192)

    def run(self, x):
        _, H, W, _ = x.shape
        out, v = self._cat_out(x, self.widths, H, W)
        self.branch1x1.run(x, v[0])
        t = self.branch3x3_1.run(x)
        self.branch3x3_2a.run(t, v[1])
        self.branch3x3_2b.run(t, v[2])
        t = self.branch3x3dbl_2.run(self.branch3x3dbl_1.run(x))
        self.branch3x3dbl_3a.run(t, v[3])
        self.branch3x3dbl_3b.run(t, v[4])
        self.branch_pool.run(pool(x, 3, 1, 1, self.pool_mode), v[5])
        return out


class FIDInceptionE_1(_FIDInceptionE):
    pool_mode = 'avg'                                          # count_include_pad=False (inception.py:279-283)


class FIDInceptionE_2(_FIDInceptionE):
    pool_mode = 'max'                                          # "likely an error in this specific Inception implementation" (inception.py:319-323)


# pytorch-fid / torchvision layer name -> (block, index) in the wrapper's `blocks`
_TV_NAMES = (('Conv2d_1a_3x3', 0, 0), ('Conv2d_2a_3x3', 0, 1), ('Conv2d_2b_3x3', 0, 2), ('Conv2d_3b_1x1', 1, 0), ('Conv2d_4a_3x3', 1, 1),
             ('Mixed_5b', 2, 0), ('Mixed_5c', 2, 1), ('Mixed_5d', 2, 2), ('Mixed_6a', 2, 3), ('Mixed_6b', 2, 4), ('Mixed_6c', 2, 5),
             ('Mixed_6d', 2, 6), ('Mixed_6e', 2, 7), ('Mixed_7a', 3, 0), ('Mixed_7b', 3, 1), ('Mixed_7c', 3, 2))


def find_fid_weights(weights=None):
    """Path of the pytorch-fid weight file: the argument, $COLDDIFF_FID_WEIGHTS, or torch hub's cache (where the reference's
    load_state_dict_from_url would have put it)."""
    cands = [weights, os.environ.get("COLDDIFF_FID_WEIGHTS"), os.path.join(torch.hub.get_dir(), "checkpoints", FID_WEIGHTS_FILE)]
    for c in cands:
        if c and os.path.exists(c):
            return c
    raise FileNotFoundError(
        f"InceptionV3: the pretrained FID weights ({FID_WEIGHTS_FILE}) were not found; the reference downloads them from "
        f"{FID_WEIGHTS_URL}.  Put the file at {cands[2]}, or pass weights=<path> / set COLDDIFF_FID_WEIGHTS "
        f"(weights='random' builds an untrained network for tests only)")


class InceptionV3(nn.Module):
    """Pretrained InceptionV3 network returning feature maps (Fid/inception.py:16-165)."""

    DEFAULT_BLOCK_INDEX = 3
    BLOCK_INDEX_BY_DIM = {64: 0, 192: 1, 768: 2, 2048: 3}

    def __init__(self, output_blocks=(DEFAULT_BLOCK_INDEX,), resize_input=True, normalize_input=True, requires_grad=False,
                 use_fid_inception=True, weights=None):
        super().__init__()
        assert use_fid_inception, "only the FID Inception (pt_inception-2015-12-05) is built; torchvision's ImageNet variant is not"
        assert not requires_grad, "the feature extractor is inference-only here (the reference never fine-tunes it)"
        self.resize_input, self.normalize_input = resize_input, normalize_input
        self.output_blocks = sorted(output_blocks)
        self.last_needed_block = max(output_blocks)
        assert self.last_needed_block <= 3, 'Last possible output block index is 3'
        self.blocks = nn.ModuleList()
        self.blocks.append(nn.Sequential(BasicConv2d(3, 32, 3, stride=2), BasicConv2d(32, 32, 3), BasicConv2d(32, 64, 3, padding=1),
                                         MaxPool2d(3, 2)))
        if self.last_needed_block >= 1:
            self.blocks.append(nn.Sequential(BasicConv2d(64, 80, 1), BasicConv2d(80, 192, 3), MaxPool2d(3, 2)))
        if self.last_needed_block >= 2:
            self.blocks.append(nn.Sequential(FIDInceptionA(192, 32), FIDInceptionA(256, 64), FIDInceptionA(288, 64), InceptionB(288),
                                             FIDInceptionC(768, 128), FIDInceptionC(768, 160), FIDInceptionC(768, 160), FIDInceptionC(768, 192)))
        if self.last_needed_block >= 3:
            self.blocks.append(nn.Sequential(InceptionD(768), FIDInceptionE_1(1280), FIDInceptionE_2(2048), AdaptiveAvgPool2d()))
        for p in self.parameters():
            p.requires_grad = False
        if isinstance(weights, dict):
            self.load_fid_state_dict(weights)
        elif weights != 'random':
            self.load_fid_state_dict(torch.load(find_fid_weights(weights), map_location='cpu'))
        self.eval()

    def load_fid_state_dict(self, sd):
        """Load a state_dict in the pytorch-fid / torchvision layout (`Mixed_5b.branch1x1.conv.weight`, ...; `fc.*`, `AuxLogits.*` and
        layers beyond `last_needed_block` are not part of this module and are skipped)."""
        own = self.state_dict()
        mapped, used = {}, set()
        for name, bi, li in _TV_NAMES:
            if bi > self.last_needed_block:
                continue
            for k, v in sd.items():
                if k.startswith(name + '.'):
                    mapped[f'blocks.{bi}.{li}.' + k[len(name) + 1:]] = v
                    used.add(k)
        missing = [k for k in own if k not in mapped and not k.endswith('num_batches_tracked')]
        assert not missing, f"FID weights: {len(missing)} tensors missing, e.g. {missing[:3]}"
        self.load_state_dict(mapped, strict=False)
        return self

    def forward(self, inp):
        """inp [B,3,H,W] in (0, 1) -> list of the selected blocks' feature maps (NCHW), ascending by index."""
        inp = rt.check(inp).float().contiguous()
        B, C, H, W = inp.shape
        assert C == 3, "InceptionV3 takes 3-channel images"
        mul, add = (2.0, -1.0) if self.normalize_input else (1.0, 0.0)
        OH, OW = (299, 299) if self.resize_input else (H, W)
        x = torch.zeros((B, OH, OW, 4), device=inp.device, dtype=torch.float32)          # NHWC, channel pad zero
        # (without a resize the same kernel is an exact copy: scale 1 puts every source index on a pixel with weight 1)
        rt.lib().cdf_resize_bilinear_nhwc(P(inp), P(x), 4, B, 3, H, W, OH, OW, mul, add, rt.stream(inp))
        outp = []
        # The metric must not depend on the training arithmetic: under COLDDIFF_PRECISION=bf16 (single-bf16 GEMM operands) the 94 conv
        # layers would drift from pytorch-fid's fp32 features.  The extractor always runs parity-grade (split precision, or exact fp32
        # when the process is in f32 mode).
        saved = rt.precision
        if saved == "bf16":
            rt.set_precision("bf16x3")
        try:
            for idx, block in enumerate(self.blocks):
                for layer in block:
                    x = layer.run(x)
                if idx in self.output_blocks:
                    outp.append(ops.nhwc_to_nchw(x, x.shape[-1]))
                if idx == self.last_needed_block:
                    break
        finally:
            if saved != rt.precision:
                rt.set_precision(saved)
        return outp


def fid_inception_v3():
    raise NotImplementedError("use InceptionV3(...): the torchvision Inception3 object the reference patches is not reproduced")

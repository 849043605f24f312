"""`Model` — the DDPM ResNet/attention UNet used by the CIFAR-10 scripts
(reference: deblurring_diffusion_pytorch/Model2.py:6-332; whitespace-identical copies in the
resolution and defading packages).  Same constructor keywords, forward signature, sub-module names
and parameter shapes as the reference; the modules are parameter containers and the arithmetic runs
in the HIP kernels through the autograd nodes of `colddiff.functions`.
"""
import math

import torch
import torch.nn as nn

from . import functions as F_
from . import runtime as rt
from .unet import anchor


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = torch.nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x, out=None):
        if not self.with_conv:
            return F_.Upsample2Fn.apply(x, out)             # bare nearest x2 (Model2.py:46-49)
        return F_.UpsampleConvFn.apply(anchor(x), x, self.conv, out)


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = torch.nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def forward(self, x, out=None):
        if not self.with_conv:
            return F_.AvgPool2Fn.apply(x, out)              # F.avg_pool2d(x, 2, 2) (Model2.py:71-72)
        # F.pad(x, (0,1,0,1)) then 3x3 stride 2: the asymmetric zero pad is folded into the gather
        return F_.ConvFn.apply(anchor(x), x, self.conv, x.shape[-1], "conv", 2, (0, 0, 1, 1), out)


def Normalize(in_channels):
    return torch.nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = Normalize(in_channels)
        self.conv1 = torch.nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.temb_proj = torch.nn.Linear(temb_channels, out_channels)
        self.norm2 = Normalize(out_channels)
        self.dropout = torch.nn.Dropout(dropout)
        self.conv2 = torch.nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            if self.use_conv_shortcut:
                self.conv_shortcut = torch.nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
            else:
                self.nin_shortcut = torch.nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, swish_temb, out=None):
        tb = F_.Linear.apply(anchor(x), swish_temb, self.temb_proj)
        return F_.ResnetBlockFn.apply(anchor(x), x, tb, self, out)


class AttnBlock(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.k = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.v = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)
        self.proj_out = torch.nn.Conv2d(in_channels, in_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, out=None):
        return F_.AttnBlockFn.apply(anchor(x), x, self, out)


class Model(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution):
        super().__init__()
        self.ch = ch
        self.temb_ch = self.ch * 4
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.resolution = resolution
        self.in_channels = in_channels
        self.out_ch = out_ch

        self.temb = nn.Module()
        self.temb.dense = nn.ModuleList([torch.nn.Linear(self.ch, self.temb_ch), torch.nn.Linear(self.temb_ch, self.temb_ch)])
        self.conv_in = torch.nn.Conv2d(in_channels, self.ch, kernel_size=3, stride=1, padding=1)

        curr_res = resolution
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        for i_level in range(self.num_resolutions):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for i_block in range(self.num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=self.temb_ch, dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            down = nn.Module()
            down.block, down.attn = block, attn
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
                curr_res = curr_res // 2
            self.down.append(down)

        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)
        self.mid.attn_1 = AttnBlock(block_in)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=self.temb_ch, dropout=dropout)

        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            skip_in = ch * ch_mult[i_level]
            for i_block in range(self.num_res_blocks + 1):
                if i_block == self.num_res_blocks:
                    skip_in = ch * in_ch_mult[i_level]
                block.append(ResnetBlock(in_channels=block_in + skip_in, out_channels=block_out, temb_channels=self.temb_ch,
                                         dropout=dropout))
                block_in = block_out
                if curr_res in attn_resolutions:
                    attn.append(AttnBlock(block_in))
            up = nn.Module()
            up.block, up.attn = block, attn
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
                curr_res = curr_res * 2
            self.up.insert(0, up)

        self.norm_out = Normalize(block_in)
        self.conv_out = torch.nn.Conv2d(block_in, out_ch, kernel_size=3, stride=1, padding=1)

        # torch.cat([h, hs.pop()], dim=1) (Model2.py:310-311) without the copy: the buffer of every concatenation is allocated when its
        # SKIP half is produced on the way down (functions.CatBuf), and whoever produces `h` on the way up writes the first half.
        # _skip_cx[j] = channels of the `h` that meets the j-th pushed skip tensor = its consumer's in_channels - the skip's channels.
        skip_ch = [ch]
        for i_level in range(self.num_resolutions):
            skip_ch += [ch * ch_mult[i_level]] * (self.num_res_blocks + (1 if i_level != self.num_resolutions - 1 else 0))
        consumers = [self.up[i_level].block[i_block].in_channels for i_level in reversed(range(self.num_resolutions))
                     for i_block in range(self.num_res_blocks + 1)]
        assert len(consumers) == len(skip_ch)
        self._skip_cx = [consumers[len(skip_ch) - 1 - j] - c for j, c in enumerate(skip_ch)]

        half = self.ch // 2
        freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))   # Model2.py:16-18
        self.register_buffer("_freq", freq, persistent=False)

    def forward(self, x, t):
        # no-grad calls are the samplers' (iterated) ones: exact-fp32 GEMMs there (runtime.MODEL_SAMPLE_PRECISION has the why)
        if rt.precision == "bf16x3" and rt.MODEL_SAMPLE_PRECISION == "f32" and not torch.is_grad_enabled():
            with rt.precision_scope("f32"):
                return self._forward(x, t)
        return self._forward(x, t)

    def _forward(self, x, t):
        rt.check(x)
        assert x.shape[2] == x.shape[3] == self.resolution
        assert self.ch % 2 == 0, "odd embedding widths (zero pad, Model2.py:22) are not used by the scripts"
        a = anchor(x)
        t = F_.batch_time(t, x.shape[0])
        temb = F_.Sinusoidal.apply(t.contiguous(), self._freq, self.ch)
        temb = F_.Linear.apply(a, temb, self.temb.dense[0])
        temb = F_.Act.apply(temb, F_.ACT_SILU)
        temb = F_.Linear.apply(a, temb, self.temb.dense[1])
        st = F_.Act.apply(temb, F_.ACT_SILU)        # nonlinearity(temb), shared by every ResnetBlock

        h = F_.ToNHWC.apply(x.float())
        B, R = h.shape[0], self.resolution
        hs, cats = [], []

        def skip_dst(res, c):                               # the concat buffer of the next skip tensor; returns its second half
            cats.append(F_.CatBuf(h, B, res, res, self._skip_cx[len(cats)], c))
            return cats[-1].second()

        hs.append(F_.ConvFn.apply(a, h, self.conv_in, self.in_channels, "conv", 1, (1, 1, 1, 1), skip_dst(R, self.ch)))
        res = R
        for i_level in range(self.num_resolutions):
            lvl = self.down[i_level]
            for i_block in range(self.num_res_blocks):
                c = lvl.block[i_block].out_channels
                if len(lvl.attn) > 0:
                    h = lvl.attn[i_block](lvl.block[i_block](hs[-1], st), skip_dst(res, c))
                else:
                    h = lvl.block[i_block](hs[-1], st, skip_dst(res, c))
                hs.append(h)
            if i_level != self.num_resolutions - 1:
                res //= 2
                hs.append(lvl.downsample(hs[-1], skip_dst(res, hs[-1].shape[-1])))

        h = hs[-1]
        h = self.mid.block_1(h, st)
        h = self.mid.attn_1(h)
        h = self.mid.block_2(h, st, cats[-1].first())

        for i_level in reversed(range(self.num_resolutions)):
            lvl = self.up[i_level]
            for i_block in range(self.num_res_blocks + 1):
                hcat = F_.Join.apply(h, hs.pop(), cats.pop())
                # where this block's result goes: the first half of the next concatenation -- directly, or through the upsampler
                nxt = cats[-1].first() if cats else None
                via_up = i_block == self.num_res_blocks and i_level != 0
                dst = None if via_up else nxt
                if len(lvl.attn) > 0:
                    h = lvl.attn[i_block](lvl.block[i_block](hcat, st), dst)
                else:
                    h = lvl.block[i_block](hcat, st, dst)
            if i_level != 0:
                h = lvl.upsample(h, cats[-1].first())

        h = F_.GroupNormFn.apply(a, h, self.norm_out, True)
        h = F_.ConvFn.apply(a, h, self.conv_out, h.shape[-1], "conv", 1, (1, 1, 1, 1))
        return F_.ToNCHW.apply(h, self.out_ch, None)

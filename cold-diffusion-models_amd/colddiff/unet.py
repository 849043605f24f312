"""`Unet` — the ConvNeXt / linear-attention denoiser shared by the four cold-diffusion packages
(reference: deblurring_diffusion_pytorch.py:83-282; identical copies in the denoising, resolution
and defading packages).

Same constructor, forward signature, sub-module names and parameter shapes as the reference (so
`state_dict`s and the authors' checkpoints load unchanged, and the same seed gives the same initial
weights), but the modules below are parameter containers only: the arithmetic runs in the HIP
kernels of libcolddiff_hip.so on NHWC activations, through the block-level autograd nodes of
`colddiff.functions`.
"""
import math

import torch
from torch import nn

from . import bf16store as BFS
from . import functions as F_
from . import ops
from . import runtime as rt


def exists(x):
    return x is not None


def default(val, d):
    return val if exists(val) else (d() if callable(d) else d)


class _Anchor(nn.Module):
    """Gives every kernel-backed module a dummy tensor that requires grad (see functions.py)."""
    _anchors = {}

    @staticmethod
    def get(device):
        key = str(device)
        a = _Anchor._anchors.get(key)
        if a is None:
            a = torch.zeros((), device=device, requires_grad=True)
            _Anchor._anchors[key] = a
        return a


_CONCAT_FREE = __import__("os").environ.get("CDF_CONCAT_FREE", "1") != "0"
_TIME_BIAS_ALL = __import__("os").environ.get("CDF_TIME_BIAS_ALL", "1") != "0"     # one launch for every block's time-bias Linear


def anchor(t):
    return _Anchor.get(t.device)


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def forward(self, x, dest=None):    # only Residual(PreNorm(LinearAttention)) occurs in the model
        if ops.is_bf(x):                # bf16 activation storage (colddiff/bf16store.py): the stream's type selects the node
            return BFS.LinAttnBlockBF.apply(anchor(x), x, self, dest)
        return F_.LinAttnBlockFn.apply(anchor(x), x, self, dest)


class SinusoidalPosEmb(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        half = dim // 2
        # init-time constant, same expression as the reference (DEBLUR:98-100)
        freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
        self.register_buffer("_freq", freq, persistent=False)

    def forward(self, t):
        return F_.Sinusoidal.apply(t.contiguous(), self._freq, self.dim)


class LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))
        self.b = nn.Parameter(torch.zeros(1, dim, 1, 1))


class PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = LayerNorm(dim)


class GELU(nn.Module):
    def forward(self, x):
        return F_.Act.apply(x, F_.ACT_GELU)


class ConvNextBlock(nn.Module):
    def __init__(self, dim, dim_out, *, time_emb_dim=None, mult=2, norm=True):
        super().__init__()
        self.dim, self.dim_out = dim, dim_out
        self.mlp = nn.Sequential(GELU(), nn.Linear(time_emb_dim, dim)) if exists(time_emb_dim) else None
        self.ds_conv = nn.Conv2d(dim, dim, 7, padding=3, groups=dim)
        self.net = nn.Sequential(
            LayerNorm(dim) if norm else nn.Identity(),
            nn.Conv2d(dim, dim_out * mult, 3, padding=1),
            GELU(),
            nn.Conv2d(dim_out * mult, dim_out, 3, padding=1),
        )
        self.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else nn.Identity()
        self.has_norm = norm
        self.has_res_conv = dim != dim_out

    def forward(self, x, gelu_t=None, dest=None, tb=None, enter_bf16=False):
        """x: NHWC feature map; gelu_t: GELU(time embedding) [B, time_dim] (shared by all blocks); dest: CatBuf whose first half
        receives the output; tb: this block's time bias when the caller computed all of them in one launch (F_.TimeBiasAll);
        enter_bf16: x is the fp32 output of the image-side block entering the bf16 stream in this block (bf16store.py)."""
        if tb is None and exists(self.mlp):
            assert exists(gelu_t), "time emb must be passed in"
            tb = F_.Linear.apply(anchor(x), gelu_t, self.mlp[1])
        if ops.is_bf(x) or enter_bf16:
            return BFS.ConvNextBlockBF.apply(anchor(x), x, tb, self, dest)
        return F_.ConvNextBlockFn.apply(anchor(x), x, tb, self, dest)


class LinearAttention(nn.Module):
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        assert dim_head == 32, "the HIP linear-attention kernels are specialised for dim_head = 32"
        self.scale = dim_head ** -0.5
        self.heads = heads
        hidden_dim = dim_head * heads
        self.to_qkv = nn.Conv2d(dim, hidden_dim * 3, 1, bias=False)
        self.to_out = nn.Conv2d(hidden_dim, dim, 1)


def Upsample(dim):
    return nn.ConvTranspose2d(dim, dim, 4, 2, 1)


def Downsample(dim):
    return nn.Conv2d(dim, dim, 4, 2, 1)


class Unet(nn.Module):
    def __init__(self, dim, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3, with_time_emb=True, residual=False):
        super().__init__()
        self.channels = channels
        self.residual = residual
        print("Is Time embed used ? ", with_time_emb)

        dims = [channels, *map(lambda m: dim * m, dim_mults)]
        in_out = list(zip(dims[:-1], dims[1:]))

        if with_time_emb:
            time_dim = dim
            self.time_mlp = nn.Sequential(SinusoidalPosEmb(dim), nn.Linear(dim, dim * 4), GELU(), nn.Linear(dim * 4, dim))
        else:
            time_dim = None
            self.time_mlp = None

        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])
        num_resolutions = len(in_out)

        for ind, (dim_in, dim_out) in enumerate(in_out):
            is_last = ind >= (num_resolutions - 1)
            self.downs.append(nn.ModuleList([
                ConvNextBlock(dim_in, dim_out, time_emb_dim=time_dim, norm=ind != 0),
                ConvNextBlock(dim_out, dim_out, time_emb_dim=time_dim),
                Residual(PreNorm(dim_out, LinearAttention(dim_out))),
                Downsample(dim_out) if not is_last else nn.Identity(),
            ]))

        mid_dim = dims[-1]
        self.mid_block1 = ConvNextBlock(mid_dim, mid_dim, time_emb_dim=time_dim)
        self.mid_attn = Residual(PreNorm(mid_dim, LinearAttention(mid_dim)))
        self.mid_block2 = ConvNextBlock(mid_dim, mid_dim, time_emb_dim=time_dim)

        for ind, (dim_in, dim_out) in enumerate(reversed(in_out[1:])):
            is_last = ind >= (num_resolutions - 1)
            self.ups.append(nn.ModuleList([
                ConvNextBlock(dim_out * 2, dim_in, time_emb_dim=time_dim),
                ConvNextBlock(dim_in, dim_in, time_emb_dim=time_dim),
                Residual(PreNorm(dim_in, LinearAttention(dim_in))),
                Upsample(dim_in) if not is_last else nn.Identity(),
            ]))

        out_dim = default(out_dim, channels)
        self.out_dim = out_dim
        self.final_conv = nn.Sequential(ConvNextBlock(dim, dim), nn.Conv2d(dim, out_dim, 1))

    def _bf16_ok(self):
        """Every block past the image-side one has channel counts a bf16 feature map can carry (multiples of 8), and the network has the
        shape the bf16 stream is wired for (at least one up stage: the image-side level's skip is never consumed)."""
        ok = self.__dict__.get("_bf16_ok_cached")
        if ok is None:
            blocks = [b for stage in self.downs for b in stage[:2]][1:] + [self.mid_block1, self.mid_block2] + \
                     [b for stage in self.ups for b in stage[:2]] + [self.final_conv[0]]
            ok = self.__dict__["_bf16_ok_cached"] = all(BFS.block_ok(b) for b in blocks) and len(self.downs) >= 2
        return ok

    # -- time embedding: SinusoidalPosEmb -> Linear -> GELU -> Linear, then GELU once for all blocks --
    def _time(self, time, ref):
        if not exists(self.time_mlp):
            return None
        a = anchor(ref)
        e = self.time_mlp[0](time)
        h = F_.Linear.apply(a, e, self.time_mlp[1])
        h = F_.Act.apply(h, F_.ACT_GELU)
        t = F_.Linear.apply(a, h, self.time_mlp[3])
        return F_.Act.apply(t, F_.ACT_GELU)

    def forward(self, x, time):
        rt.check(x)
        assert x.shape[2] % 8 == 0 and x.shape[3] % 8 == 0, "Unet needs H, W divisible by 8"
        orig_x = x
        a = anchor(x)
        time = F_.batch_time(time, x.shape[0])
        gt = self._time(time, x)
        x = F_.ToNHWC.apply(x.float())
        # Skip connections without copies (DEBLUR:266, 274): the skip tensor of every level an up stage consumes is produced INSIDE its
        # concat buffer (second half), and the producer of the tensor it is concatenated with writes the first half (F_.CatBuf).
        nskip = len(self.ups)                                  # (the first level's skip is appended upstream but never consumed)
        # every block's time bias in one launch (the blocks share GELU(t_emb)); tbs: block -> its [B, r4(dim)] column slice
        tbs = {}
        if _TIME_BIAS_ALL and gt is not None:
            if "_tb_blocks" not in self.__dict__:
                blocks = [b for stage in self.downs for b in stage[:2]] + [self.mid_block1, self.mid_block2] + \
                         [b for stage in self.ups for b in stage[:2]]
                self.__dict__["_tb_blocks"] = [b for b in blocks if exists(b.mlp)]
                self.__dict__["_tb_lins"] = [b.mlp[1] for b in self._tb_blocks]
            tbs = dict(zip(map(id, self._tb_blocks), F_.TimeBiasAll.apply(a, gt, self)))
        T = lambda blk: tbs.get(id(blk))
        # bf16 activation storage ("bf16" arithmetic mode): the stream between blocks, every saved activation and every GEMM input is ONE
        # bf16 plane from the output of the image-side block to the input of the 3-channel output conv (colddiff/bf16store.py)
        bfs = BFS.enabled() and _CONCAT_FREE and self._bf16_ok()
        conv = (lambda *a_: BFS.ConvFnBF.apply(*a_)) if bfs else (lambda *a_: F_.ConvFn.apply(*a_))
        h = []
        for lvl, (convnext, convnext2, attn, downsample) in enumerate(self.downs):
            x = convnext(x, gt, tb=T(convnext))
            if bfs and lvl == 0:
                # the image-side block's fp32 output enters the stream inside the next block (rounded there; its gradient comes back fp32)
                x = convnext2(x, gt, tb=T(convnext2), enter_bf16=True)
            else:
                x = convnext2(x, gt, tb=T(convnext2))
            cat = None
            if _CONCAT_FREE and lvl >= len(self.downs) - nskip:
                B_, H_, W_, C_ = x.shape
                cat = F_.CatBuf(x, B_, H_, W_, C_, C_, dtype=x.dtype)
            x = attn(x, cat)
            h.append((x, cat))
            if not isinstance(downsample, nn.Identity):
                x = conv(a, x, downsample, x.shape[-1], "conv", 2, (1, 1, 1, 1))

        x = self.mid_block1(x, gt, tb=T(self.mid_block1))
        x = self.mid_attn(x)
        x = self.mid_block2(x, gt, h[-1][1], tb=T(self.mid_block2))     # (its output is the first half of the deepest concat)

        for j, (convnext, convnext2, attn, upsample) in enumerate(self.ups):
            skip, cat = h.pop()
            if cat is not None:
                x = (BFS.JoinBF if bfs else F_.Join).apply(x, skip, cat)
            else:
                x = F_.Concat.apply(x, skip)
            x = convnext(x, gt, tb=T(convnext))
            x = convnext2(x, gt, tb=T(convnext2))
            x = attn(x)
            if not isinstance(upsample, nn.Identity):
                nxt = h[-1][1] if j + 1 < len(self.ups) else None      # the next stage's concat buffer takes the upsampled map
                x = conv(a, x, upsample, x.shape[-1], "convT", 2, (1, 1, 1, 1), nxt)

        x = self.final_conv[0](x)
        if bfs:
            x = BFS.ToF32.apply(x)
        x = F_.ConvFn.apply(a, x, self.final_conv[1], x.shape[-1], "conv", 1, (0, 0, 0, 0))
        return F_.ToNCHW.apply(x, self.out_dim, orig_x.float().contiguous() if self.residual else None)

"""Block-level autograd nodes of the UNets.

Each node runs a fused chain of HIP kernels in forward and a hand-written backward; parameter
gradients are accumulated by the kernels straight into ``param.grad`` (one flat arena, see
``colddiff.flat``), so autograd only carries activation gradients between blocks.  Every node takes
an ``anchor`` (a dummy scalar that requires grad) first, so its backward also runs when the
activation input itself does not require grad (the first block sees the image).
"""
import os

import torch

from . import convdesc as cd
from . import ops
from . import parallel
from . import runtime as rt
from .runtime import P, r4


def _used(ctx, *mods):
    """Forward side of _done: tell the data-parallel engine that this node's backward will write the gradients of these
    modules' own parameters once more in this pass (a module may run several times per loss, e.g. RESOL's
    'Final_random_mean_and_actual' calls the network twice: RESOL:702-716), and remember them on the node."""
    ctx.owned = tuple(m for m in mods if m is not None)
    if parallel._engine is not None and ctx.needs_input_grad[0]:      # anchor: True iff autograd is recording
        for m in ctx.owned:
            parallel.grads_used(list(m.parameters(recurse=False)))


def _done(ctx):
    """Tell the data-parallel engine that this node's contribution to the gradients of its modules' own parameters is enqueued."""
    if parallel._engine is not None:
        for m in ctx.owned:
            parallel.grads_ready(list(m.parameters(recurse=False)))

ACT_NONE, ACT_GELU, ACT_SILU = 0, 1, 2


def _conv_plans(kind, H, W, k, stride, pad):
    """(fwd plan, dgrad plan, wgrad plan, packed-kind prefix, (s_r, s_c) factory) for a conv module."""
    if kind == "conv":
        pt, pl, pb, pr = pad
        return (cd.conv_fwd(H, W, k, k, stride, pt, pl, pb, pr), cd.conv_dgrad(H, W, k, k, stride, pt, pl, pb, pr),
                cd.conv_wgrad(H, W, k, k, stride, pt, pl, pb, pr))
    return cd.convT_fwd(H, W, k, k, stride, pad[0]), cd.convT_dgrad(H, W, k, k, stride, pad[0]), cd.convT_wgrad(H, W, k, k, stride, pad[0])


def conv_forward(x, Cin, weight, bias, kind="conv", stride=1, pad=None, xs=None, split_out=False, planes_only=False, **epi):
    """x [B,H,W,*] -> conv(x) with the weight in its PyTorch layout ([Cout,Cin,k,k] or [Cin,Cout,k,k]).
    xs: optional pre-split (hi, lo) bf16 planes of x (see want_presplit).
    split_out: return (y, (hi, lo) planes of y) -- fused into the epilogue on the pre-split path."""
    k = weight.shape[-1]
    if pad is None:
        pad = (k // 2,) * 4
    _, H, W, _ = x.shape
    plan = _conv_plans(kind, H, W, k, stride, pad)[0]
    Cout = weight.shape[0] if kind == "conv" else weight.shape[1]
    sfx = _sp_suffix(Cin * k * k, Cout)
    wp = ops.packed(weight, ("conv_fwd" if kind == "conv" else "convT_fwd") + sfx)
    if xs is None and sfx and _ALWAYS_PRESPLIT and rt.precision != "f32" and Cin % 8 == 0 and x.device.type != "meta":
        xs = ops.split_bf16(x[..., :Cin] if x.shape[-1] != Cin else x)     # every bf16x3 GEMM goes through the LDS-DMA kernel
    if xs is not None and sfx:
        return ops.conv_gemm_presplit(plan, xs, Cin, wp, Cout, bias=bias, split_out=split_out, planes_only=planes_only, **epi)
    y = ops.conv_gemm(plan, x, Cin, wp, Cout, bias=bias, **epi)
    return (y, ops.split_bf16(y)) if split_out else y


def want_presplit(Cin, Cout, k):
    """True when a conv's operands should be handed over as bf16 planes produced once up front (bf16x3 / bf16 modes, GEMM routed to the
    split-precision kernels, channel counts vector-friendly)."""
    return rt.precision != "f32" and Cin % 8 == 0 and Cout % 8 == 0 and bool(_sp_suffix(Cin * k * k, Cout)) and bool(_sp_suffix(Cout * k * k, Cin))


_CIN4 = os.environ.get("CDF_CIN4", "1") != "0"    # direct kernels for the <= 4-input-channel image-side convs
_CIN_DGRAD2 = os.environ.get("CDF_CIN_DGRAD2", "1") != "0"   # their 3x3 data gradient in two stages (ops.conv_cin_dgrad2)
_ATTN_FUSED = int(os.environ.get("CDF_ATTN_FUSED", "1"))    # to_out folded into the linear-attention product (ops.linattn_project): 0 never, 1 where dim <= heads*32, 2 always
_LEAN = os.environ.get("CDF_LEAN", "1") != "0"    # skip fp32 copies of tensors only ever consumed as bf16 planes
_LINEAR_SMALL_M = 256      # batch sizes up to this use the skinny-linear kernels
_ALWAYS_PRESPLIT = os.environ.get("CDF_ALWAYS_PRESPLIT", "0") != "0"
_PRESPLIT_1X1 = os.environ.get("CDF_PRESPLIT_1X1", "0") != "0"
_AUTO_PRESPLIT = os.environ.get("CDF_AUTO_PRESPLIT", "1") != "0"
_SP_KMIN = int(os.environ.get("CDF_SP_KMIN", "64"))     # tuning knob: smallest K routed to the bf16 matrix cores
_LN_FUSE = os.environ.get("CDF_LN_FUSE", "1") != "0"  # conv1's data gradient runs the LayerNorm backward in its epilogue (ops.conv_dgrad_lnbwd)
_PRE_GRAD = os.environ.get("CDF_PRE_GRAD", "0") != "0"  # conv1's epilogue stores GELU'(pre) (same erf / exp evaluation as GELU); conv2's data gradient multiplies by it


def _sp_suffix(K, N):
    """Route a dense conv to the split-precision bf16 MFMA kernel when enabled and the GEMM is deep and
    wide enough (K = taps*Cin >= 64, N >= 64); the tiny first/last layers, K = 32 attention products and
    the time-embedding linears stay on the exact-fp32 kernel."""
    return "_sp" if (rt.precision != "f32" and K >= _SP_KMIN and N >= 64) else ""


def conv_backward(x, Cin, dy, weight, bias, kind="conv", stride=1, pad=None, need_dx=True, dx=None, dx_accumulate=0, mul=None,
                  mul_mode=0, xs=None, dys=None, split_dx=False, planes_only=False, ln=None):
    """Gradients of conv_forward: returns dx (optionally fused with an activation-gradient multiply),
    accumulates into weight.grad / bias.grad.
    ln = (h, norm, mean, rstd): x was LayerNorm(h) -- returns (d, fused): fused True => d is already dh, the LayerNorm backward ran in the
    data-gradient GEMM's epilogue and norm.g / norm.b have their gradients; False => d is dx as usual (the caller runs ops.layernorm_bwd)."""
    k = weight.shape[-1]
    if pad is None:
        pad = (k // 2,) * 4
    _, H, W, _ = x.shape
    _, pd, pw = _conv_plans(kind, H, W, k, stride, pad)
    KK = k * k
    if kind == "conv":
        Cout = weight.shape[0]
        s_r, s_c = KK, Cin * KK
    else:
        Cout = weight.shape[1]
        s_r, s_c = Cout * KK, KK
    fuse_bias = bias is not None and kind == "conv"        # dY rows stream through the wgrad kernel exactly once
    ops.wgrad_into(ops.grad_of(weight), pw, x, Cin, dy, Cout, 1, s_r, s_c, gbias=ops.grad_of(bias) if fuse_bias else None,
                   xa_s=xs, xb_s=dys)
    if bias is not None and not fuse_bias:
        ops.colsum_into(ops.grad_of(bias), dy, Cout)
    if not need_dx:
        return None
    sfx = _sp_suffix(Cout * KK, Cin)
    wd = ops.packed(weight, ("conv_dgrad" if kind == "conv" else "convT_dgrad") + sfx)
    if dys is None and sfx and _ALWAYS_PRESPLIT and rt.precision != "f32" and Cout % 8 == 0 and dy.device.type != "meta":
        dys = ops.split_bf16(dy[..., :Cout] if dy.shape[-1] != Cout else dy)
    if dys is not None and sfx:
        if (ln is not None and _LN_FUSE and kind == "conv" and dx is None and mul is None and not split_dx and dys[1] is not None
                and ops.conv_dgrad_lnbwd_ok(pd, x.shape[0], Cout, Cin)):
            h, norm, mean, rstd = ln
            return ops.conv_dgrad_lnbwd(pd, dys, Cout, wd, Cin, h, norm.g, norm.b, mean, rstd), True
        g = ops.conv_gemm_presplit(pd, dys, Cout, wd, Cin, y=dx, mul=mul, mul_mode=mul_mode, accumulate=dx_accumulate,
                                   split_out=split_dx, planes_only=planes_only)
        return (g, False) if ln is not None else g
    g = ops.conv_gemm(pd, dy, Cout, wd, Cin, y=dx, mul=mul, mul_mode=mul_mode, accumulate=dx_accumulate)
    if ln is not None:
        return g, False
    return (g, ops.split_bf16(g)) if split_dx else g


def batch_time(time, B):
    """The per-sample tables of this engine (time bias, per-sample GEMM bias) are indexed by the image row, where the reference's
    `h + rearrange(condition, 'b c -> b c 1 1')` (DEBLUR:160) BROADCASTS: a [1] time against a larger image batch is legal upstream (the GMM
    scripts call `all_sample(1, imgs)`, DENOISE:1203) -- expanded here; any other mismatch fails as torch's broadcast would."""
    if time is None or time.dim() == 0 or time.shape[0] == B:
        return time
    if time.shape[0] != 1:
        raise RuntimeError(f"The size of tensor a ({B}) must match the size of tensor b ({time.shape[0]}) at non-singleton dimension 0")
    return time.expand(B, *time.shape[1:]).contiguous()


class ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.C = x.shape[1]
        return ops.nchw_to_nhwc(rt.check(x.contiguous()))

    @staticmethod
    def backward(ctx, dy):
        return ops.nhwc_to_nchw(dy, ctx.C)


class ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, C, add):
        ctx.has_add = add is not None
        return ops.nhwc_to_nchw(x, C, add)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        return ops.nchw_to_nhwc(dy), None, (dy if ctx.has_add else None)


class CatBuf:
    """The buffer of a skip concatenation torch.cat((x, h), dim=1) (DEBLUR:274), allocated when the SKIP half is produced on the down
    path: the attention block writes h straight into channels [Cx, Cx + Ch) and, on the up path, the producer of x (mid_block2 or the
    transposed-conv upsampler) writes into channels [0, Cx) -- every kernel takes a pixel pitch, so neither half is ever copied
    (the round-1 `Concat` node spent two cdf_axpby passes per stage on it).  Not a tensor: autograd does not look inside."""

    def __init__(self, ref, B, H, W, Cx, Ch, dtype=torch.float32):
        self.Cx, self.Ch = Cx, Ch
        self.buf = torch.empty((B, H, W, Cx + Ch), device=ref.device, dtype=dtype)

    def first(self):
        return self.buf[..., :self.Cx]

    def second(self):
        return self.buf[..., self.Cx:]


class Join(torch.autograd.Function):
    """The two halves already sit side by side in one CatBuf: hand out the whole buffer; the backward hands out channel-slice views."""

    @staticmethod
    def forward(ctx, a, b, cat):
        assert a.data_ptr() == cat.buf.data_ptr() and b.data_ptr() == cat.buf.data_ptr() + 4 * cat.Cx and a.shape[-1] == cat.Cx and \
            b.shape[-1] == cat.Ch and ops.ld_of(a) == cat.Cx + cat.Ch and ops.ld_of(b) == cat.Cx + cat.Ch, "Join: the halves are not the two slices of this CatBuf"
        ctx.Ca = cat.Cx
        out = cat.buf.view(cat.buf.shape)               # (a fresh tensor object over the same storage)
        out._cdf_grad_f32 = True                        # (its gradient is handed on as channel-slice views: planes of the whole would be wasted)
        return out

    @staticmethod
    def backward(ctx, d):
        return d[..., :ctx.Ca], d[..., ctx.Ca:], None


class Concat(torch.autograd.Function):
    """torch.cat((a, b), dim=channel) on NHWC maps; the backward hands out channel-slice views."""

    @staticmethod
    def forward(ctx, a, b):
        B, H, W, Ca = a.shape
        Cb = b.shape[-1]
        ctx.Ca = Ca
        out = torch.empty((B, H, W, Ca + Cb), device=a.device, dtype=torch.float32)
        L, S = rt.lib(), rt.stream(a)
        rows = B * H * W
        L.cdf_axpby(P(out), Ca + Cb, P(a), ops.ld_of(a), rows, Ca, 0.0, 1.0, S)
        L.cdf_axpby(P(out) + 4 * Ca, Ca + Cb, P(b), ops.ld_of(b), rows, Cb, 0.0, 1.0, S)
        return out

    @staticmethod
    def backward(ctx, d):
        return d[..., :ctx.Ca], d[..., ctx.Ca:]


class Sinusoidal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, freq, dim):
        if t.dtype != torch.int64:
            # the kernel reads int64 steps (every caller upstream passes torch.long: DEBLUR:393, 545); other integer types and
            # integer-valued floats mean the same thing -- anything else would be silently truncated, so it is refused
            if t.is_floating_point() and not bool((t == t.round()).all()):
                raise TypeError("time steps must be integers (the embedding kernel takes int64 steps)")
            t = t.to(torch.int64)
        t = t.contiguous()
        out = torch.empty((t.shape[0], dim), device=freq.device, dtype=torch.float32)
        rt.lib().cdf_sinusoidal(P(t), P(freq), P(out), dim, t.shape[0], dim, rt.stream(out))
        return out

    @staticmethod
    def backward(ctx, d):
        return None, None, None


class Act(torch.autograd.Function):
    """GELU / SiLU on a [B, K] vector (time-embedding MLPs)."""

    @staticmethod
    def forward(ctx, x, act):
        ctx.act = act
        ctx.save_for_backward(x)
        y = torch.empty_like(x)
        rt.lib().cdf_act_fwd(P(x), x.stride(0), P(y), y.stride(0), x.shape[0], x.shape[1], act, rt.stream(x))
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        rt.lib().cdf_act_bwd(P(x), x.stride(0), P(dy), dy.stride(0), P(dx), dx.stride(0), x.shape[0], x.shape[1], ctx.act, 0, rt.stream(x))
        return dx, None


class Linear(torch.autograd.Function):
    """y = x @ W^T + b on [B, K] -> [B, r4(N)] (padded so the result can serve as a per-sample bias)."""

    @staticmethod
    def forward(ctx, anchor, x, lin):
        W, b = lin.weight, lin.bias
        N, K = W.shape
        B = x.shape[0]
        ctx.lin = lin
        _used(ctx, lin)
        ctx.save_for_backward(x)
        if B <= _LINEAR_SMALL_M:                             # batch-rows-only GEMM: dedicated kernel (see k_misc.hip)
            wp = ops.packed(W, "lin_fwd")                    # [1][K][r4(N)]
            y = torch.empty((B, r4(N)), device=x.device, dtype=torch.float32)
            rt.lib().cdf_linear_small(P(x), x.stride(0), P(wp), wp.shape[-1], P(b), P(y), r4(N), B, K, N, rt.stream(x))
            return y
        plan = cd.conv_fwd(1, 1, 1, 1, 1, 0, 0, 0, 0)
        x4 = x.view(B, 1, 1, x.shape[1])
        y = ops.conv_gemm(plan, x4, K, ops.packed(W, "lin_fwd"), N, bias=b)
        return y.view(B, r4(N))

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        lin = ctx.lin
        W, b = lin.weight, lin.bias
        N, K = W.shape
        B = x.shape[0]
        dy = dy.contiguous()
        if B <= _LINEAR_SMALL_M:
            L, S = rt.lib(), rt.stream(x)
            L.cdf_linear_small_wgrad(P(dy), dy.stride(0), P(x), x.stride(0), P(ops.grad_of(W)), P(ops.grad_of(b)) if b is not None else 0,
                                     B, N, K, S)
            dx = None
            if ctx.needs_input_grad[1]:
                dx = torch.empty((B, r4(K)), device=x.device, dtype=torch.float32)
                L.cdf_linear_small(P(dy), dy.stride(0), P(W), K, 0, P(dx), r4(K), B, N, K, S)
                dx = dx[:, :x.shape[1]]
            _done(ctx)
            return None, dx, None
        dy4, x4 = dy.view(B, 1, 1, dy.shape[1]), x.view(B, 1, 1, x.shape[1])
        wp = cd.conv_wgrad(1, 1, 1, 1, 1, 0, 0, 0, 0)
        ops.wgrad_into(ops.grad_of(W), wp, dy4, N, x4, K, 0, K, 1)          # dW[n][k] += dy[b][n] x[b][k]
        if b is not None:
            ops.colsum_into(ops.grad_of(b), dy4, N)
        dx = None
        if ctx.needs_input_grad[1]:
            plan = cd.conv_fwd(1, 1, 1, 1, 1, 0, 0, 0, 0)
            dx = ops.conv_gemm(plan, dy4, N, W.detach().view(1, N, K), K).view(B, r4(K))[:, :x.shape[1]]
        _done(ctx)
        return None, dx, None


class TimeBiasAll(torch.autograd.Function):
    """Every ConvNextBlock's time bias `mlp(t) = Linear(time_dim, dim_i)(GELU(t_emb))` (DEBLUR:142-144, 160) in ONE launch: the blocks
    share their input, so their Linear layers are one GEMM against the row-concatenation of the weights ([sum r4(dim_i)][time_dim],
    refreshed once per weights epoch by a foreach-copy + one transpose pack).  Returns one column-slice view per block.  Backward: the
    blocks write their bias gradients straight into the matching slices of one buffer (`GradSlot`), the data gradient is ONE launch
    against the concatenated weight, the weight / bias gradients one skinny launch per layer (they land in separate parameters).
    16 launches -> 1 per forward (the 200-step sampler runs 3200 of them per image batch), 32 -> 17 per backward."""

    @staticmethod
    def forward(ctx, anchor, gt, owner):
        lins = owner._tb_lins
        B, K = gt.shape[0], lins[0].weight.shape[1]
        key = (rt.weights_epoch, str(gt.device), tuple((l.weight.data_ptr(), l.weight._version, l.bias._version) for l in lins))
        cache = owner.__dict__.get("_tb_cache")
        if cache is None or cache[0] != key:
            offs, o = [], 0
            for l in lins:
                offs.append(o)
                o += r4(l.weight.shape[0])
            if cache is None or cache[1].device != gt.device or cache[1].shape[0] != o:
                wcat = torch.zeros((o, K), device=gt.device, dtype=torch.float32)
                bcat = torch.zeros((o,), device=gt.device, dtype=torch.float32)
                wfwd = torch.empty((1, K, o), device=gt.device, dtype=torch.float32)
            else:
                wcat, bcat, wfwd = cache[1], cache[2], cache[3]
            with torch.no_grad():
                torch._foreach_copy_([wcat[a:a + l.weight.shape[0]] for a, l in zip(offs, lins)], [l.weight for l in lins])
                torch._foreach_copy_([bcat[a:a + l.bias.shape[0]] for a, l in zip(offs, lins)], [l.bias for l in lins])
            rt.lib().cdf_pack_weight(P(wcat), P(wfwd), 1, K, o, o, 0, 1, K, rt.stream(gt))       # [K][J]: wfwd[k][j] = wcat[j][k]
            cache = owner.__dict__["_tb_cache"] = (key, wcat, bcat, wfwd, offs, o)
        _, wcat, bcat, wfwd, offs, J = cache
        y = torch.empty((B, J), device=gt.device, dtype=torch.float32)
        rt.lib().cdf_linear_small(P(gt), gt.stride(0), P(wfwd), J, P(bcat), P(y), J, B, K, J, rt.stream(gt))
        ctx.owner, ctx.offs, ctx.J = owner, offs, J
        ctx.slot = GradSlot(B, J, gt.device)
        _used(ctx, *lins)
        ctx.save_for_backward(gt)
        outs = []
        for a, l in zip(offs, lins):
            v = y[:, a:a + r4(l.weight.shape[0])]
            v._cdf_gslot = (ctx.slot, a)                      # where this block's backward puts d(time bias)
            outs.append(v)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        (gt,) = ctx.saved_tensors
        lins, offs, J = ctx.owner._tb_lins, ctx.offs, ctx.J
        B, K = gt.shape
        L, S = rt.lib(), rt.stream(gt)
        gcat = ctx.slot.buffer()
        for a, l, g in zip(offs, lins, grads):
            if g is None:
                continue
            w = r4(l.weight.shape[0])
            dst = gcat[:, a:a + w]
            if g.data_ptr() != dst.data_ptr():                # (not written in place: a block that did not see the slot)
                dst.copy_(g[:, :w])
            N = l.weight.shape[0]
            L.cdf_linear_small_wgrad(P(dst), J, P(gt), gt.stride(0), P(ops.grad_of(l.weight)), P(ops.grad_of(l.bias)), B, N, K, S)
        dgt = None
        if ctx.needs_input_grad[1]:
            wcat = ctx.owner._tb_cache[1]
            dgt = torch.empty((B, r4(K)), device=gt.device, dtype=torch.float32)
            L.cdf_linear_small(P(gcat), J, P(wcat), K, 0, P(dgt), r4(K), B, J, K, S)     # dgt[b][k] = sum_j gcat[b][j] wcat[j][k]
            dgt = dgt[:, :K]
        _done(ctx)
        return None, dgt, None


class GradSlot:
    """A [B, J] gradient buffer handed out lazily (zeroed: blocks whose time bias gets no gradient leave zeros)."""

    def __init__(self, B, J, device):
        self.shape, self.device, self._buf = (B, J), device, None

    def buffer(self):
        if self._buf is None:
            self._buf = torch.zeros(self.shape, device=self.device, dtype=torch.float32)
        return self._buf

    def view(self, off, width):
        return self.buffer()[:, off:off + width]


class ConvFn(torch.autograd.Function):
    """A single dense convolution module (Down/Upsample, final 1x1, conv_in/out ...)."""

    @staticmethod
    def forward(ctx, anchor, x, mod, Cin, kind, stride, pad, dest=None):
        k = mod.weight.shape[-1]
        Cout = mod.weight.shape[0] if kind == "conv" else mod.weight.shape[1]
        # 4x4 stride-2 down / transposed up-sampling convs: every input pixel feeds several taps and N tiles, and the
        # same planes serve the weight gradient -> split once, use the LDS-DMA kernels (1x1 convs read x once: not worth it)
        xs = ops.split_bf16(x) if (_AUTO_PRESPLIT and (k > 1 or _PRESPLIT_1X1) and want_presplit(Cin, Cout, k)) else None
        # dest: a CatBuf (the conv writes its first half) or the destination view itself (a skip tensor written into its second half)
        ydst = dest.first() if isinstance(dest, CatBuf) else dest
        y = conv_forward(x, Cin, mod.weight, mod.bias, kind, stride, pad, xs=xs, **({"y": ydst} if ydst is not None else {}))
        ctx.mod, ctx.cfg = mod, (Cin, kind, stride, pad)
        _used(ctx, mod)
        ctx.has_xs = xs is not None
        ctx.save_for_backward(x, *(xs or (None, None)))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, x_hi, x_lo = ctx.saved_tensors
        Cin, kind, stride, pad = ctx.cfg
        xs = (x_hi, x_lo) if ctx.has_xs else None
        dys = ops.split_or_planes(dy) if xs is not None else None
        dx = conv_backward(x, Cin, dy, ctx.mod.weight, ctx.mod.bias, kind, stride, pad, need_dx=ctx.needs_input_grad[1], xs=xs, dys=dys)
        _done(ctx)
        return None, dx, None, None, None, None, None, None


class ConvNextBlockFn(torch.autograd.Function):
    """ConvNextBlock.forward (deblurring_diffusion_pytorch.py:156-165) as one node:
    h = ds_conv(x) + b + mlp(t); hn = LayerNorm(h); a = GELU(conv3x3(hn)); o = conv3x3(a) + res_conv(x)."""

    @staticmethod
    def forward(ctx, anchor, x, tbias, m, dest=None):
        dim, dim_out = m.dim, m.dim_out
        Cp = x.shape[-1]
        grad_on = ctx.needs_input_grad[0]   # anchor: True iff autograd is recording
        wdw = ops.packed(m.ds_conv.weight, "dw")
        h = ops.dwconv7(x, wdw, ops.padded_vec(m.ds_conv.bias, Cp), tbias)
        c1, c2 = m.net[1], m.net[3]
        B, H, W, _ = x.shape
        mid = c1.weight.shape[0]
        # operands that feed several GEMMs (fwd now, dgrad/wgrad later, every N tile) are split into bf16 hi/lo ONCE, by
        # the kernel that produces them (LayerNorm, the GELU epilogue of conv1, the GELU' epilogue of conv2's dgrad)
        sp1, sp2 = want_presplit(dim, mid, 3), want_presplit(mid, dim_out, 3)
        # When every consumer of a tensor reads its bf16 planes (fwd / dgrad GEMMs always, the wgrad GEMM from ops.WGRAD_SP_MIN_M
        # pixels up), its fp32 copy is not written at all: LN output, GELU output, conv2's data gradient.
        lean = _LEAN and B * H * W >= ops.WGRAD_SP_MIN_M
        hn_s = None
        if m.has_norm:
            if sp1:
                hn, mean, rstd, hn_s = ops.layernorm_fwd(h, m.net[0].g, m.net[0].b, m.net[0].eps, grad_on, split_out=True,
                                                         planes_only=lean)
            else:
                hn, mean, rstd = ops.layernorm_fwd(h, m.net[0].g, m.net[0].b, m.net[0].eps, grad_on)
        else:
            hn, mean, rstd = h, None, None
            hn_s = ops.split_bf16(hn) if sp1 else None
        pre = ops.new_feat(x, B, H, W, mid) if grad_on else None
        # image-side block (dim <= 4 input channels): direct vector-ALU convolutions instead of K <= 36 GEMMs
        ctx.cin4 = _CIN4 and ops.cin4_ok(hn, dim, c1.weight) and ops.cin4_ok(x, dim, c1.weight)
        ctx.pre_grad = False
        if ctx.cin4:
            if sp2:
                a, a_s = ops.conv_cin4_fwd(hn, c1.weight, c1.bias, act=ACT_GELU, pre=pre, split_out=True, planes_only=lean)
            else:
                a, a_s = ops.conv_cin4_fwd(hn, c1.weight, c1.bias, act=ACT_GELU, pre=pre), None
        elif sp2:
            # (pre-split path: the epilogue can hand back GELU'(v) in place of the pre-activation -- bit-identical to evaluating it in backward)
            ctx.pre_grad = bool(_PRE_GRAD and grad_on and hn_s is not None and _sp_suffix(dim * 9, mid))
            a, a_s = conv_forward(hn, dim, c1.weight, c1.bias, act=ACT_GELU, pre=pre, xs=hn_s, split_out=True, planes_only=lean,
                                  **({"pre_grad": True} if ctx.pre_grad else {}))
        else:
            a, a_s = conv_forward(hn, dim, c1.weight, c1.bias, act=ACT_GELU, pre=pre, xs=hn_s), None
        ctx.res4 = bool(m.has_res_conv and _CIN4 and ops.cin4_ok(x, dim, m.res_conv.weight))
        if ctx.res4:
            res = ops.conv_cin4_fwd(x, m.res_conv.weight, m.res_conv.bias)
        elif m.has_res_conv:
            res = conv_forward(x, dim, m.res_conv.weight, m.res_conv.bias)
        else:
            res = x
        o = conv_forward(a, mid, c2.weight, c2.bias, res=res, xs=a_s, **({"y": dest.first()} if dest is not None else {}))
        ctx.m = m
        _used(ctx, m.ds_conv, m.net[0] if m.has_norm else None, c1, c2, m.res_conv if m.has_res_conv else None)
        ctx.has_t = tbias is not None
        ctx.tslot = getattr(tbias, "_cdf_gslot", None) if tbias is not None else None
        # the data gradient of this block goes to a GEMM as bf16 planes when its input came from a ConvNeXt block or a strided conv (not from
        # an attention block or a skip join, whose backward passes read fp32): written by the depthwise data-gradient kernel itself
        ctx.dx_planes = bool(grad_on and ops.want_grad_planes(x.shape[-1]) and not getattr(x, "_cdf_grad_f32", False))
        ctx.split = (hn_s is not None, a_s is not None)
        ctx.save_for_backward(x, h, hn if m.has_norm else None, mean, rstd, pre, a, *(hn_s or (None, None)), *(a_s or (None, None)))
        return o

    @staticmethod
    def backward(ctx, do):
        x, h, hn, mean, rstd, pre, a, hn_hi, hn_lo, a_hi, a_lo = ctx.saved_tensors
        m = ctx.m
        if hn is None:
            hn = h
        hn_s = (hn_hi, hn_lo) if ctx.split[0] else None
        a_s = (a_hi, a_lo) if ctx.split[1] else None
        dim = m.dim
        c1, c2 = m.net[1], m.net[3]
        mid = c1.weight.shape[0]
        need_dx = ctx.needs_input_grad[1]
        # residual branch
        dx = None
        if ctx.res4:
            dx = ops.conv_cin4_bwd(x, do, m.res_conv.weight, m.res_conv.bias, need_dx)
        elif m.has_res_conv:
            dx = conv_backward(x, dim, do, m.res_conv.weight, m.res_conv.bias, need_dx=need_dx)
        # (without a res_conv the residual gradient is `do` itself: added by the depthwise data-gradient kernel below)
        # conv2 -> (fused GELU') -> conv1
        do_s = ops.split_or_planes(do) if a_s is not None else None       # (planes written by the producer of `do`, else cdf_split_bf16)
        if hn_s is not None:
            lean = _LEAN and a_s is not None and a.shape[0] * a.shape[1] * a.shape[2] >= ops.WGRAD_SP_MIN_M
            dpre, dpre_s = conv_backward(a, mid, do, c2.weight, c2.bias, mul=pre, mul_mode=3 if ctx.pre_grad else 1, xs=a_s, dys=do_s, split_dx=True,
                                         planes_only=lean)
        else:
            dpre, dpre_s = conv_backward(a, mid, do, c2.weight, c2.bias, mul=pre, mul_mode=3 if ctx.pre_grad else 1, xs=a_s, dys=do_s), None
        ln_done = False
        if ctx.cin4:
            # weight / bias gradient by the direct kernel (reads dpre once); the data gradient in two stages (sum over the mid
            # channels per pixel as a 1x1 GEMM, then the nine shifted 3-vectors: ops.conv_cin_dgrad2) instead of a K = 9 mid
            # gather-GEMM with 3 useful output columns (0.4 ms at 128 x 128)
            ops.conv_cin4_bwd(hn, dpre, c1.weight, c1.bias, False)
            if _CIN_DGRAD2 and c1.weight.shape[-1] == 3:
                dhn = ops.conv_cin_dgrad2(dpre, mid, c1.weight)
            else:
                pd = _conv_plans("conv", hn.shape[1], hn.shape[2], 3, 1, (1, 1, 1, 1))[1]
                dhn = ops.conv_gemm(pd, dpre, mid, ops.packed(c1.weight, "conv_dgrad"), dim)
        else:
            dhn = conv_backward(hn, dim, dpre, c1.weight, c1.bias, xs=hn_s, dys=dpre_s, ln=(h, m.net[0], mean, rstd) if m.has_norm else None)
            if m.has_norm:
                dhn, ln_done = dhn
        if ln_done:
            dh = dhn
        elif m.has_norm:
            dh = ops.layernorm_bwd(dhn, h, m.net[0].g, m.net[0].b, mean, rstd)
        else:
            dh = dhn
        dsb_out = ctx.tslot[0].view(ctx.tslot[1], x.shape[-1]) if ctx.tslot is not None else None
        dtb = ops.dwconv7_wgrad(x, dh, m.ds_conv.weight, m.ds_conv.bias, ctx.has_t, dsb_out=dsb_out)
        if need_dx:
            if m.has_res_conv:
                dx = ops.dwconv7(dh, ops.packed(m.ds_conv.weight, "dw"), None, None, flip=1, y=dx, accumulate=1, planes=ctx.dx_planes)
            else:
                dx = ops.dwconv7(dh, ops.packed(m.ds_conv.weight, "dw"), None, None, flip=1, res=do, planes=ctx.dx_planes)
        _done(ctx)
        return None, dx, dtb, None, None


def kv_forward(xn, dim, w_qkv):
    """kv = xn . Wkv^T ([B,H,W,2 HD]): the k | v rows of to_qkv as a 1x1 convolution of their own (ops.linattn_fold)."""
    N2 = w_qkv.shape[0] // 3 * 2
    plan = _conv_plans("conv", xn.shape[1], xn.shape[2], 1, 1, (0, 0, 0, 0))[0]
    return ops.conv_gemm(plan, xn, dim, ops.packed(w_qkv, "kv_fwd" + _sp_suffix(dim, N2)), N2)


def kv_backward(xn, dim, dkv, w_qkv, dxn):
    """rows HD .. 3 HD of to_qkv.weight.grad += dkv^T xn;  dxn += dkv . Wkv (dxn None: the data gradient was taken elsewhere)."""
    HD = w_qkv.shape[0] // 3
    _, pd, pw = _conv_plans("conv", xn.shape[1], xn.shape[2], 1, 1, (0, 0, 0, 0))
    ops.wgrad_into(ops.grad_of(w_qkv)[HD:], pw, xn, dim, dkv, 2 * HD, 1, 1, dim)
    if dxn is None:
        return None
    return ops.conv_gemm(pd, dkv, 2 * HD, ops.packed(w_qkv, "kv_dgrad" + _sp_suffix(2 * HD, dim)), dim, y=dxn, accumulate=1)


def kv_planes_ok(xn, dim, heads):
    """The k | v projection's backward runs on pre-split operand planes: dk | dv leave the attention backward kernel AS bf16 hi / lo planes
    (same bytes as fp32) and LayerNorm's output is kept as planes too, so the data gradient (K = 2 HD = 256 -> dim) and the weight
    gradient are the LDS-DMA plane GEMMs instead of the in-kernel-split / exact-fp32 kernels reading 1 GB of fp32 dk | dv each at
    128 x 128 (0.53 + 0.41 ms per step there)."""
    B, H, W, _ = xn.shape
    return (_KV_PLANES and rt.precision != "f32" and heads <= 4 and dim % 8 == 0 and dim >= 64 and B * H * W >= ops.WGRAD_SP_MIN_M
            and xn.device.type != "meta")


def kv_backward_planes(xn, xn_s, dim, dkv_s, w_qkv, dxn):
    """kv_backward with both operands as bf16 planes: rows HD .. 3 HD of to_qkv.weight.grad += dkv^T xn;  dxn += dkv . Wkv."""
    HD = w_qkv.shape[0] // 3
    B, H, W, _ = xn.shape
    _, pd, pw = _conv_plans("conv", H, W, 1, 1, (0, 0, 0, 0))
    ops.wgrad_into(ops.grad_of(w_qkv)[HD:], pw, xn, dim, ops.shape_only(B, H, W, 2 * HD), 2 * HD, 1, 1, dim, xa_s=xn_s, xb_s=dkv_s)
    return ops.conv_gemm_presplit(pd, dkv_s, 2 * HD, ops.packed(w_qkv, "kv_dgrad_sp"), dim, y=dxn, accumulate=1)


_KV_PLANES = os.environ.get("CDF_KV_PLANES", "1") != "0"
_ATTN_QFOLD = os.environ.get("CDF_ATTN_QFOLD", "1") != "0"   # q projection folded into the attention product too (dim <= heads*32)
_ATTN_KVCTX = os.environ.get("CDF_ATTN_KVCTX", "1") != "0"   # ... with the k|v projection and the context in one kernel (ops.linattn_kvctx)


class LinAttnBlockFn(torch.autograd.Function):
    """Residual(PreNorm(dim, LinearAttention(dim))) (deblurring_diffusion_pytorch.py:83-89,123-131,167-187)."""

    @staticmethod
    def forward(ctx, anchor, x, m, dest=None):
        norm, att = m.fn.norm, m.fn.fn
        dim = x.shape[-1]
        grad_on = ctx.needs_input_grad[0]   # anchor: True iff autograd is recording
        ydst = {"y": dest.second()} if dest is not None else {}      # skip tensor: produced in place in its concat buffer (CatBuf)
        ctx.qfold_possible = dim % 4 == 0 and _ATTN_FUSED >= 1 and _ATTN_QFOLD and dim <= att.heads * 32 and att.heads <= 4
        ctx.kv_planes = bool(grad_on and ctx.qfold_possible and kv_planes_ok(x, dim, att.heads))
        xn_s = None
        if ctx.kv_planes:
            xn, mean, rstd, xn_s = ops.layernorm_fwd(x, norm.g, norm.b, norm.eps, grad_on, split_out=True)
        else:
            xn, mean, rstd = ops.layernorm_fwd(x, norm.g, norm.b, norm.eps, grad_on)
        ctx.m = m
        _used(ctx, norm, att.to_qkv, att.to_out)
        # to_out folded into the attention product where that shrinks the batched GEMMs (dim <= heads*32: the 128- and 64-pixel
        # levels of the CelebA net); above, the per-image [HD x dim] GEMMs over a few hundred pixels are latency-bound and the
        # plain form (K = 32 head products + a dense to_out GEMM over all images) is faster (16 x 16: 1.75 vs 2.2 ms per 12 passes)
        ctx.fused = dim % 4 == 0 and (_ATTN_FUSED == 2 or (_ATTN_FUSED == 1 and dim <= att.heads * 32))
        ctx.qfold = ctx.fused and _ATTN_QFOLD and dim <= att.heads * 32 and att.heads <= 4
        if ctx.qfold:
            # q folded in as well: only k | v are projected, y = xn . N_b + b + x (ops.linattn_fold)
            if _ATTN_KVCTX and ops.linattn_kvctx_ok(xn, dim, att.heads):
                kv, cx, cxs, kmax, ksum = ops.linattn_kvctx(xn, dim, att.to_qkv.weight, att.heads, att.scale)    # one pass: k | v never re-read
            else:
                kv = kv_forward(xn, dim, att.to_qkv.weight)
                cx, cxs, kmax, ksum = ops.linattn_context(kv, att.heads, att.scale, koff=0)
            y, Mb, Nb = ops.linattn_fold(xn, cxs, att.to_qkv.weight, att.to_out.weight, att.to_out.bias, x, att.heads, **ydst)
            ctx.save_for_backward(x, xn, mean, rstd, kv, Mb, cx, cxs, kmax, ksum, Nb, *(xn_s or (None, None)))
            y._cdf_grad_f32 = True                    # (this block's backward reads its incoming gradient as fp32)
            return y
        qkv = conv_forward(xn, dim, att.to_qkv.weight, None)
        if ctx.fused:
            # output projection folded into the attention product: the attention output is never materialised (ops.linattn_project)
            cx, cxs, kmax, ksum = ops.linattn_context(qkv, att.heads, att.scale)
            y, Mb = ops.linattn_project(qkv, cxs, att.to_out.weight, att.to_out.bias, x, att.heads, **ydst)
            ctx.save_for_backward(x, xn, mean, rstd, qkv, Mb, cx, cxs, kmax, ksum)
            y._cdf_grad_f32 = True
            return y
        o, cx, cxs, kmax, ksum = ops.linattn_fwd(qkv, att.heads, att.scale)
        y = conv_forward(o, att.heads * 32, att.to_out.weight, att.to_out.bias, res=x, **ydst)
        ctx.save_for_backward(x, xn, mean, rstd, qkv, o, cx, cxs, kmax, ksum)
        y._cdf_grad_f32 = True
        return y

    @staticmethod
    def backward(ctx, dy):
        norm, att = ctx.m.fn.norm, ctx.m.fn.fn
        if ctx.qfold:
            x, xn, mean, rstd, kv, Mb, cx, cxs, kmax, ksum, Nb, xn_hi, xn_lo = ctx.saved_tensors
            dim = x.shape[-1]
            dxn, dctx, rvec = ops.linattn_fold_bwd(xn, dy, Mb, Nb, cx, cxs, att.to_qkv.weight, att.to_out.weight, att.to_out.bias,
                                                   att.heads, att.scale)
            if ctx.kv_planes:
                B_, H_, W_, C2 = kv.shape
                dkv_s = ops.split_planes_like(kv, B_, H_, W_, C2)
                ops.linattn_bwd_core(kv, dctx, rvec, kmax, ksum, None, att.heads, koff=0, planes=dkv_s)
                kv_backward_planes(xn, (xn_hi, xn_lo), dim, dkv_s, att.to_qkv.weight, dxn)
            else:
                dkv = torch.empty(kv.shape, device=kv.device, dtype=torch.float32)
                ops.linattn_bwd_core(kv, dctx, rvec, kmax, ksum, dkv, att.heads, koff=0)
                kv_backward(xn, dim, dkv, att.to_qkv.weight, dxn)
            dx = ops.layernorm_bwd(dxn, x, norm.g, norm.b, mean, rstd, add=dy, planes=ops.want_grad_planes(dim))     # + the residual branch, same pass
            _done(ctx)
            return None, dx, None, None
        x, xn, mean, rstd, qkv, o, cx, cxs, kmax, ksum = ctx.saved_tensors
        dim, HD = x.shape[-1], att.heads * 32
        if ctx.fused:
            B, H, W, _ = qkv.shape
            dqkv = torch.empty((B, H, W, 3 * HD), device=qkv.device, dtype=torch.float32)
            dctx, rvec = ops.linattn_project_bwd(qkv, dy, o, cx, cxs, att.to_out.weight, att.to_out.bias, dqkv, att.heads, att.scale)   # (o = Mb here)
            ops.linattn_bwd_core(qkv, dctx, rvec, kmax, ksum, dqkv, att.heads)
        else:
            do = conv_backward(o, HD, dy, att.to_out.weight, att.to_out.bias)
            dqkv = ops.linattn_bwd(qkv, do, cx, cxs, kmax, ksum, att.heads, att.scale)
        dxn = conv_backward(xn, dim, dqkv, att.to_qkv.weight, None)
        dx = ops.layernorm_bwd(dxn, x, norm.g, norm.b, mean, rstd, add=dy, planes=ops.want_grad_planes(dim))
        _done(ctx)
        return None, dx, None, None


# ===================================================================================================
# DDPM `Model` family (Model2.py)
# ===================================================================================================
GN_GROUPS, GN_EPS = 32, 1e-6


def _seed():
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class GroupNormFn(torch.autograd.Function):
    """Normalize(C) [+ swish] as a standalone node (norm_out -> nonlinearity, Model2.py:329-330)."""

    @staticmethod
    def forward(ctx, anchor, x, norm, silu):
        y, mean, rstd = ops.groupnorm_fwd(x, norm.weight, norm.bias, GN_GROUPS, GN_EPS, silu)
        ctx.norm, ctx.silu = norm, silu
        _used(ctx, norm)
        ctx.save_for_backward(x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        dx = ops.groupnorm_bwd(dy, x, ctx.norm.weight, ctx.norm.bias, mean, rstd, GN_GROUPS, ctx.silu)
        _done(ctx)
        return None, dx, None, None


class Upsample2Fn(torch.autograd.Function):
    """F.interpolate(scale_factor=2, mode='nearest') alone: `Upsample(with_conv=False)` (Model2.py:36-50)."""

    @staticmethod
    def forward(ctx, x, out=None):
        return ops.upsample2(x, out=out)

    @staticmethod
    def backward(ctx, dy):
        return ops.upsample2_bwd(dy.contiguous()), None                # each source pixel collects its 2 x 2 copies


class AvgPool2Fn(torch.autograd.Function):
    """F.avg_pool2d(x, kernel_size=2, stride=2): `Downsample(with_conv=False)` (Model2.py:53-73)."""

    @staticmethod
    def forward(ctx, x, out=None):
        B, H, W, C = x.shape
        y = torch.empty((B, H // 2, W // 2, C), device=x.device, dtype=torch.float32) if out is None else out
        rt.lib().cdf_pool2d(P(rt.check(x)), ops.ld_of(x), P(y), ops.ld_of(y), B, H, W, C, 2, 2, 0, 1, rt.stream(x))
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = ops.upsample2(dy.contiguous())                             # every input pixel gets a quarter of its window's gradient
        rt.lib().cdf_scale(P(dx), dx.numel(), 0.25, rt.stream(dx))
        return dx, None


class UpsampleConvFn(torch.autograd.Function):
    """F.interpolate(scale 2, nearest) -> Conv2d 3x3 (Model2.py:36-50); the x2 map is rebuilt in backward."""

    @staticmethod
    def forward(ctx, anchor, x, conv, out=None):
        up = ops.upsample2(x)
        C = x.shape[-1]
        ctx.sp = _AUTO_PRESPLIT and want_presplit(C, conv.weight.shape[0], 3)
        y = conv_forward(up, C, conv.weight, conv.bias, xs=ops.split_bf16(up) if ctx.sp else None, **({"y": out} if out is not None else {}))
        ctx.conv = conv
        _used(ctx, conv)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        up = ops.upsample2(x)
        ups, dys = (ops.split_bf16(up), ops.split_bf16(dy)) if ctx.sp else (None, None)
        dup = conv_backward(up, x.shape[-1], dy, ctx.conv.weight, ctx.conv.bias, xs=ups, dys=dys)
        _done(ctx)
        return None, ops.upsample2_bwd(dup), None, None


class ResnetBlockFn(torch.autograd.Function):
    """ResnetBlock.forward (Model2.py:114-133): GN+swish -> conv1 (+temb bias) -> GN+swish -> dropout -> conv2 (+shortcut)."""

    @staticmethod
    def forward(ctx, anchor, x, tbias, m, out=None):
        cin, cout = m.in_channels, m.out_channels
        # the 3x3 convs' operands are split into bf16 hi/lo planes once (forward, data gradient and weight gradient read them)
        sp1 = _AUTO_PRESPLIT and want_presplit(cin, cout, 3)
        sp2 = _AUTO_PRESPLIT and want_presplit(cout, cout, 3)
        # GroupNorm + SiLU (+ dropout) write the next conv's bf16 operand planes themselves; when every consumer reads the planes
        # (fwd / dgrad GEMMs always, the weight-gradient GEMM from ops.WGRAD_SP_MIN_M pixels up) the fp32 copy is not written at all
        B, H, W, _ = x.shape
        lean = _LEAN and B * H * W >= ops.WGRAD_SP_MIN_M
        if sp1:
            h1, mean1, rstd1, h1_s = ops.groupnorm_fwd(x, m.norm1.weight, m.norm1.bias, GN_GROUPS, GN_EPS, True, split_out=True, planes_only=lean)
        else:
            (h1, mean1, rstd1), h1_s = ops.groupnorm_fwd(x, m.norm1.weight, m.norm1.bias, GN_GROUPS, GN_EPS, True), None
        h2 = conv_forward(h1, cin, m.conv1.weight, m.conv1.bias, sbias=tbias, xs=h1_s)
        p = m.dropout.p if m.training else 0.0
        seed = _seed() if p > 0 else 0
        drop = (p, seed) if p > 0 else None
        if sp2:
            h3, mean2, rstd2, h3_s = ops.groupnorm_fwd(h2, m.norm2.weight, m.norm2.bias, GN_GROUPS, GN_EPS, True, split_out=True, planes_only=lean,
                                                       drop=drop)
        else:
            (h3, mean2, rstd2), h3_s = ops.groupnorm_fwd(h2, m.norm2.weight, m.norm2.bias, GN_GROUPS, GN_EPS, True, drop=drop), None
        if cin != cout:
            sc_mod = m.conv_shortcut if m.use_conv_shortcut else m.nin_shortcut
            sc = conv_forward(x, cin, sc_mod.weight, sc_mod.bias)
        else:
            sc = x
        o = conv_forward(h3, cout, m.conv2.weight, m.conv2.bias, res=sc, xs=h3_s, **({"y": out} if out is not None else {}))
        ctx.m, ctx.drop = m, (p, seed)
        _used(ctx, m.norm1, m.conv1, m.norm2, m.conv2, sc_mod if cin != cout else None)
        ctx.split = (sp1, sp2)
        ctx.save_for_backward(x, h1, h2, h3, mean1, rstd1, mean2, rstd2, *(h1_s or (None, None)), *(h3_s or (None, None)))
        return o

    @staticmethod
    def backward(ctx, do):
        x, h1, h2, h3, mean1, rstd1, mean2, rstd2, h1_hi, h1_lo, h3_hi, h3_lo = ctx.saved_tensors
        m = ctx.m
        cin, cout = m.in_channels, m.out_channels
        p, seed = ctx.drop
        h1_s = (h1_hi, h1_lo) if ctx.split[0] else None
        h3_s = (h3_hi, h3_lo) if ctx.split[1] else None
        sc_mod = None
        if cin != cout:
            sc_mod = m.conv_shortcut if m.use_conv_shortcut else m.nin_shortcut
            dx = conv_backward(x, cin, do, sc_mod.weight, sc_mod.bias)
        else:
            dx = ops.copy_feat(do)
        dh3 = conv_backward(h3, cout, do, m.conv2.weight, m.conv2.bias, xs=h3_s, dys=ops.split_bf16(do) if h3_s is not None else None)
        dh2 = ops.groupnorm_bwd(dh3, h2, m.norm2.weight, m.norm2.bias, mean2, rstd2, GN_GROUPS, True, drop=(p, seed) if p > 0 else None)
        dtb = ops.colsum_new(dh2, cout, dh2.shape[0])
        dh1 = conv_backward(h1, cin, dh2, m.conv1.weight, m.conv1.bias, xs=h1_s, dys=ops.split_bf16(dh2) if h1_s is not None else None)
        ops.groupnorm_bwd(dh1, x, m.norm1.weight, m.norm1.bias, mean1, rstd1, GN_GROUPS, True, dx=dx)
        _done(ctx)
        return None, dx, dtb, None, None


class AttnBlockFn(torch.autograd.Function):
    """AttnBlock.forward (Model2.py:164-188): GN -> q,k,v 1x1 -> softmax(q k^T C^-0.5) v -> proj_out + x."""

    @staticmethod
    def forward(ctx, anchor, x, m, out=None):
        B, H, W, C = x.shape
        n = H * W
        hn, mean, rstd = ops.groupnorm_fwd(x, m.norm.weight, m.norm.bias, GN_GROUPS, GN_EPS, False)
        q = conv_forward(hn, C, m.q.weight, m.q.bias).view(B, n, C)
        k = conv_forward(hn, C, m.k.weight, m.k.bias).view(B, n, C)
        v = conv_forward(hn, C, m.v.weight, m.v.bias).view(B, n, C)
        s = ops.bgemm_nt(q, k)                               # [B, n, n]  w_[b,i,j] = sum_c q[b,i,c] k[b,j,c]
        pm = ops.softmax_rows(s, n, float(int(C) ** (-0.5)))
        o = ops.bgemm_nn(pm, v, K=n).view(B, H, W, C)        # h_[b,i,c] = sum_j P[b,i,j] v[b,j,c]
        y = conv_forward(o, C, m.proj_out.weight, m.proj_out.bias, res=x, **({"y": out} if out is not None else {}))
        ctx.m = m
        _used(ctx, m.norm, m.q, m.k, m.v, m.proj_out)
        ctx.save_for_backward(x, hn, mean, rstd, q, k, v, pm, o)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, hn, mean, rstd, q, k, v, pm, o = ctx.saved_tensors
        m = ctx.m
        B, H, W, C = x.shape
        n = H * W
        scale = float(int(C) ** (-0.5))
        do = conv_backward(o, C, dy, m.proj_out.weight, m.proj_out.bias).view(B, n, C)
        dp = ops.bgemm_nt(do, v)                              # dP[b,i,j] = sum_c dO[b,i,c] v[b,j,c]
        dv = ops.bgemm_tn(pm, do, CA=n)                       # dv[b,j,c] = sum_i P[b,i,j] dO[b,i,c]
        ds = ops.softmax_rows_bwd(pm, dp, n, scale)
        dq = ops.bgemm_nn(ds, k, K=n)                         # dq[b,i,c] = sum_j dS[b,i,j] k[b,j,c]
        dk = ops.bgemm_tn(ds, q, CA=n)                        # dk[b,j,c] = sum_i dS[b,i,j] q[b,i,c]
        dhn = conv_backward(hn, C, dq.view(B, H, W, C), m.q.weight, m.q.bias)
        conv_backward(hn, C, dk.view(B, H, W, C), m.k.weight, m.k.bias, dx=dhn, dx_accumulate=1)
        conv_backward(hn, C, dv.view(B, H, W, C), m.v.weight, m.v.bias, dx=dhn, dx_accumulate=1)
        dx = ops.copy_feat(dy)
        ops.groupnorm_bwd(dhn, x, m.norm.weight, m.norm.bias, mean, rstd, GN_GROUPS, False, dx=dx)
        _done(ctx)
        return None, dx, None, None

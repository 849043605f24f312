"""Block-level autograd nodes of the `Unet` with bf16 ACTIVATION STORAGE -- the "bf16" mode BASELINE configs 3 and 5 name
(bf16, fp32 accumulate, fp32 master weights: SURVEY 8(d)) as an engine of its own rather than fp32 tensors with bf16 GEMM operands.

One bf16 plane is the ONLY stored form of every GEMM input, every activation saved for backward (block input, depthwise output, LayerNorm
output, pre-activation, GELU output) and the inter-block residual stream in both directions; GEMM epilogues write bf16 only.  fp32 stays
where the arithmetic needs it: MFMA accumulators, LayerNorm statistics, softmax / context of the linear attention, time-embedding MLP and
biases, master weights, every parameter gradient, degradations, loss, optimizer.  The image-side block (3 -> 64 channels, 4-channel
fp32 input, no LayerNorm) keeps its 4-channel tensors in fp32 (a per-channel gradient there is a sum over every pixel of the batch of
values rounded at the stream's precision: round 5's simulation of this engine in the oracle put `downs.0.0.mlp.1.bias` at 9 % instead of
3 % of its tolerance scale with a bf16 4-channel stream) and hands a bf16 tensor to the next block.

Reference blocks: ConvNextBlock DEBLUR:135-165, Residual(PreNorm(LinearAttention)) DEBLUR:83-89,123-131,167-187, Down / Upsample
DEBLUR:105-109, Unet.forward DEBLUR:256-282.  Tolerance against the fp32 oracle: runtime.BF16_TOLERANCE (unchanged by the storage type),
asserted by tests/test_bf16_storage.py (simulator) and tests/test_gpu_parity2.py (MI355X, bench shape).
"""
import os

import torch

from . import functions as F_
from . import ops
from . import runtime as rt
from .functions import _conv_plans, _done, _used

ACT_GELU = F_.ACT_GELU


def enabled():
    """bf16 tensors between kernels: the default of the "bf16" arithmetic mode (COLDDIFF_BF16_STORAGE=0 keeps fp32 tensors with bf16
    GEMM operand planes -- the round 2-4 form of the mode, for A/B)."""
    return rt.precision == "bf16" and os.environ.get("COLDDIFF_BF16_STORAGE", "1") != "0"


def enabled_for(net):
    """Does this Unet's forward run the bf16 stream right now (mode on and every block's channel counts legal)?"""
    return enabled() and hasattr(net, "_bf16_ok") and net._bf16_ok()


def block_ok(m):
    """A ConvNeXt block whose every tensor is a legal bf16 feature map (channel counts in multiples of 8)."""
    return m.dim % 8 == 0 and m.dim_out % 8 == 0 and m.net[1].weight.shape[0] % 8 == 0


# -- dense convolutions on bf16 tensors ------------------------------------------------------------------------------------------------
def conv_fwd(x, Cin, weight, bias, kind="conv", stride=1, pad=None, **epi):
    k = weight.shape[-1]
    if pad is None:
        pad = (k // 2,) * 4
    _, H, W, _ = x.shape
    plan = _conv_plans(kind, H, W, k, stride, pad)[0]
    Cout = weight.shape[0] if kind == "conv" else weight.shape[1]
    wp = ops.packed(weight, ("conv_fwd" if kind == "conv" else "convT_fwd") + "_sp")
    return ops.conv_gemm_bf(plan, x, Cin, wp, Cout, bias=bias, **epi)


def conv_bwd(x, Cin, dy, weight, bias, kind="conv", stride=1, pad=None, need_dx=True, mul=None, mul_mode=0):
    """Weight / bias gradients accumulated (fp32), data gradient returned as a bf16 tensor (optionally x GELU'(mul))."""
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
    fuse_bias = bias is not None and kind == "conv"
    ops.wgrad_into(ops.grad_of(weight), pw, x, Cin, dy, Cout, 1, s_r, s_c, gbias=ops.grad_of(bias) if fuse_bias else None,
                   xa_s=(x, None), xb_s=(dy, None))
    if bias is not None and not fuse_bias:
        ops.colsum_into(ops.grad_of(bias), dy, Cout)
    if not need_dx:
        return None
    wd = ops.packed(weight, ("conv_dgrad" if kind == "conv" else "convT_dgrad") + "_sp")
    return ops.conv_gemm_bf(pd, dy, Cout, wd, Cin, mul=mul, mul_mode=mul_mode)


# -- boundaries -------------------------------------------------------------------------------------------------------------------------
class ToF32(torch.autograd.Function):
    """bf16 feature map -> fp32 (the stream leaving towards the 3-channel output conv); backward rounds the gradient."""

    @staticmethod
    def forward(ctx, x):
        return ops.to_f32(x)

    @staticmethod
    def backward(ctx, dy):
        return ops.to_bf16(dy.contiguous() if dy.stride(-1) != 1 else dy)


class JoinBF(torch.autograd.Function):
    """F_.Join for a bf16 CatBuf."""

    @staticmethod
    def forward(ctx, a, b, cat):
        assert a.data_ptr() == cat.buf.data_ptr() and b.data_ptr() == cat.buf.data_ptr() + 2 * cat.Cx and a.shape[-1] == cat.Cx and \
            b.shape[-1] == cat.Ch and ops.ld_of(a) == cat.Cx + cat.Ch and ops.ld_of(b) == cat.Cx + cat.Ch, "Join: the halves are not the two slices of this CatBuf"
        ctx.Ca = cat.Cx
        return cat.buf.view(cat.buf.shape)

    @staticmethod
    def backward(ctx, d):
        return d[..., :ctx.Ca], d[..., ctx.Ca:], None


# -- blocks -----------------------------------------------------------------------------------------------------------------------------
class ConvFnBF(torch.autograd.Function):
    """Down / Upsample (4 x 4 stride-2 conv / transposed conv) on the bf16 stream: the tensor is the GEMM's operand plane as it is."""

    @staticmethod
    def forward(ctx, anchor, x, mod, Cin, kind, stride, pad, dest=None):
        ydst = dest.first() if isinstance(dest, F_.CatBuf) else dest
        y = conv_fwd(x, Cin, mod.weight, mod.bias, kind, stride, pad, **({"y": ydst} if ydst is not None else {}))
        ctx.mod, ctx.cfg = mod, (Cin, kind, stride, pad)
        _used(ctx, mod)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        Cin, kind, stride, pad = ctx.cfg
        dx = conv_bwd(x, Cin, dy, ctx.mod.weight, ctx.mod.bias, kind, stride, pad, need_dx=ctx.needs_input_grad[1])
        _done(ctx)
        return None, dx, None, None, None, None, None, None


class ConvNextBlockBF(torch.autograd.Function):
    """ConvNextBlock.forward (DEBLUR:156-165) on the bf16 stream:
    h = ds_conv(x) + b + mlp(t); hn = LayerNorm(h); a = GELU(conv3x3(hn)); o = conv3x3(a) + res_conv(x) -- x, h, hn, pre, a, res, o
    all bf16; backward likewise (do, dpre, dhn, dh, dx)."""

    @staticmethod
    def forward(ctx, anchor, x, tbias, m, dest=None):
        dim = m.dim
        grad_on = ctx.needs_input_grad[0]
        B, H, W, _ = x.shape
        # an fp32 input is the image-side block's output ENTERING the stream: rounded here, and the data gradient goes back to that block
        # in fp32 (its per-channel bias gradients are sums over every pixel of the batch of this tensor's values: rounding it at the
        # stream's precision put them at 5 % of their scale against a 6 % bound on one seed, fp32 keeps them where fp32 tensors had them)
        ctx.in_f32 = not ops.is_bf(x)
        if ctx.in_f32:
            x = ops.to_bf16(x[..., :dim] if x.shape[-1] != dim else x)
        wdw = ops.packed(m.ds_conv.weight, "dw")
        h = ops.dwconv7_bf(x, wdw, m.ds_conv.bias, tbias)
        c1, c2 = m.net[1], m.net[3]
        mid = c1.weight.shape[0]
        if m.has_norm:
            hn, mean, rstd = ops.layernorm_fwd_bf(h, m.net[0].g, m.net[0].b, m.net[0].eps, grad_on)
        else:
            hn, mean, rstd = h, None, None
        pre = ops.new_bf(x, B, H, W, mid) if grad_on else None
        a = conv_fwd(hn, dim, c1.weight, c1.bias, act=ACT_GELU, pre=pre, pre_grad=pre is not None and F_._PRE_GRAD)   # pre = GELU'(v)
        res = conv_fwd(x, dim, m.res_conv.weight, m.res_conv.bias) if m.has_res_conv else x
        o = conv_fwd(a, mid, c2.weight, c2.bias, res=res, **({"y": dest.first()} if dest is not None else {}))
        ctx.m = m
        _used(ctx, m.ds_conv, m.net[0] if m.has_norm else None, c1, c2, m.res_conv if m.has_res_conv else None)
        ctx.has_t = tbias is not None
        ctx.tslot = getattr(tbias, "_cdf_gslot", None) if tbias is not None else None
        ctx.save_for_backward(x, h, hn if m.has_norm else None, mean, rstd, pre, a)
        return o

    @staticmethod
    def backward(ctx, do):
        x, h, hn, mean, rstd, pre, a = ctx.saved_tensors
        m = ctx.m
        if hn is None:
            hn = h
        dim = m.dim
        c1, c2 = m.net[1], m.net[3]
        mid = c1.weight.shape[0]
        need_dx = ctx.needs_input_grad[1]
        if do.stride(-1) != 1:
            do = do.contiguous()
        dx = conv_bwd(x, dim, do, m.res_conv.weight, m.res_conv.bias, need_dx=need_dx) if m.has_res_conv else None
        dpre = conv_bwd(a, mid, do, c2.weight, c2.bias, mul=pre, mul_mode=3 if F_._PRE_GRAD else 1)      # conv2 -> (x the stored GELU')
        dhn = conv_bwd(hn, dim, dpre, c1.weight, c1.bias)
        dh = ops.layernorm_bwd_bf(dhn, h, m.net[0].g, m.net[0].b, mean, rstd) if m.has_norm else dhn
        dsb_out = ctx.tslot[0].view(ctx.tslot[1], x.shape[-1]) if ctx.tslot is not None else None
        dtb = ops.dwconv7_wgrad_bf(x, dh, m.ds_conv.weight, m.ds_conv.bias, ctx.has_t, dsb_out=dsb_out)
        if need_dx:
            wdw = ops.packed(m.ds_conv.weight, "dw")
            if ctx.in_f32:
                dx = ops.dwconv7_bf(dh, wdw, None, None, flip=1, res=dx if m.has_res_conv else do, out_f32=True)
            elif m.has_res_conv:
                ops.dwconv7_bf(dh, wdw, None, None, flip=1, y=dx, accumulate=1)
            else:
                dx = ops.dwconv7_bf(dh, wdw, None, None, flip=1, res=do)                # + the residual gradient, same pass
        _done(ctx)
        return None, dx, dtb, None, None


class LinAttnBlockBF(torch.autograd.Function):
    """Residual(PreNorm(LinearAttention)) on the bf16 stream.  The block's inside -- LayerNorm output, k | v, softmax statistics, context,
    the per-image folded matrices -- is the fp32 computation of F_.LinAttnBlockFn (q folded form; its k | v GEMM takes bf16 operands
    in this mode as before); the stream tensors x / y / dy / dx cross the block's boundary as bf16: LayerNorm reads bf16 x, the output
    product's residual operand is read as bf16, y is rounded once when stored, the LayerNorm backward adds the bf16 residual gradient and
    stores bf16."""

    @staticmethod
    def forward(ctx, anchor, x, m, dest=None):
        y, saved = _attn_forward(ctx, x, m, dest)
        ctx.save_for_backward(*saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        return _attn_backward(ctx, dy)


def _attn_forward(ctx, x, m, dest):
    norm, att = m.fn.norm, m.fn.fn
    dim = x.shape[-1]
    grad_on = ctx.needs_input_grad[0]
    ctx.qfold = dim % 8 == 0 and dim <= att.heads * 32 and att.heads <= 4 and F_._ATTN_FUSED >= 1 and F_._ATTN_QFOLD
    ctx.kv_planes = bool(grad_on and ctx.qfold and F_.kv_planes_ok(x, dim, att.heads))     # k | v backward on operand planes (functions.py)
    if ctx.kv_planes:
        xn, mean, rstd, xnb = ops.layernorm_fwd_bf(x, norm.g, norm.b, norm.eps, grad_on, out_f32=True, planes=True)
    else:
        (xn, mean, rstd), xnb = ops.layernorm_fwd_bf(x, norm.g, norm.b, norm.eps, grad_on, out_f32=True), None
    ctx.m = m
    _used(ctx, norm, att.to_qkv, att.to_out)
    if ctx.qfold:
        # (the 128- and 64-pixel levels of the CelebA net: 80 % of the attention bytes) the output product reads the residual from the bf16
        # stream and writes its result into it (cdf_conv_gemm_io): x and y cross the boundary once each, as bf16
        if F_._ATTN_KVCTX and ops.linattn_kvctx_ok(xn, dim, att.heads):
            kv, cx, cxs, kmax, ksum = ops.linattn_kvctx(xn, dim, att.to_qkv.weight, att.heads, att.scale)
        else:
            kv = F_.kv_forward(xn, dim, att.to_qkv.weight)
            cx, cxs, kmax, ksum = ops.linattn_context(kv, att.heads, att.scale, koff=0)
        yb, Mb, Nb = ops.linattn_fold(xn, cxs, att.to_qkv.weight, att.to_out.weight, att.to_out.bias, x, att.heads,
                                      y=dest.second() if dest is not None else None)
        return yb, (x, xn, mean, rstd, kv, Mb, cx, cxs, kmax, ksum, Nb, xnb)
    xf = ops.to_f32(x)                                          # (deeper levels: the plain forms keep an fp32 residual operand)
    qkv = F_.conv_forward(xn, dim, att.to_qkv.weight, None)
    ctx.fused = dim % 4 == 0 and (F_._ATTN_FUSED == 2 or (F_._ATTN_FUSED == 1 and dim <= att.heads * 32))
    if ctx.fused:
        cx, cxs, kmax, ksum = ops.linattn_context(qkv, att.heads, att.scale)
        y, Mb = ops.linattn_project(qkv, cxs, att.to_out.weight, att.to_out.bias, xf, att.heads)
        saved = (x, xn, mean, rstd, qkv, Mb, cx, cxs, kmax, ksum)
    else:
        o, cx, cxs, kmax, ksum = ops.linattn_fwd(qkv, att.heads, att.scale)
        y = F_.conv_forward(o, att.heads * 32, att.to_out.weight, att.to_out.bias, res=xf)
        saved = (x, xn, mean, rstd, qkv, o, cx, cxs, kmax, ksum)
    yb = ops.to_bf16(y[..., :dim] if y.shape[-1] != dim else y, dest.second() if dest is not None else None)
    return yb, saved


def _attn_backward(ctx, dy):
    norm, att = ctx.m.fn.norm, ctx.m.fn.fn
    if dy.stride(-1) != 1:
        dy = dy.contiguous()
    dyf = ops.to_f32(dy)
    if ctx.qfold:
        x, xn, mean, rstd, kv, Mb, cx, cxs, kmax, ksum, Nb, xnb = ctx.saved_tensors
        dim = x.shape[-1]
        dxn, dctx, rvec = ops.linattn_fold_bwd(xn, dyf, Mb, Nb, cx, cxs, att.to_qkv.weight, att.to_out.weight, att.to_out.bias, att.heads, att.scale)
        if ctx.kv_planes:
            B_, H_, W_, C2 = kv.shape
            dkv_s = (ops.new_bf(kv, B_, H_, W_, C2), None)
            ops.linattn_bwd_core(kv, dctx, rvec, kmax, ksum, None, att.heads, koff=0, planes=dkv_s)
            F_.kv_backward_planes(xn, (xnb, None), dim, dkv_s, att.to_qkv.weight, dxn)
        else:
            dkv = torch.empty(kv.shape, device=kv.device, dtype=torch.float32)
            ops.linattn_bwd_core(kv, dctx, rvec, kmax, ksum, dkv, att.heads, koff=0)
            F_.kv_backward(xn, dim, dkv, att.to_qkv.weight, dxn)
    else:
        x, xn, mean, rstd, qkv, o, cx, cxs, kmax, ksum = ctx.saved_tensors
        dim, HD = x.shape[-1], att.heads * 32
        if ctx.fused:
            B, H, W, _ = qkv.shape
            dqkv = torch.empty((B, H, W, 3 * HD), device=qkv.device, dtype=torch.float32)
            dctx, rvec = ops.linattn_project_bwd(qkv, dyf, o, cx, cxs, att.to_out.weight, att.to_out.bias, dqkv, att.heads, att.scale)
            ops.linattn_bwd_core(qkv, dctx, rvec, kmax, ksum, dqkv, att.heads)
        else:
            do = F_.conv_backward(o, HD, dyf, att.to_out.weight, att.to_out.bias)
            dqkv = ops.linattn_bwd(qkv, do, cx, cxs, kmax, ksum, att.heads, att.scale)
        dxn = F_.conv_backward(xn, dim, dqkv, att.to_qkv.weight, None)
    dx = ops.layernorm_bwd_bf(dxn, x, norm.g, norm.b, mean, rstd, add=dy)     # + the residual branch, same pass; bf16 out
    _done(ctx)
    return None, dx, None, None

"""Functional (non-autograd) wrappers over the colddiff C ABI operating on torch HIP tensors.

Feature maps are NHWC views ``[B, H, W, C]`` with unit channel stride and an arbitrary pixel pitch
(``t.stride(-2)``), so a channel slice of a wider buffer is a valid operand.  Tensors whose logical
channel count is not a multiple of 4 are stored padded to 4 with zero pad channels.
"""
import os
import weakref

import torch

from . import convdesc as cd
from . import runtime as rt
from .runtime import P, r4


def ld_of(t):
    if t.is_contiguous():                      # (the common case, ~1300 calls per training step: no stride arithmetic)
        return t.shape[-1]
    assert t.stride(-1) == 1, "feature maps need unit channel stride"
    if t.dim() == 4:
        B, H, W, _ = t.shape
        ld = t.stride(2) if W > 1 else (t.stride(1) if H > 1 else t.stride(0))
        assert (W == 1 or t.stride(2) == ld) and (H == 1 or t.stride(1) == W * ld) and (B == 1 or t.stride(0) == H * W * ld), \
            "feature map is not pixel-contiguous: %s %s" % (tuple(t.shape), t.stride())
        return ld
    return t.stride(-2) if t.shape[-2] > 1 else t.shape[-1]


def shape_only(B, H, W, C):
    """A [B,H,W,r4(C)] tensor WITHOUT storage (meta device): stands for a feature map whose fp32 copy is not
    materialised because every consumer reads its bf16 hi/lo planes.  Its data_ptr() is 0, so any kernel handed
    it as data fails its null check instead of reading garbage."""
    return torch.empty((B, H, W, r4(C)), device="meta", dtype=torch.float32)


def new_feat(ref, B, H, W, C, zero=False):
    f = torch.zeros if (zero or C % 4) else torch.empty
    return f((B, H, W, r4(C)), device=ref.device, dtype=torch.float32)


# ---------------------------------------------------------------------------------------------------
# packed-weight cache
# ---------------------------------------------------------------------------------------------------
_pack_cache = {}          # (id(param), kind) -> [key, out, weakref, geom, last_used_epoch]
_pack_tables = {}         # device -> (signature, device table tensor, nentries, nblocks): the cdf_pack_many descriptor table
_PACK_ALL = __import__("os").environ.get("CDF_PACK_ALL", "1") != "0"
# smallest pixel count whose weight gradient runs on the bf16 matrix cores (below, the fp32-MFMA kernel; 2048 until the end of round 2:
# the 4 x 4-pixel level of the 32 x 32 configurations is M = 512 with 512 -> 1024 channels, 62 -> 28 us per launch)
WGRAD_SP_MIN_M = int(__import__("os").environ.get("CDF_WGRAD_SP_MIN_M", "512"))
_SP_WGRAD_MIN_PIX = int(__import__("os").environ.get("CDF_SP_WGRAD_MIN_PIX", "128"))   # smallest pixel count per split of the in-kernel-split weight gradient
_ATTN_KV_FUSED = __import__("os").environ.get("CDF_ATTN_KV_FUSED", "1") != "0"    # one-kernel k / v attention backward (k_attn.hip)


def _pack(src, T, R, C, s_t, s_r, s_c):
    dst = torch.empty((T, R, r4(C)), device=src.device, dtype=torch.float32)
    rt.lib().cdf_pack_weight(P(src), P(dst), T, R, C, r4(C), s_t, s_r, s_c, rt.stream(src))
    return dst


def _pack_geom(w, kind):
    """(T, R, C, ldc, s_t, s_r, s_c, bf16, off) of dst[t][r][c] = src[off + c*s_c + r*s_r + t*s_t] for a GEMM layout of parameter w, or None."""
    g = _pack_geom0(w, kind)
    return g if (g is None or len(g) == 9) else g + (0,)


def _pack_geom0(w, kind):
    if kind.startswith("kv_"):
        # the k | v rows of a LinearAttention to_qkv weight [3 HD, Ci, 1, 1] as a conv of its own (rows HD .. 3 HD)
        Co3, Ci = w.shape[0], w.shape[1]
        HD = Co3 // 3
        N2, off = 2 * HD, HD * Ci
        if kind == "kv_fwd":
            return (1, Ci, N2, r4(N2), 1, 1, Ci, False, off)
        if kind == "kv_dgrad":
            return (1, N2, Ci, r4(Ci), 1, Ci, 1, False, off)
        if kind == "kv_fwd_sp":
            return (1, N2, Ci, (Ci + 31) // 32 * 32, 1, Ci, 1, True, off)
        if kind == "kv_dgrad_sp":
            return (1, Ci, N2, (N2 + 31) // 32 * 32, 1, 1, Ci, True, off)
        return None
    if kind in ("conv_fwd", "conv_dgrad"):
        Co, Ci, KH, KW = w.shape
        KK = KH * KW
        return (KK, Ci, Co, r4(Co), 1, KK, Ci * KK, False) if kind == "conv_fwd" else (KK, Co, Ci, r4(Ci), 1, Ci * KK, KK, False)
    if kind in ("convT_fwd", "convT_dgrad"):
        Ci, Co, KH, KW = w.shape
        KK = KH * KW
        return (KK, Ci, Co, r4(Co), 1, Co * KK, KK, False) if kind == "convT_fwd" else (KK, Co, Ci, r4(Ci), 1, KK, Co * KK, False)
    if kind in ("conv_fwd_sp", "conv_dgrad_sp", "convT_fwd_sp", "convT_dgrad_sp"):
        # bf16 hi/lo planes [KK][N][ldk] (K contiguous) for the split-precision kernels
        if kind.startswith("convT"):
            Ci, Co, KH, KW = w.shape
            KK = KH * KW
            # (N, K, s_n, s_k): fwd: N=Cout (stride KK), K=Cin (stride Co*KK); dgrad: N=Cin (stride Co*KK), K=Cout (stride KK)
            N, K, s_n, s_k = (Co, Ci, KK, Co * KK) if kind == "convT_fwd_sp" else (Ci, Co, Co * KK, KK)
        else:
            Co, Ci, KH, KW = w.shape
            KK = KH * KW
            N, K, s_n, s_k = (Co, Ci, Ci * KK, KK) if kind == "conv_fwd_sp" else (Ci, Co, KK, Ci * KK)
        return (KK, N, K, (K + 31) // 32 * 32, 1, s_n, s_k, True)
    if kind == "cin_z":
        # [1][Cout][9 Cin]: the 3x3 weight as the [K = Cout][N = Cin 9] matrix of the two-stage small-Cin data gradient (conv_cin_dgrad2)
        Co, Ci, KH, KW = w.shape
        return (1, Co, Ci * KH * KW, r4(Ci * KH * KW), 0, Ci * KH * KW, 1, False)
    if kind == "dw":
        return (49, 1, w.shape[0], r4(w.shape[0]), 1, 0, 49, False)
    if kind == "lin_fwd":
        N, K = w.shape
        return (1, K, N, r4(N), 0, 1, K, False)
    return None


def _pack_one(w, kind, geom, out=None):
    """Pack one layout (own launch); reuses `out` when its buffers fit."""
    if kind == "cin4":
        # [k*k][4][r4(Cout)] for the direct <= 4-input-channel convolution kernels (rows >= Cin zero)
        Co, Ci, KH, KW = w.shape
        out = torch.empty((KH * KW, 4, r4(Co)), device=w.device, dtype=torch.float32)
        rt.lib().cdf_pack_cin4(P(w), P(out), r4(Co), Co, Ci, KH, rt.stream(w))
        return out
    if geom is None:
        raise ValueError(kind)
    T, R, C, ldc, s_t, s_r, s_c, bf16, off = geom
    src = P(w) + 4 * off
    if bf16:
        want_lo = rt.precision == "bf16x3"                   # bf16 mode: hi plane only
        if out is None or (out[1] is not None) != want_lo:
            out = (torch.empty((T, R, ldc), device=w.device, dtype=torch.int16),
                   torch.empty((T, R, ldc), device=w.device, dtype=torch.int16) if want_lo else None)
        rt.lib().cdf_pack_weight_bf16(src, P(out[0]), P(out[1]), T, R, C, ldc, s_t, s_r, s_c, rt.stream(w))
        return out
    if out is None:
        out = torch.empty((T, R, ldc), device=w.device, dtype=torch.float32)
    rt.lib().cdf_pack_weight(src, P(out), T, R, C, ldc, s_t, s_r, s_c, rt.stream(w))
    return out


def _repack_all(device, epoch_used):
    """The weights changed (optimizer step): rewrite, IN PLACE and in ONE launch, every cached layout on `device` that the previous
    weights epoch used.  Returns the number of layouts refreshed."""
    import struct
    L = rt.lib()
    ents = []
    for slot, ent in list(_pack_cache.items()):
        key, out, ref, geom, used = ent
        p = ref()
        if p is None or geom is None or used != epoch_used or key[2] != epoch_used or p.device != device:
            continue
        if key[0] != p.data_ptr() or key[1] != p._version or key[4] != rt.precision:
            continue                                         # re-homed / rewritten through torch / other arithmetic mode: individual path
        ents.append((slot, ent, p))
    if len(ents) < 2:
        return 0
    recs, first, sig = [], 0, []
    for slot, ent, p in ents:
        T, R, C, ldc, s_t, s_r, s_c, bf16, off = ent[3]
        out = ent[1]
        d0, d1 = (P(out[0]), P(out[1])) if bf16 else (P(out), 0)
        recs.append(struct.pack("<QQQqqqiiiiii", p.data_ptr() + 4 * off, d0, d1, s_t, s_r, s_c, T, R, C, ldc, 1 if bf16 else 0, first))
        sig.append((p.data_ptr() + 4 * off, d0, d1))
        first += L.cdf_pack_blocks(T, R, ldc, s_t)
    sig = tuple(sig)
    tab = _pack_tables.get(device)
    if tab is None or tab[0] != sig:
        blob = b"".join(recs)
        assert len(blob) == len(recs) * L.cdf_pack_entry_bytes()
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
        tab = _pack_tables[device] = (sig, t, len(recs), first)
    L.cdf_pack_many(P(tab[1]), tab[2], tab[3], rt.stream(tab[1]))
    for slot, ent, p in ents:
        k = ent[0]
        ent[0] = (k[0], k[1], rt.weights_epoch, k[3], k[4])
    return len(ents)


def packed(param, kind):
    """GEMM-layout copy of a parameter, cached until the parameter changes.

    kinds: conv_fwd  [KK][Cin][Cout]   conv_dgrad  [KK][Cout][Cin]   (weight [Cout,Cin,KH,KW])
           convT_fwd [KK][Cin][Cout]   convT_dgrad [KK][Cout][Cin]   (weight [Cin,Cout,KH,KW])
           *_sp      bf16 hi [/ lo] planes [KK][N][ldk] for the split-precision / bf16 kernels
           kv_*      the same four layouts of rows HD .. 3 HD of a to_qkv weight [3 HD, C, 1, 1] (the k | v projection alone)
           dw        [49][C]                                          (weight [C,1,7,7])
           lin_fwd   [1][K][N]                                        (weight [N,K])
    When only the weights epoch moved (an optimizer step rewrote the arena), the first request re-packs EVERY layout the previous
    epoch used in one cdf_pack_many launch, in place.
    """
    key = (param.data_ptr(), param._version, rt.weights_epoch, kind, rt.precision)
    slot = (id(param), kind)
    hit = _pack_cache.get(slot)
    if hit is not None and hit[2]() is param:                # the weakref guards against id()/address reuse by a new tensor
        if hit[0] == key:
            hit[4] = rt.weights_epoch
            return hit[1]
        k = hit[0]
        if (_PACK_ALL and hit[3] is not None and k[0] == key[0] and k[1] == key[1] and k[4] == key[4] and param.device.type != "meta"
                and hit[4] == k[2]):
            if _repack_all(param.device, k[2]) and hit[0] == key:
                hit[4] = rt.weights_epoch
                return hit[1]
    w = param.detach()
    geom = _pack_geom(w, kind) if kind != "cin4" else None
    reuse = hit[1] if (hit is not None and hit[2]() is param and hit[3] == geom and geom is not None) else None
    out = _pack_one(w, kind, geom, reuse)
    _pack_cache[slot] = [key, out, weakref.ref(param, lambda _r, slot=slot: _pack_cache.pop(slot, None)), geom, rt.weights_epoch]
    return out


def padded_vec(v, n):
    """1-D parameter zero-padded to n entries (for C % 4 != 0 biases)."""
    if v is None or v.shape[-1] == n:
        return v
    out = torch.zeros(v.shape[:-1] + (n,), device=v.device, dtype=torch.float32)
    out[..., :v.shape[-1]] = v.detach()
    return out


def grad_of(p):
    """The gradient buffer kernels accumulate into (allocated on first use)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


# ---------------------------------------------------------------------------------------------------
# conv / linear primitives
# ---------------------------------------------------------------------------------------------------
_zero_pages = {}


def zero_page(device):
    z = _zero_pages.get(str(device))
    if z is None:
        z = torch.zeros(64, device=device, dtype=torch.float32)
        _zero_pages[str(device)] = z
    return z


def split_bf16(x):
    """fp32 feature map [.., C] (any pixel pitch) -> (hi, lo) bf16 planes, contiguous, pitch roundup8(C).
    In "bf16" mode the operands are single bf16 values: lo is None and is never written or read."""
    C = x.shape[-1]
    ld = (C + 7) // 8 * 8
    f = torch.zeros if ld != C else torch.empty
    hi = f(x.shape[:-1] + (ld,), device=x.device, dtype=torch.int16)
    lo = f(x.shape[:-1] + (ld,), device=x.device, dtype=torch.int16) if rt.precision == "bf16x3" else None
    rt.lib().cdf_split_bf16(P(x), ld_of(x), P(hi), P(lo), ld, x.numel() // C, C, rt.stream(x))
    return hi, lo


# A backward kernel whose result feeds the next block's GEMMs can write it as bf16 planes too (cdf_dwconv7_planes, cdf_layernorm_c_bwd_planes):
# the planes ride on the gradient tensor through the autograd engine (`_cdf_planes`, with the tensor's version counter -- the engine sums
# gradients IN PLACE when a tensor has several consumers, which bumps it) and the consumer takes them instead of launching cdf_split_bf16.
GRAD_PLANES = os.environ.get("CDF_GRAD_PLANES", "1") != "0"


def want_grad_planes(C):
    return GRAD_PLANES and rt.precision == "bf16x3" and C % 8 == 0 and C >= 64


def attach_planes(t, planes):
    t._cdf_planes = (planes, t._version)
    return t


def grad_planes(t):
    """The (hi, lo) planes a producer attached to this very tensor, if they still describe its contents."""
    rec = getattr(t, "_cdf_planes", None)
    if rec is None or rec[1] != t._version:
        return None
    hi = rec[0][0]
    if hi.shape != t.shape or hi.device != t.device or not t.is_contiguous() or rec[0][1] is None:
        return None
    return rec[0]


def split_or_planes(t):
    return grad_planes(t) or split_bf16(t)


def split_planes_like(ref, B, H, W, C):
    """Empty (hi, lo) bf16 planes for a [B,H,W,C] feature map (pitch roundup8(C), padding zeroed) on ref's device."""
    ld = (C + 7) // 8 * 8
    f = torch.zeros if ld != C else torch.empty
    return (f((B, H, W, ld), device=ref.device, dtype=torch.int16),
            f((B, H, W, ld), device=ref.device, dtype=torch.int16) if rt.precision == "bf16x3" else None)


def conv_gemm_presplit(plan, xs, Cin, wp, Cout, y=None, bias=None, sbias=None, res=None, pre=None, mul=None, act=0, mul_mode=0,
                       accumulate=0, split_out=False, planes_only=False, pre_grad=False):
    """conv_gemm with the activation already split into bf16 hi/lo planes (xs) and (hi, lo) packed weights (wp).
    split_out: also return the output's own (hi, lo) planes, written by the same epilogue (fused cdf_split_bf16).
    planes_only (with split_out): do not materialise the fp32 output at all (a shape_only stand-in is returned)."""
    hi, lo = xs
    B = hi.shape[0]
    planes_only = planes_only and split_out and Cout % 4 == 0 and y is None and not accumulate
    ys = split_planes_like(hi, B, plan.OH, plan.OW, Cout) if (split_out and Cout % 4 == 0) else None
    if y is None:
        y = shape_only(B, plan.OH, plan.OW, Cout) if planes_only else new_feat(hi, B, plan.OH, plan.OW, Cout)
    ldv = lambda t: 0 if t is None else ld_of(t)
    # grids far below one tile per CU (a few hundred pixels): split-K through a workspace (include/colddiff.h)
    M = B * plan.QH * plan.QW
    ws, nws = None, 0
    if M <= 4096 and hi.device.type != "meta":
        ks = rt.lib().cdf_conv_gemm_bf16x_ksplit(M, Cout, plan.nphase, plan.desc[2], rt.tune_ptr())
        if ks > 1:
            nws = ks * M * r4(Cout)
            ws = torch.empty((nws,), device=hi.device, dtype=torch.float32)
    if pre_grad:
        # `pre` receives GELU'(v) out of the activation's own erf / exp evaluation (the backward multiplies by it: mul_mode 3)
        assert pre is not None and act in (1, 2)
        rt.lib().cdf_conv_gemm_bf16x_io(P(hi), P(lo), hi.shape[-1], P(zero_page(hi.device)), P(wp[0]), P(wp[1]), wp[0].shape[-1], P(y), ld_of(y),
                                        B, plan.H, plan.W, Cin, plan.OH, plan.OW, Cout, plan.QH, plan.QW, plan.os, plan.istride, plan.nphase,
                                        plan.desc, P(bias), P(sbias), 0 if sbias is None else sbias.stride(0), P(res), ldv(res), P(pre),
                                        ldv(pre), P(mul), ldv(mul), act, mul_mode, accumulate, IO_PRE_GRAD, P(ys[0]) if ys else 0,
                                        P(ys[1]) if ys else 0, ys[0].shape[-1] if ys else 0, P(ws), nws, rt.tune_ptr(), rt.stream(hi))
    else:
        rt.lib().cdf_conv_gemm_bf16x(P(hi), P(lo), hi.shape[-1], P(zero_page(hi.device)), P(wp[0]), P(wp[1]), wp[0].shape[-1], P(y), ld_of(y),
                                     B, plan.H, plan.W, Cin, plan.OH, plan.OW, Cout, plan.QH, plan.QW, plan.os, plan.istride, plan.nphase,
                                     plan.desc, P(bias), P(sbias), 0 if sbias is None else sbias.stride(0), P(res), ldv(res), P(pre),
                                     ldv(pre), P(mul), ldv(mul), act, mul_mode, accumulate, P(ys[0]) if ys else 0, P(ys[1]) if ys else 0,
                                     ys[0].shape[-1] if ys else 0, P(ws), nws, rt.tune_ptr(), rt.stream(hi))
    if split_out:
        return y, (ys if ys is not None else split_bf16(y))
    return y


def conv_dgrad_lnbwd_ok(plan, B, Cin, Cout):
    """The data-gradient GEMM can run the LayerNorm backward in its epilogue (include/colddiff.h: cdf_conv_gemm_bf16x_lnbwd)."""
    return bool(plan.os == 1 and plan.istride == 1 and plan.OH == plan.H and plan.OW == plan.W and
                rt.lib().cdf_conv_gemm_bf16x_lnbwd_ok(B, plan.H, plan.W, Cin, Cout, plan.nphase, plan.desc[2]))


def conv_dgrad_lnbwd(plan, dys, Cin, wp, Cout, h, g_param, b_param, mean, rstd):
    """dh = LayerNorm'(h)[conv_dgrad(dy)] in one launch: dys = dy's (hi, lo) planes (Cin channels), wp = the (hi, lo) data-gradient
    packing, Cout = the LayerNorm's width; accumulates the g / b gradients into the parameters."""
    hi, lo = dys
    B = hi.shape[0]
    dh = torch.empty((B, plan.H, plan.W, Cout), device=hi.device, dtype=torch.float32)
    M = B * plan.H * plan.W
    part = torch.empty((M // 64 * 2 * Cout,), device=hi.device, dtype=torch.float32)
    rt.lib().cdf_conv_gemm_bf16x_lnbwd(P(hi), P(lo), hi.shape[-1], P(zero_page(hi.device)), P(wp[0]), P(wp[1]), wp[0].shape[-1],
                                       B, plan.H, plan.W, Cin, Cout, plan.desc, P(h), ld_of(h), P(mean), P(rstd), P(g_param), P(dh), Cout,
                                       P(grad_of(g_param)), P(grad_of(b_param)), P(part), rt.tune_ptr(), rt.stream(hi))
    return dh


def cin4_ok(x, Cin, weight, stride=1):
    """The direct small-Cin kernels apply: <= 4 input channels held with pitch 4, k in {1, 3}, stride 1, Cout = 4 * 2^j <= 256."""
    Cout, _, k, k2 = weight.shape
    lp = Cout // 4
    return (Cin <= 4 and x.dim() == 4 and x.shape[-1] == 4 and x.is_contiguous() and stride == 1 and k == k2 and k in (1, 3)
            and Cout % 4 == 0 and 1 <= lp <= 64 and (lp & (lp - 1)) == 0)


def conv_cin4_fwd(x, weight, bias, act=0, pre=None, split_out=False, planes_only=False):
    """y = act(conv(x) + bias) for x [B,H,W,4]; optionally the pre-activation and the bf16 planes of y."""
    B, H, W, _ = x.shape
    Cout, _, k, _ = weight.shape
    planes_only = planes_only and split_out
    y = shape_only(B, H, W, Cout) if planes_only else torch.empty((B, H, W, Cout), device=x.device, dtype=torch.float32)
    ys = split_planes_like(x, B, H, W, Cout) if split_out else None
    wp = packed(weight, "cin4")
    rt.lib().cdf_conv_cin4_fwd(P(x), P(wp), wp.shape[-1], P(bias), P(y), Cout, P(pre), 0 if pre is None else ld_of(pre),
                               P(ys[0]) if ys else 0, P(ys[1]) if ys else 0, ys[0].shape[-1] if ys else 0, B, H, W, Cout, k, act, rt.stream(x))
    return (y, ys) if split_out else y


def conv_cin4_bwd(x, dy, weight, bias, need_dx, dx=None, dx_accumulate=0):
    """Accumulates weight / bias gradients; returns dx [B,H,W,4] if need_dx."""
    L, S = rt.lib(), rt.stream(x)
    B, H, W, _ = x.shape
    Cout, Cin, k, _ = weight.shape
    KK = k * k
    nch = L.cdf_conv_cin4_nchunk(B * H * W)
    part = torch.empty((nch, KK * Cin, Cout), device=x.device, dtype=torch.float32)
    bsum = torch.empty((nch, Cout), device=x.device, dtype=torch.float32) if bias is not None else None
    L.cdf_conv_cin4_wgrad(P(x), P(dy), ld_of(dy), P(part), P(bsum), B, H, W, Cin, Cout, k, S)
    if bias is not None:
        L.cdf_unpack_reduce_bias(P(part), P(grad_of(weight)), nch, KK, Cin, Cout, Cout, 1, KK, Cin * KK, P(bsum), P(grad_of(bias)), Cout, 1, 1, S)
    else:
        L.cdf_unpack_reduce(P(part), P(grad_of(weight)), nch, KK, Cin, Cout, Cout, 1, KK, Cin * KK, 1, 1, S)
    if not need_dx:
        return None
    if dx is None:
        dx = torch.empty((B, H, W, 4), device=x.device, dtype=torch.float32)
        dx_accumulate = 0
    wp = packed(weight, "cin4")
    L.cdf_conv_cin4_dgrad(P(dy), ld_of(dy), P(wp), wp.shape[-1], P(dx), B, H, W, Cout, k, dx_accumulate, S)
    return dx


def conv_cin_dgrad2(dy, Cout, weight):
    """Data gradient of a 3x3 'same' conv with <= 4 input channels in two stages (k_conv_cin4.hip: tapsum3_kernel): the sum over
    the Cout channels first, per pixel, as a 1x1 GEMM onto the 9 Cin (c, ky, kx) columns (dy is read ONCE instead of nine times
    through a K = 9 Cout gather-GEMM with 3 useful output columns), then the nine shifted 3-vectors are added.  Returns dx [B,H,W,4]."""
    B, H, W, _ = dy.shape
    Cin = weight.shape[1]
    plan = cd.conv_fwd(H, W, 1, 1, 1, 0, 0, 0, 0)
    z = conv_gemm(plan, dy, Cout, packed(weight, "cin_z"), 9 * Cin)
    dx = torch.empty((B, H, W, 4), device=dy.device, dtype=torch.float32)
    rt.lib().cdf_conv_cin4_tapsum3(P(z), ld_of(z), P(dx), B, H, W, Cin, 0, rt.stream(dy))
    return dx


def conv_gemm(plan, x, Cin, wp, Cout, y=None, bias=None, sbias=None, res=None, pre=None, mul=None, act=0, mul_mode=0,
              accumulate=0):
    B = x.shape[0]
    if y is None:
        y = new_feat(x, B, plan.OH, plan.OW, Cout)
    ldv = lambda t: 0 if t is None else ld_of(t)
    if isinstance(wp, tuple):                     # (hi, lo) bf16 planes -> split-precision kernel
        hi, lo = wp
        rt.lib().cdf_conv_gemm_bf16(P(x), ld_of(x), P(hi), P(lo), hi.shape[-1], P(y), ld_of(y), B, plan.H, plan.W, Cin, plan.OH,
                                    plan.OW, Cout, plan.QH, plan.QW, plan.os, plan.istride, plan.nphase, plan.desc, P(bias),
                                    P(sbias), 0 if sbias is None else sbias.stride(0), P(res), ldv(res), P(pre), ldv(pre), P(mul),
                                    ldv(mul), act, mul_mode, accumulate, 3 if rt.precision == "bf16x3" else 1, rt.stream(x))
        return y
    rt.lib().cdf_conv_gemm(P(x), ld_of(x), P(wp), wp.shape[-1], P(y), ld_of(y), B, plan.H, plan.W, Cin, plan.OH, plan.OW, Cout,
                           plan.QH, plan.QW, plan.os, plan.istride, plan.nphase, plan.desc, P(bias), P(sbias),
                           0 if sbias is None else sbias.stride(0), P(res), ldv(res), P(pre), ldv(pre), P(mul), ldv(mul),
                           act, mul_mode, accumulate, 0, 1, 0, 0, 0, 1, 0, 0, 0, rt.stream(x))
    return y


def best_nsplit(tiles, slots, max_ns, cap=256):
    """Split-K factor: time ~ ceil(tiles*ns / slots) / ns (equal-length block rounds per unit of work).
    The SMALLEST ns within 3 % of the optimum wins: every split costs a partial-sum slab that is written and
    read back by the reduction (at 128x128 pixels ns = 227 instead of 56 meant 4x the slab traffic for 1 %)."""
    hi = max(1, min(max_ns, cap))
    cost = [-(-tiles * ns // slots) / ns for ns in range(1, hi + 1)]
    best = min(cost)
    for ns, c in enumerate(cost, 1):
        if c <= best * 1.03:
            return ns
    return hi


def wgrad_into(gparam, wplan, xa, CA, xb, CB, s_t, s_r, s_c, gbias=None, xa_s=None, xb_s=None):
    """gparam[c*s_c + r*s_r + t*s_t] += sum_m xa[pixA(m,t)][r] * xb[pixB(m,t)][c]
    gbias (optional, only when xb = dY is visited row by row): gbias[c] += sum_m xb[m][c], fused into the same pass."""
    L = rt.lib()
    B = xa.shape[0]
    M = B * wplan.QH * wplan.QW
    ldo = r4(CB)
    if xa_s is not None and xb_s is not None and CA >= 64 and CB >= 64 and M >= WGRAD_SP_MIN_M:
        # both operands already split into bf16 hi/lo planes: copy + MFMA only (64-wide tiles for 64-channel sides);
        # xa / xb themselves may be shape-only stand-ins here
        dev = xa_s[0].device
        ntap_blocks = (wplan.ntaps + 1) // 2 if (CA <= 64 < CB and wplan.same_b) else wplan.ntaps    # two taps per tile there
        slots = 512
        if L.cdf_conv_wgrad_bf16x_is_row3(wplan.QH, wplan.QW, CA, CB, wplan.ntaps, 1 if wplan.same3x3 else 0, rt.tune_ptr()):
            ntap_blocks, slots = 3, 256                                # one block per row of taps, one 512-thread block per CU
        tiles = (1 if CA <= 64 else (CA + 127) // 128) * (1 if CB <= 64 else (CB + 127) // 128) * ntap_blocks
        ns = best_nsplit(tiles, slots, M // 512)
        ws = torch.empty((ns, wplan.ntaps, CA, ldo), device=dev, dtype=torch.float32)
        S = rt.stream(xa_s[0])
        bsum = torch.empty((ns, ldo), device=dev, dtype=torch.float32) if gbias is not None else None
        L.cdf_conv_wgrad_bf16x(P(xa_s[0]), P(xa_s[1]), ld_of(xa_s[0]), P(xb_s[0]), P(xb_s[1]), ld_of(xb_s[0]), P(zero_page(dev)),
                               P(ws), ldo, B, wplan.QH, wplan.QW, wplan.HA, wplan.WA, wplan.sa, wplan.HB, wplan.WB, wplan.sb, CA, CB,
                               wplan.ntaps, wplan.desc, ns, P(bsum), rt.tune_ptr(), S)
        _reduce_slabs(L, ws, gparam, ns, wplan.ntaps, CA, CB, ldo, s_t, s_r, s_c, bsum, gbias, S)
        return
    # (bf16 activation storage: the operands of a layer too thin for the plane kernel above are widened for the kernels below)
    xa, xb = f32_of(xa, CA), f32_of(xb, CB)
    if rt.precision != "f32" and CA >= 128 and CB >= 128 and M >= WGRAD_SP_MIN_M:   # full 128x128 tiles only; thinner layers are faster on the fp32 kernel
        # split-precision bf16 MFMA kernel: 128x128 tiles, resident twice per CU (512 slots)
        tiles = ((CA + 127) // 128) * ((CB + 127) // 128) * wplan.ntaps
        # (a block of this kernel walks its pixels in dependent 32-pixel steps of ~4 us each -- load, split, transposing LDS pass, MFMA --
        #  so a thin layer at 16 x 16 pixels took 70 us for 1 GFLOP with M // 512 = 16 splits: splits down to _SP_WGRAD_MIN_PIX pixels)
        ns = best_nsplit(tiles, 512, max(1, M // _SP_WGRAD_MIN_PIX))
        ws = torch.empty((ns, wplan.ntaps, CA, ldo), device=xa.device, dtype=torch.float32)
        S = rt.stream(xa)
        bsum = torch.empty((ns, ldo), device=xa.device, dtype=torch.float32) if gbias is not None else None
        L.cdf_conv_wgrad_bf16(P(xa), ld_of(xa), P(xb), ld_of(xb), P(ws), ldo, B, wplan.QH, wplan.QW, wplan.HA, wplan.WA, wplan.sa,
                              wplan.HB, wplan.WB, wplan.sb, CA, CB, wplan.ntaps, wplan.desc, ns, P(bsum), S)
        _reduce_slabs(L, ws, gparam, ns, wplan.ntaps, CA, CB, ldo, s_t, s_r, s_c, bsum, gbias, S)
        return
    tiles_f32 = (1 if CA <= 64 else (CA + 127) // 128) * (1 if CB <= 64 else (CB + 127) // 128) * wplan.ntaps
    # (one or two output tiles -- 64 -> 3, the 64-channel k|v projection: a block keeps ONE 4 KB chunk in flight, so the bandwidth
    #  comes from filling all 1024 block slots: up to 1024 splits there, the slabs stay a few percent of the operand bytes)
    ns = best_nsplit(tiles_f32, 1024, max(1, M // 256), cap=1024 if tiles_f32 <= 2 else 256)
    ws = torch.empty((ns, wplan.ntaps, CA, ldo), device=xa.device, dtype=torch.float32)
    S = rt.stream(xa)
    bsum = torch.empty((ns, ldo), device=xa.device, dtype=torch.float32) if gbias is not None else None
    L.cdf_conv_wgrad(P(xa), ld_of(xa), P(xb), ld_of(xb), P(ws), ldo, B, wplan.QH, wplan.QW, wplan.HA, wplan.WA, wplan.sa,
                     wplan.HB, wplan.WB, wplan.sb, CA, CB, wplan.ntaps, wplan.desc, ns, 1, 0, 0, 0, P(bsum), S)
    _reduce_slabs(L, ws, gparam, ns, wplan.ntaps, CA, CB, ldo, s_t, s_r, s_c, bsum, gbias, S)


def _reduce_slabs(L, ws, gparam, ns, ntaps, CA, CB, ldo, s_t, s_r, s_c, bsum, gbias, S):
    """gparam += sum over the split-K slabs (and gbias += sum over the bias partials, in the same launch)."""
    if gbias is not None:
        L.cdf_unpack_reduce_bias(P(ws), P(gparam), ns, ntaps, CA, CB, ldo, s_t, s_r, s_c, P(bsum), P(gbias), ldo, 1, 1, S)
    else:
        L.cdf_unpack_reduce(P(ws), P(gparam), ns, ntaps, CA, CB, ldo, s_t, s_r, s_c, 1, 1, S)


def colsum_into(gvec, x, C, nseg=1):
    """gvec[seg][c] += sum over the rows of segment seg of x (x: [..., C] pitched rows)."""
    L = rt.lib()
    ld = ld_of(x)
    rows = x.numel() // x.shape[-1]
    rps = rows // nseg
    nch = L.cdf_colsum_nchunk(rps)
    ws = torch.empty((nseg * nch * C,), device=x.device, dtype=torch.float32)
    L.cdf_colsum_io(P(x), P(gvec), P(ws), nseg, rps, C, ld, gvec.stride(0) if gvec.dim() == 2 else C, 1, 1 if is_bf(x) else 0, rt.stream(x))


def copy_feat(src, C=None):
    C = src.shape[-1] if C is None else C
    dst = torch.empty(src.shape, device=src.device, dtype=torch.float32)
    rows = src.numel() // src.shape[-1]
    rt.lib().cdf_axpby(P(dst), ld_of(dst), P(src), ld_of(src), rows, src.shape[-1], 0.0, 1.0, rt.stream(src))
    return dst


def add_into(dst, src):
    rows = dst.numel() // dst.shape[-1]
    rt.lib().cdf_axpby(P(dst), ld_of(dst), P(src), ld_of(src), rows, dst.shape[-1], 1.0, 1.0, rt.stream(dst))
    return dst


# ---------------------------------------------------------------------------------------------------
# norms / depthwise / attention primitives
# ---------------------------------------------------------------------------------------------------
def layernorm_fwd(x, g, b, eps, save, split_out=False, planes_only=False):
    """split_out: also return the output's bf16 (hi, lo) planes for the conv that consumes it (fused cdf_split_bf16);
    planes_only: do not materialise the fp32 output (a shape_only stand-in is returned)."""
    B, H, W, C = x.shape
    M = B * H * W
    planes_only = planes_only and split_out
    y = shape_only(B, H, W, C) if planes_only else torch.empty((B, H, W, C), device=x.device, dtype=torch.float32)
    mean = torch.empty((M,), device=x.device, dtype=torch.float32) if save else None
    rstd = torch.empty((M,), device=x.device, dtype=torch.float32) if save else None
    ys = split_planes_like(x, B, H, W, C) if split_out else None
    rt.lib().cdf_layernorm_c_fwd(P(x), ld_of(x), P(y), C, P(g), P(b), P(mean), P(rstd), M, C, eps, P(ys[0]) if ys else 0,
                                 P(ys[1]) if ys else 0, ys[0].shape[-1] if ys else 0, rt.stream(x))
    if split_out:
        return y, mean, rstd, ys
    return y, mean, rstd


def layernorm_bwd(dy, x, g_param, b_param, mean, rstd, dx=None, add=None, planes=False):
    """returns dx (accumulating into `dx` if given; add: a second tensor added in the same pass -- the residual gradient of
    Residual(PreNorm(..)) --, into a new dx); accumulates g/b gradients into the params.
    planes: dx also as bf16 (hi, lo) planes, attached to it (attach_planes)."""
    L = rt.lib()
    B, H, W, C = x.shape
    M = B * H * W
    acc = 1 if dx is not None else 0
    assert not (acc and add is not None)
    if dx is None:
        dx = torch.empty((B, H, W, C), device=x.device, dtype=torch.float32)
    part = torch.empty((L.cdf_layernorm_blocks(M, C) * 2 * C,), device=x.device, dtype=torch.float32)
    if planes and C % 8 == 0 and dx.is_contiguous():
        ps = split_planes_like(x, B, H, W, C)
        L.cdf_layernorm_c_bwd_planes(P(dy), ld_of(dy), P(x), ld_of(x), P(g_param), P(mean), P(rstd), P(dx), ld_of(dx), P(add),
                                     0 if add is None else ld_of(add), P(grad_of(g_param)), P(grad_of(b_param)), P(part), M, C, acc, 1,
                                     P(ps[0]), P(ps[1]), C, rt.stream(x))
        return attach_planes(dx, ps)
    L.cdf_layernorm_c_bwd(P(dy), ld_of(dy), P(x), ld_of(x), P(g_param), P(mean), P(rstd), P(dx), ld_of(dx), P(add),
                          0 if add is None else ld_of(add), P(grad_of(g_param)), P(grad_of(b_param)), P(part), M, C, acc, 1, rt.stream(x))
    return dx


def dwconv7(x, wp, bias, sbias, flip=0, y=None, accumulate=0, res=None, planes=False):
    """y = dwconv(x) [+ bias + sbias] [+ old y] [+ res];  planes: y also as bf16 (hi, lo) planes, attached to it (attach_planes)"""
    B, H, W, Cp = x.shape
    if y is None:
        y = torch.empty((B, H, W, Cp), device=x.device, dtype=torch.float32)
    if planes and Cp % 8 == 0 and y.is_contiguous():
        ps = split_planes_like(x, B, H, W, Cp)
        rt.lib().cdf_dwconv7_planes(P(x), ld_of(x), P(wp), wp.shape[-1], P(bias), P(sbias), 0 if sbias is None else sbias.stride(0), P(y),
                                    ld_of(y), B, H, W, Cp, flip, accumulate, P(res), 0 if res is None else ld_of(res), P(ps[0]), P(ps[1]), Cp,
                                    rt.stream(x))
        return attach_planes(y, ps)
    rt.lib().cdf_dwconv7(P(x), ld_of(x), P(wp), wp.shape[-1], P(bias), P(sbias), 0 if sbias is None else sbias.stride(0), P(y),
                         ld_of(y), B, H, W, Cp, flip, accumulate, P(res), 0 if res is None else ld_of(res), rt.stream(x))
    return y


def dwconv7_wgrad(x, dy, w_param, b_param, want_dsb, dsb_out=None):
    """Accumulates weight / bias gradients; returns the per-sample bias gradient [B, Cp] (written into `dsb_out`, a [B, Cp] view with
    any row pitch whose pad columns are already zero, when given)."""
    L = rt.lib()
    B, H, W, Cp = x.shape
    C = w_param.shape[0]
    ws = torch.empty((B * L.cdf_dwconv7_wgrad_nchunk(H) * 50 * C,), device=x.device, dtype=torch.float32)
    dsb = None
    if want_dsb:
        dsb = dsb_out if dsb_out is not None else torch.zeros((B, Cp), device=x.device, dtype=torch.float32)
    L.cdf_dwconv7_wgrad(P(x), ld_of(x), P(dy), ld_of(dy), P(grad_of(w_param)), P(grad_of(b_param)), P(dsb),
                        0 if dsb is None else dsb.stride(0), P(ws), B, H, W, C, 1, rt.stream(x))
    return dsb


def _head_gemm(x, x_off, w, out, out_off, B, n, heads, b_trans):
    """out[b, :, out_off + h*32 + j] = sum_k x[b, :, x_off + h*32 + k] * (w[b,h,k,j] | w[b,h,j,k] if b_trans),
    one batched K=32 GEMM per head (batch = B)."""
    L, S = rt.lib(), rt.stream(x)
    plan = _one_tap(n)
    ldx, ldo = ld_of(x), ld_of(out)
    # one launch: blockIdx.z = b * heads + h
    L.cdf_conv_gemm(P(x) + 4 * x_off, ldx, P(w), 32, P(out) + 4 * out_off, ldo, 1, 1, n, 32, 1, n, 32, 1, n, 1, 1, 1, plan.desc,
                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1 if b_trans else 0, B, n * ldx, heads * 1024, n * ldo, heads, 32, 1024, 32, S)


def linattn_fwd(qkv, heads, scale):
    """LinearAttention core: returns (out [B,H,W,HD], ctx, ctxs, kmax, ksum)."""
    L = rt.lib()
    B, H, W, _ = qkv.shape
    n, HD = H * W, heads * 32
    dev = qkv.device
    out = torch.empty((B, H, W, HD), device=dev, dtype=torch.float32)
    ctx = torch.empty((B, heads, 32, 32), device=dev, dtype=torch.float32)
    ctxs = torch.empty_like(ctx)
    kmax = torch.empty((B, HD), device=dev, dtype=torch.float32)
    ksum = torch.empty((B, HD), device=dev, dtype=torch.float32)
    ws = torch.empty((L.cdf_linattn_ws_floats(B, n, heads),), device=dev, dtype=torch.float32)
    L.cdf_linattn_context(P(qkv), ld_of(qkv), HD, P(ctx), P(ctxs), P(kmax), P(ksum), P(ws), B, n, heads, scale, 1, rt.stream(qkv))
    _head_gemm(qkv, 0, ctxs, out, 0, B, n, heads, False)          # out = q . (scale*ctx)
    return out, ctx, ctxs, kmax, ksum


def linattn_bwd(qkv, dout, ctx, ctxs, kmax, ksum, heads, scale):
    L = rt.lib()
    B, H, W, _ = qkv.shape
    n, HD = H * W, heads * 32
    dev, S = qkv.device, rt.stream(qkv)
    dqkv = torch.empty((B, H, W, 3 * HD), device=dev, dtype=torch.float32)
    dctx = torch.empty_like(ctx)
    rvec = torch.empty((B, HD), device=dev, dtype=torch.float32)
    ws = torch.empty((L.cdf_linattn_ws_floats(B, n, heads),), device=dev, dtype=torch.float32)
    L.cdf_linattn_dcontext(P(qkv), ld_of(qkv), P(dout), ld_of(dout), P(ctx), P(dctx), P(rvec), P(ws), B, n, heads, scale, S)
    if _ATTN_KV_FUSED and heads <= 4:
        _head_gemm(dout, 0, ctxs, dqkv, 0, B, n, heads, True)      # dq[n,d] = sum_e dout[n,e] ctxs[d,e]
        L.cdf_linattn_bwd_kv(P(qkv), ld_of(qkv), HD, P(dctx), P(rvec), P(kmax), P(ksum), P(dqkv), ld_of(dqkv), HD, B, n, heads, S)
        return dqkv
    pn = torch.empty((B, H, W, HD), device=dev, dtype=torch.float32)
    dp = torch.empty((B, H, W, HD), device=dev, dtype=torch.float32)
    L.cdf_linattn_softk(P(qkv), ld_of(qkv), P(kmax), P(ksum), P(pn), HD, B, n, heads, S)
    _head_gemm(dout, 0, ctxs, dqkv, 0, B, n, heads, True)          # dq[n,d] = sum_e dout[n,e] ctxs[d,e]
    _head_gemm(qkv, 2 * HD, dctx, dp, 0, B, n, heads, True)        # dP[n,d] = sum_e v[n,e] dctx[d,e]
    _head_gemm(pn, 0, dctx, dqkv, 2 * HD, B, n, heads, False)      # dv[n,e] = sum_d P[n,d] dctx[d,e]
    L.cdf_linattn_dk(P(pn), HD, P(dp), HD, P(rvec), P(dqkv) + 4 * HD, 3 * HD, B, n, heads, S)
    return dqkv


# -- linear attention with the output projection folded in (round 2) ---------------------------------------------------
# Residual(PreNorm(LinearAttention)) ends in  y = to_out(q . blockdiag_h(scale ctx_h)) + b + x.  Both maps are linear in q, so
# y[n] = q[n] . M_b + b + x[n]  with  M_b = blockdiag(scale ctx_b) . W_out^T  ([HD x dim], ONE small matrix per image).  The
# attention output o ([B,n,HD], 268 MB per micro-batch at 128 x 128) is never written or read, the K = 32 per-head products
# (1.3 TB/s: 32-deep GEMMs are all epilogue) and the separate to_out convolution become one K = 128 batched GEMM, and in the
# backward pass d_o disappears the same way:  dq = dy . M_b^T,  dM_b = q^T dy  (one pass over q and dy instead of the two passes
# (o, dy) for dW_out and (q, d_o) for dctx),  d(scale ctx_h) = dM_b[h] . W_h^T,  dW_h = sum_b (scale ctx_bh)^T dM_b[h].
def _headsplit_plan(heads):
    return cd.WgradPlan(1, 32, heads, 32, 1, heads, 32, 1, [(h, 0, h, 0) for h in range(heads)])


_HEADSPLIT = {}


def linattn_project(qkv, ctxs, w_out, b_out, res, heads, y=None):
    """y = q . M_b + b_out + res,  M_b = blockdiag(ctxs[b]) . W_out^T.  Returns (y, Mb).  y: optional destination (a channel slice of a
    wider buffer is fine)."""
    L, S = rt.lib(), rt.stream(qkv)
    B, H, W, _ = qkv.shape
    n, HD, dim = H * W, heads * 32, w_out.shape[0]
    wp = packed(w_out, "conv_fwd")                            # [1][HD][r4(dim)]: row h*32 + e = W_out[:, h*32 + e]
    ldw = wp.shape[-1]
    Mb = torch.empty((B, HD, ldw), device=qkv.device, dtype=torch.float32)
    # M_b[h*32 + d][c] = sum_e ctxs[b,h][d][e] W_out[c][h*32 + e]: B x heads GEMMs of 32 x 32 x dim
    L.cdf_conv_gemm(P(ctxs), 32, P(wp), ldw, P(Mb), ldw, 1, 1, 32, 32, 1, 32, dim, 1, 32, 1, 1, 1, _one_tap(32).desc,
                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, B, heads * 1024, 0, HD * ldw, heads, 1024, 32 * ldw, 32 * ldw, S)
    if y is None:
        y = new_feat(qkv, B, H, W, dim)
    ldq, ldy = ld_of(qkv), ld_of(y)
    # y[b] = q[b] . M_b (+ bias + residual): one GEMM per image, K = HD
    L.cdf_conv_gemm(P(qkv), ldq, P(Mb), ldw, P(y), ldy, 1, 1, n, HD, 1, n, dim, 1, n, 1, 1, 1, _one_tap(n).desc,
                    P(b_out), 0, 0, P(res), 0 if res is None else ld_of(res), 0, 0, 0, 0, 0, 0, 0, 0, B, n * ldq, HD * ldw, n * ldy, 1, 0, 0, 0, S)
    return y, Mb


def linattn_project_bwd(qkv, dy, Mb, ctx, ctxs, w_out, b_out, dqkv, heads, scale):
    """Backward of linattn_project: writes dq into dqkv[..., :HD], accumulates the to_out weight / bias gradients, returns
    (dctx, rvec) -- the gradient w.r.t. the (unscaled) context and rvec[b, h*32+d] = sum_e dctx*ctx for the softmax backward."""
    L, S = rt.lib(), rt.stream(qkv)
    B, H, W, _ = qkv.shape
    n, HD, dim = H * W, heads * 32, w_out.shape[0]
    ldw, ldq, lddy, lddq = Mb.shape[-1], ld_of(qkv), ld_of(dy), ld_of(dqkv)
    dev = qkv.device
    # dq[b] = dy[b] . M_b^T  (b_trans: M_b is a plain [HD][ldw >= dim] matrix)
    L.cdf_conv_gemm(P(dy), lddy, P(Mb), ldw, P(dqkv), lddq, 1, 1, n, dim, 1, n, HD, 1, n, 1, 1, 1, _one_tap(n).desc,
                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, B, n * lddy, HD * ldw, n * lddq, 1, 0, 0, 0, S)
    # dM_b = q[b]^T dy[b]: per-image weight-gradient GEMM over the n pixels, split-K slabs laid out [split][b] and summed in one pass
    wplan = cd.conv_wgrad(1, n, 1, 1, 1, 0, 0, 0, 0)
    tiles = ((HD + 127) // 128) * (1 if dim <= 64 else (dim + 127) // 128) * B
    ns = best_nsplit(tiles, 1024, max(1, n // 256))
    ws = torch.empty((ns, B, HD, ldw), device=dev, dtype=torch.float32)
    bsum = torch.empty((B * ns, ldw), device=dev, dtype=torch.float32) if b_out is not None else None    # column sums of dy as it streams by
    L.cdf_conv_wgrad(P(qkv), ldq, P(dy), lddy, P(ws), ldw, 1, 1, n, 1, n, 1, 1, n, 1, HD, dim, 1, wplan.desc, ns, B, n * ldq, n * lddy, -1, P(bsum), S)
    dMb = torch.empty((B, HD, ldw), device=dev, dtype=torch.float32)
    L.cdf_unpack_reduce(P(ws), P(dMb), ns, B, HD, dim, ldw, HD * ldw, ldw, 1, 0, 1, S)
    if b_out is not None:
        L.cdf_unpack_reduce(P(bsum), P(grad_of(b_out)), B * ns, 1, 1, dim, ldw, 0, 0, 1, 1, 1, S)
    return _linattn_out_bwd(dMb, ctx, ctxs, w_out, heads, scale)


def _linattn_out_bwd(dMb, ctx, ctxs, w_out, heads, scale):
    """From dM_b ([B][HD][ldw]): (dctx, rvec), and the to_out weight gradient accumulated."""
    L, S = rt.lib(), rt.stream(dMb)
    B, HD, ldw = dMb.shape
    dim, dev = w_out.shape[0], dMb.device
    # d(ctxs)[b,h][d][e] = sum_c dM_b[h*32 + d][c] W_out[c][h*32 + e]; dctx = scale * that; rvec = rowwise <dctx, ctx>
    wp = packed(w_out, "conv_fwd")
    raw = torch.empty((B, heads, 32, 32), device=dev, dtype=torch.float32)
    L.cdf_conv_gemm(P(dMb), ldw, P(wp), ldw, P(raw), 32, 1, 1, 32, dim, 1, 32, 32, 1, 32, 1, 1, 1, _one_tap(32).desc,
                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, B, HD * ldw, 0, heads * 1024, heads, 32 * ldw, 32 * ldw, 1024, S)
    dctx = torch.empty_like(raw)
    rvec = torch.empty((B, HD), device=dev, dtype=torch.float32)
    L.cdf_linattn_dctx_finish(P(raw), P(ctx), P(dctx), P(rvec), B * heads * 32, scale, S)
    # dW_out[c][h*32 + e] += sum_{b,d} ctxs[b,h][d][e] dM_b[h*32 + d][c]: a weight-gradient GEMM whose "taps" are the heads
    hp = _HEADSPLIT.get(heads)
    if hp is None:
        hp = _HEADSPLIT[heads] = _headsplit_plan(heads)
    nsw = max(1, min(16, (B * 32) // 64))                      # B * 32 contraction rows: a few blocks each instead of one long serial loop
    wsw = torch.empty((nsw, heads, 32, ldw), device=dev, dtype=torch.float32)
    L.cdf_conv_wgrad(P(ctxs), 32, P(dMb), ldw, P(wsw), ldw, B, 1, 32, heads, 32, 1, heads, 32, 1, 32, dim, heads, hp.desc, nsw, 1, 0, 0, 0, 0, S)
    L.cdf_unpack_reduce(P(wsw), P(grad_of(w_out)), nsw, heads, 32, dim, ldw, 32, 1, HD, 1, 1, S)
    return dctx, rvec


# -- ... and with the q projection folded in as well (round 2) ------------------------------------------------------------------
# q = xn . Wq^T is linear too (to_qkv has no bias, q is used as it is), so y[n] = xn[n] . N_b + b + x[n] with
# N_b = Wq^T . M_b ([dim x dim] per image).  Where dim <= heads*32 (the 128- and 64-pixel levels of the CelebA net) that is fewer
# FLOPs than q . M_b AND q never exists: to_qkv shrinks to the k | v projection (256 instead of 384 output channels: a third less
# written, re-read by the context pass, the data gradient and the weight gradient), the batched GEMMs stream xn / dy (dim channels)
# instead of q / dq (128), and Wq's gradient is dWq = sum_b M_b . dN_b^T with dN_b = xn_b^T dy_b.
def linattn_fold(xn, ctxs, w_qkv, w_out, b_out, res, heads, y=None):
    """y = xn . N_b + b_out + res.  Returns (y, Mb, Nb)."""
    L, S = rt.lib(), rt.stream(xn)
    B, H, W, _ = xn.shape
    n, HD, dim = H * W, heads * 32, w_out.shape[0]
    wp = packed(w_out, "conv_fwd")                            # [1][HD][r4(dim)]
    ldw = wp.shape[-1]
    Mb = torch.empty((B, HD, ldw), device=xn.device, dtype=torch.float32)
    L.cdf_conv_gemm(P(ctxs), 32, P(wp), ldw, P(Mb), ldw, 1, 1, 32, 32, 1, 32, dim, 1, 32, 1, 1, 1, _one_tap(32).desc,
                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, B, heads * 1024, 0, HD * ldw, heads, 1024, 32 * ldw, 32 * ldw, S)
    wq = packed(w_qkv, "conv_fwd")                            # [1][dim][r4(3 HD)]: row i, columns 0 .. HD-1 = Wq[:, i]
    Nb = torch.empty((B, dim, ldw), device=xn.device, dtype=torch.float32)
    L.cdf_conv_gemm(P(wq), wq.shape[-1], P(Mb), ldw, P(Nb), ldw, 1, 1, dim, HD, 1, dim, dim, 1, dim, 1, 1, 1, _one_tap(dim).desc,
                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, B, 0, HD * ldw, dim * ldw, 1, 0, 0, 0, S)
    if is_bf(res):
        # bf16 activation storage: the residual is read from the bf16 stream and the result enters it rounded once -- no fp32 copy of either
        if y is None:
            y = new_bf(xn, B, H, W, dim)
        ldx, ldy = ld_of(xn), ld_of(y)
        L.cdf_conv_gemm_io(P(xn), ldx, P(Nb), ldw, 0, ldy, 1, 1, n, dim, 1, n, dim, 1, n, 1, 1, 1, _one_tap(n).desc,
                           P(b_out), 0, 0, P(res), ld_of(res), 0, 0, 0, 0, 0, 0, 0, 0, B, n * ldx, dim * ldw, n * ldy, 1, 0, 0, 0, IO_RES, P(y), ldy, S)
        return y, Mb, Nb
    if y is None:
        y = new_feat(xn, B, H, W, dim)
    ldx, ldy = ld_of(xn), ld_of(y)
    L.cdf_conv_gemm(P(xn), ldx, P(Nb), ldw, P(y), ldy, 1, 1, n, dim, 1, n, dim, 1, n, 1, 1, 1, _one_tap(n).desc,
                    P(b_out), 0, 0, P(res), 0 if res is None else ld_of(res), 0, 0, 0, 0, 0, 0, 0, 0, B, n * ldx, dim * ldw, n * ldy, 1, 0, 0, 0, S)
    return y, Mb, Nb


def linattn_kvctx_ok(xn, dim, heads):
    """The fused k|v projection + context kernel applies (bf16 matrix-core modes, 4 heads, whole 128-pixel tiles, K a multiple of 32)."""
    B, H, W, _ = xn.shape
    return rt.precision != "f32" and heads == 4 and (H * W) % 128 == 0 and dim % 32 == 0 and 64 <= dim <= 512 and xn.device.type != "meta"


def linattn_kvctx(xn, dim, w_qkv, heads, scale):
    """kv = xn . Wkv^T ([B,H,W,2 HD]) and (ctx, ctxs, kmax, ksum) of LinearAttention in one pass over the pixels (k_conv_sp.hip:
    linattn_kvctx_kernel): k and v are written once and not read back."""
    L, S = rt.lib(), rt.stream(xn)
    B, H, W, _ = xn.shape
    n, HD = H * W, heads * 32
    dev = xn.device
    wp = packed(w_qkv, "kv_fwd_sp")                           # (hi, lo | None) planes [1][2 HD][roundup32(dim)]
    kv = torch.empty((B, H, W, 2 * HD), device=dev, dtype=torch.float32)
    P_ = L.cdf_linattn_kvctx_parts(B, n, rt.KVCTX_SLOTS)
    ws = torch.empty((B * P_ * (2 * HD + heads * 1024),), device=dev, dtype=torch.float32)
    L.cdf_linattn_kvctx(P(xn), ld_of(xn), P(wp[0]), P(wp[1]), wp[0].shape[-1], P(kv), 2 * HD, P(ws), B, n, dim, heads, rt.KVCTX_SLOTS, S)
    ctx = torch.empty((B, heads, 32, 32), device=dev, dtype=torch.float32)
    ctxs = torch.empty_like(ctx)
    kmax = torch.empty((B, HD), device=dev, dtype=torch.float32)
    ksum = torch.empty((B, HD), device=dev, dtype=torch.float32)
    L.cdf_linattn_finalize(P(ws), P_, P(ctx), P(ctxs), P(kmax), P(ksum), B, heads, scale, S)
    return kv, ctx, ctxs, kmax, ksum


def linattn_fold_bwd(xn, dy, Mb, Nb, ctx, ctxs, w_qkv, w_out, b_out, heads, scale):
    """Backward of linattn_fold w.r.t. everything but k | v: returns (dxn, dctx, rvec) with dxn = dy . N_b^T (the q path's share of the
    LayerNorm-output gradient); accumulates the gradients of Wq (rows 0 .. HD-1 of to_qkv), to_out.weight and to_out.bias."""
    L, S = rt.lib(), rt.stream(xn)
    B, H, W, _ = xn.shape
    n, HD, dim = H * W, heads * 32, w_out.shape[0]
    ldw, ldx, lddy = Mb.shape[-1], ld_of(xn), ld_of(dy)
    dev = xn.device
    dxn = new_feat(xn, B, H, W, dim)
    L.cdf_conv_gemm(P(dy), lddy, P(Nb), ldw, P(dxn), ld_of(dxn), 1, 1, n, dim, 1, n, dim, 1, n, 1, 1, 1, _one_tap(n).desc,
                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, B, n * lddy, dim * ldw, n * ld_of(dxn), 1, 0, 0, 0, S)
    # dN_b = xn[b]^T dy[b] (split-K slabs [split][b], summed in one pass); the bias gradient = column sums of dy as it streams by
    wplan = cd.conv_wgrad(1, n, 1, 1, 1, 0, 0, 0, 0)
    t1 = 1 if dim <= 64 else (dim + 127) // 128
    ns = best_nsplit(t1 * t1 * B, 1024, max(1, n // 256))
    ws = torch.empty((ns, B, dim, ldw), device=dev, dtype=torch.float32)
    bsum = torch.empty((B * ns, ldw), device=dev, dtype=torch.float32) if b_out is not None else None
    L.cdf_conv_wgrad(P(xn), ldx, P(dy), lddy, P(ws), ldw, 1, 1, n, 1, n, 1, 1, n, 1, dim, dim, 1, wplan.desc, ns, B, n * ldx, n * lddy, -1, P(bsum), S)
    dNb = torch.empty((B, dim, ldw), device=dev, dtype=torch.float32)
    L.cdf_unpack_reduce(P(ws), P(dNb), ns, B, dim, dim, ldw, dim * ldw, ldw, 1, 0, 1, S)
    if b_out is not None:
        L.cdf_unpack_reduce(P(bsum), P(grad_of(b_out)), B * ns, 1, 1, dim, ldw, 0, 0, 1, 1, 1, S)
    # dM_b = Wq . dN_b  (Wq = rows 0 .. HD-1 of the to_qkv weight, a plain [HD][dim] matrix in place)
    wq_rows = w_qkv.detach()
    dMb = torch.empty((B, HD, ldw), device=dev, dtype=torch.float32)
    L.cdf_conv_gemm(P(wq_rows), dim, P(dNb), ldw, P(dMb), ldw, 1, 1, HD, dim, 1, HD, dim, 1, HD, 1, 1, 1, _one_tap(HD).desc,
                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, B, 0, dim * ldw, HD * ldw, 1, 0, 0, 0, S)
    # dWq[hd][i] += sum_b sum_c M_b[hd][c] dN_b[i][c]
    T = torch.empty((B, HD, ldw), device=dev, dtype=torch.float32)
    L.cdf_conv_gemm(P(Mb), ldw, P(dNb), ldw, P(T), ldw, 1, 1, HD, dim, 1, HD, dim, 1, HD, 1, 1, 1, _one_tap(HD).desc,
                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, B, HD * ldw, dim * ldw, HD * ldw, 1, 0, 0, 0, S)
    L.cdf_unpack_reduce(P(T), P(grad_of(w_qkv)), B, 1, HD, dim, ldw, 0, dim, 1, 1, 1, S)
    dctx, rvec = _linattn_out_bwd(dMb, ctx, ctxs, w_out, heads, scale)
    return dxn, dctx, rvec


def linattn_bwd_core(qkv, dctx, rvec, kmax, ksum, dqkv, heads, koff=None, planes=None):
    """The softmax / v part of the attention backward given dctx and rvec (dq is already in dqkv).  koff: channel offset of k | v in
    qkv's rows and of dk | dv in dqkv's (default heads*32; 0: both are (k|v) tensors, fused kernel only).
    planes = (hi, lo | None): dk | dv are written as bf16 operand planes instead of into dqkv (koff = 0 form, fused kernel)."""
    L = rt.lib()
    B, H, W, _ = qkv.shape
    n, HD = H * W, heads * 32
    dev, S = qkv.device, rt.stream(qkv)
    koff = HD if koff is None else koff
    if planes is not None:
        assert koff == 0 and heads <= 4
        L.cdf_linattn_bwd_kv_planes(P(qkv), ld_of(qkv), 0, P(dctx), P(rvec), P(kmax), P(ksum), P(planes[0]), P(planes[1]), planes[0].shape[-1], 0,
                                    B, n, heads, S)
        return planes
    if (_ATTN_KV_FUSED or koff != HD) and heads <= 4:
        # one pass: P recomputed from k, dP and dv on the fp32 matrix cores, dk / dv written straight into dqkv (k_attn.hip)
        L.cdf_linattn_bwd_kv(P(qkv), ld_of(qkv), koff, P(dctx), P(rvec), P(kmax), P(ksum), P(dqkv), ld_of(dqkv), koff, B, n, heads, S)
        return dqkv
    if koff != HD:
        raise ValueError("the (k|v)-only attention backward needs the fused kernel (heads <= 4)")
    pn = torch.empty((B, H, W, HD), device=dev, dtype=torch.float32)
    dp = torch.empty((B, H, W, HD), device=dev, dtype=torch.float32)
    L.cdf_linattn_softk(P(qkv), ld_of(qkv), P(kmax), P(ksum), P(pn), HD, B, n, heads, S)
    _head_gemm(qkv, 2 * HD, dctx, dp, 0, B, n, heads, True)        # dP[n,d] = sum_e v[n,e] dctx[d,e]
    _head_gemm(pn, 0, dctx, dqkv, 2 * HD, B, n, heads, False)      # dv[n,e] = sum_d P[n,d] dctx[d,e]
    L.cdf_linattn_dk(P(pn), HD, P(dp), HD, P(rvec), P(dqkv) + 4 * HD, 3 * HD, B, n, heads, S)
    return dqkv


def linattn_context(qkv, heads, scale, koff=None):
    """(ctx, ctxs = scale * ctx, kmax, ksum) of LinearAttention (no output product).  koff: channel offset of k in qkv's rows
    (default heads*32: the (q|k|v) tensor; 0 for a (k|v) tensor)."""
    L = rt.lib()
    B, H, W, _ = qkv.shape
    n, HD = H * W, heads * 32
    koff = HD if koff is None else koff
    dev = qkv.device
    ctx = torch.empty((B, heads, 32, 32), device=dev, dtype=torch.float32)
    ctxs = torch.empty_like(ctx)
    kmax = torch.empty((B, HD), device=dev, dtype=torch.float32)
    ksum = torch.empty((B, HD), device=dev, dtype=torch.float32)
    ws = torch.empty((L.cdf_linattn_ws_floats(B, n, heads),), device=dev, dtype=torch.float32)
    L.cdf_linattn_context(P(qkv), ld_of(qkv), koff, P(ctx), P(ctxs), P(kmax), P(ksum), P(ws), B, n, heads, scale, 1, rt.stream(qkv))
    return ctx, ctxs, kmax, ksum


def nchw_to_nhwc(x):
    B, C, H, W = x.shape
    y = new_feat(x, B, H, W, C)
    rt.lib().cdf_nchw_to_nhwc(P(x), P(y), B, C, H * W, y.shape[-1], rt.stream(x))
    return y


def nhwc_to_nchw(x, C, add=None):
    B, H, W, _ = x.shape
    y = torch.empty((B, C, H, W), device=x.device, dtype=torch.float32)
    rt.lib().cdf_nhwc_to_nchw(P(x), P(y), P(add), B, C, H * W, ld_of(x), rt.stream(x))
    return y


# ---------------------------------------------------------------------------------------------------
# GroupNorm / batched GEMMs / resampling (the DDPM `Model` family)
# ---------------------------------------------------------------------------------------------------
def groupnorm_fwd(x, gamma, beta, groups, eps, silu, split_out=False, planes_only=False, drop=None):
    """drop = (p, seed): the dropout of ops.dropout applied to the activated output in the same pass; split_out: also return the
    output's bf16 (hi, lo) planes (fused cdf_split_bf16); planes_only: do not materialise the fp32 output (shape_only stand-in)."""
    L = rt.lib()
    B, H, W, C = x.shape
    HW = H * W
    nch = L.cdf_groupnorm_nchunk(HW)
    planes_only = planes_only and split_out
    y = shape_only(B, H, W, C) if planes_only else torch.empty((B, H, W, C), device=x.device, dtype=torch.float32)
    mean = torch.empty((B * groups,), device=x.device, dtype=torch.float32)
    rstd = torch.empty((B * groups,), device=x.device, dtype=torch.float32)
    ws = torch.empty((B * nch * 2 * C,), device=x.device, dtype=torch.float32)
    ys = split_planes_like(x, B, H, W, C) if split_out else None
    p, seed = drop if drop is not None else (0.0, 0)
    L.cdf_groupnorm_fwd_ex(P(x), ld_of(x), P(y), C, P(gamma), P(beta), P(mean), P(rstd), P(ws), B, HW, C, groups, eps, 1 if silu else 0,
                           float(p), int(seed), P(ys[0]) if ys else 0, P(ys[1]) if ys else 0, ys[0].shape[-1] if ys else 0, rt.stream(x))
    if split_out:
        return y, mean, rstd, ys
    return y, mean, rstd


def groupnorm_bwd(dy, x, gamma_p, beta_p, mean, rstd, groups, silu, dx=None, drop=None):
    """drop = (p, seed): dy is the gradient of the dropped output (groupnorm_fwd(..., drop=...)); the mask is applied in the same pass"""
    L = rt.lib()
    B, H, W, C = x.shape
    HW = H * W
    nch = L.cdf_groupnorm_nchunk(HW)
    acc = 1 if dx is not None else 0
    if dx is None:
        dx = torch.empty((B, H, W, C), device=x.device, dtype=torch.float32)
    ws = torch.empty((B * nch * 2 * C + B * 2 * C + B * groups * 2,), device=x.device, dtype=torch.float32)
    p, seed = drop if drop is not None else (0.0, 0)
    L.cdf_groupnorm_bwd_ex(P(dy), ld_of(dy), P(x), ld_of(x), P(gamma_p), P(beta_p), P(mean), P(rstd), P(dx), ld_of(dx),
                           P(grad_of(gamma_p)), P(grad_of(beta_p)), P(ws), B, HW, C, groups, 1 if silu else 0, acc, 1, float(p), int(seed), rt.stream(x))
    return dx


_ONE_TAP = None


def _one_tap(n):
    return cd.conv_fwd(1, n, 1, 1, 1, 0, 0, 0, 0)


def bgemm_nt(a, b):
    """[nb, n, K] x [nb, m, K]^T -> [nb, n, r4(m)] (valid columns :m)."""
    nb, n, K = a.shape
    m = b.shape[1]
    out = new_feat(a, nb, 1, n, m).view(nb, n, r4(m))
    plan = _one_tap(n)
    rt.lib().cdf_conv_gemm(P(a), a.stride(1), P(b), b.stride(1), P(out), r4(m), 1, 1, n, K, 1, n, m, 1, n, 1, 1, 1, plan.desc,
                           0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, nb, a.stride(0), b.stride(0), n * r4(m), 1, 0, 0, 0, rt.stream(a))
    return out


def bgemm_nn(a, b, K=None):
    """[nb, n, K] x [nb, K, m] -> [nb, n, r4(m)]; b is [K][m] row-major with pitch b.stride(1)."""
    nb, n = a.shape[0], a.shape[1]
    K = a.shape[2] if K is None else K
    m = b.shape[2]
    out = new_feat(a, nb, 1, n, m).view(nb, n, r4(m))
    plan = _one_tap(n)
    rt.lib().cdf_conv_gemm(P(a), a.stride(1), P(b), b.stride(1), P(out), r4(m), 1, 1, n, K, 1, n, m, 1, n, 1, 1, 1, plan.desc,
                           0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, nb, a.stride(0), b.stride(0), n * r4(m), 1, 0, 0, 0, rt.stream(a))
    return out


def bgemm_tn(a, b, CA=None):
    """[nb, K, n]^T x [nb, K, m] -> [nb, n(:CA), r4(m)]  (reduction over the row index)."""
    nb, K = a.shape[0], a.shape[1]
    CA = a.shape[2] if CA is None else CA
    m = b.shape[2]
    out = torch.empty((nb, CA, r4(m)), device=a.device, dtype=torch.float32)
    tap = cd.conv_wgrad(1, K, 1, 1, 1, 0, 0, 0, 0)
    rt.lib().cdf_conv_wgrad(P(a), a.stride(1), P(b), b.stride(1), P(out), r4(m), 1, 1, K, 1, K, 1, 1, K, 1, CA, m, 1, tap.desc, 1, nb,
                            a.stride(0), b.stride(0), CA * r4(m), 0, rt.stream(a))
    return out


def softmax_rows(s, n, scale):
    p = torch.empty_like(s)
    rows = s.numel() // s.shape[-1]
    rt.lib().cdf_softmax_rows_fwd(P(s), P(p), rows, n, s.shape[-1], scale, rt.stream(s))
    return p


def softmax_rows_bwd(p, dp, n, scale):
    ds = torch.zeros_like(p) if p.shape[-1] != n else torch.empty_like(p)
    rows = p.numel() // p.shape[-1]
    rt.lib().cdf_softmax_rows_bwd(P(p), P(dp), P(ds), rows, n, p.shape[-1], scale, rt.stream(p))
    return ds


def upsample2(x, out=None):
    B, H, W, C = x.shape
    y = torch.empty((B, 2 * H, 2 * W, C), device=x.device, dtype=torch.float32) if out is None else out
    rt.lib().cdf_upsample2(P(x), ld_of(x), P(y), ld_of(y), B, H, W, C, rt.stream(x))
    return y


def upsample2_bwd(dy, dx=None):
    B, H2, W2, C = dy.shape
    acc = 1 if dx is not None else 0
    if dx is None:
        dx = torch.empty((B, H2 // 2, W2 // 2, C), device=dy.device, dtype=torch.float32)
    rt.lib().cdf_upsample2_bwd(P(dy), ld_of(dy), P(dx), ld_of(dx), B, H2 // 2, W2 // 2, C, acc, rt.stream(dy))
    return dx


def dropout(x, p, seed):
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    rows = x.numel() // x.shape[-1]
    rt.lib().cdf_dropout(P(x), ld_of(x), P(y), x.shape[-1], rows, x.shape[-1], p, seed, rt.stream(x))
    return y


def colsum_new(x, C, nseg):
    """[nseg, r4(C)] column sums per segment (fresh tensor)."""
    out = torch.zeros((nseg, r4(C)), device=x.device, dtype=torch.float32)
    colsum_into(out, x, C, nseg)
    return out


# ---------------------------------------------------------------------------------------------------
# bf16 ACTIVATION STORAGE ("bf16" arithmetic mode, BASELINE configs 3 / 5): feature maps between kernels are torch.bfloat16 tensors
# [B, H, W, C] (C % 8 == 0, unit channel stride, any pixel pitch) -- ONE plane, the GEMMs' "hi" plane IS the tensor.  fp32 stays:
# accumulators, norm statistics, master weights, time biases, parameter gradients.  colddiff/bf16store.py holds the block-level nodes.
# ---------------------------------------------------------------------------------------------------
BF = torch.bfloat16
IO_RES, IO_PRE, IO_MUL, IO_PRE_GRAD = 1, 2, 4, 8           # include/colddiff.h: io_bf16 bits of cdf_conv_gemm_bf16x_io


def is_bf(t):
    return t is not None and t.dtype == BF


def new_bf(ref, B, H, W, C):
    assert C % 8 == 0, "bf16 feature maps keep channel counts in multiples of 8 (16-byte GEMM operand rows)"
    return torch.empty((B, H, W, C), device=ref.device, dtype=BF)


def to_f32(x, C=None):
    """bf16 feature map -> fresh fp32 tensor (exact); for the few kernels without a bf16-input form."""
    C = x.shape[-1] if C is None else C
    y = torch.empty(x.shape[:-1] + (r4(C),), device=x.device, dtype=torch.float32)
    rt.lib().cdf_bf16_to_f32(P(x), ld_of(x), P(y), y.shape[-1], x.numel() // x.shape[-1], r4(C), rt.stream(x))
    return y


def f32_of(x, C=None):
    return to_f32(x, C) if is_bf(x) else x


def to_bf16(x, out=None):
    """fp32 feature map -> bf16 (round to nearest even: cdf_split_bf16 with the hi plane only); `out`: destination view (any pitch % 8)."""
    C = x.shape[-1]
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=BF)
    rt.lib().cdf_split_bf16(P(x), ld_of(x), P(out), 0, ld_of(out), x.numel() // C, C, rt.stream(x))
    return out


def conv_gemm_bf(plan, x, Cin, wp, Cout, y=None, bias=None, sbias=None, res=None, pre=None, mul=None, act=0, mul_mode=0, pre_grad=False):
    """The pre-split GEMM on a bf16 tensor (x IS its operand plane; wp = (hi, None) packed weights) -> bf16 output (y: optional
    destination view).  res / mul: bf16 or fp32 tensors; pre: a bf16 (or fp32) tensor that receives the pre-activation."""
    assert is_bf(x) and wp[1] is None and Cin % 8 == 0 and Cout % 4 == 0
    B = x.shape[0]
    if y is None:
        y = new_bf(x, B, plan.OH, plan.OW, Cout)             # (asserts Cout % 8: a bf16 tensor must be a legal operand plane of the next GEMM)
    ldv = lambda t: 0 if t is None else ld_of(t)
    M = B * plan.QH * plan.QW
    ws, nws = None, 0
    if M <= 4096 and x.device.type != "meta":
        ks = rt.lib().cdf_conv_gemm_bf16x_ksplit(M, Cout, plan.nphase, plan.desc[2], rt.tune_ptr())
        if ks > 1:
            nws = ks * M * r4(Cout)
            ws = torch.empty((nws,), device=x.device, dtype=torch.float32)
    io = (IO_RES if is_bf(res) else 0) | (IO_PRE if is_bf(pre) else 0) | (IO_MUL if is_bf(mul) else 0) | (IO_PRE_GRAD if pre_grad else 0)
    rt.lib().cdf_conv_gemm_bf16x_io(P(x), 0, ld_of(x), P(zero_page(x.device)), P(wp[0]), 0, wp[0].shape[-1], 0, Cout,
                                    B, plan.H, plan.W, Cin, plan.OH, plan.OW, Cout, plan.QH, plan.QW, plan.os, plan.istride, plan.nphase,
                                    plan.desc, P(bias), P(sbias), 0 if sbias is None else sbias.stride(0), P(res), ldv(res), P(pre),
                                    ldv(pre), P(mul), ldv(mul), act, mul_mode, 0, io, P(y), 0, ld_of(y), P(ws), nws, rt.tune_ptr(), rt.stream(x))
    return y


def dwconv7_bf(x, wp, bias, sbias, flip=0, y=None, accumulate=0, res=None, out_f32=False):
    """ops.dwconv7 on bf16 tensors (x, y, res all bf16; out_f32: an fp32 y from bf16 x / res)."""
    B, H, W, Cp = x.shape
    assert is_bf(x) and (res is None or is_bf(res))
    if y is None:
        y = torch.empty((B, H, W, Cp), device=x.device, dtype=torch.float32 if out_f32 else BF)
    rt.lib().cdf_dwconv7_io(P(x), ld_of(x), P(wp), wp.shape[-1], P(bias), P(sbias), 0 if sbias is None else sbias.stride(0), P(y),
                            ld_of(y), B, H, W, Cp, flip, accumulate, P(res), 0 if res is None else ld_of(res), 1 if is_bf(y) else 2, rt.stream(x))
    return y


def dwconv7_wgrad_bf(x, dy, w_param, b_param, want_dsb, dsb_out=None):
    L = rt.lib()
    B, H, W, Cp = x.shape
    C = w_param.shape[0]
    ws = torch.empty((B * L.cdf_dwconv7_wgrad_nchunk(H) * 50 * C,), device=x.device, dtype=torch.float32)
    dsb = None
    if want_dsb:
        dsb = dsb_out if dsb_out is not None else torch.zeros((B, Cp), device=x.device, dtype=torch.float32)
    L.cdf_dwconv7_wgrad_io(P(x), ld_of(x), P(dy), ld_of(dy), P(grad_of(w_param)), P(grad_of(b_param)), P(dsb),
                           0 if dsb is None else dsb.stride(0), P(ws), B, H, W, C, 1, 1, rt.stream(x))
    return dsb


def layernorm_fwd_bf(x, g, b, eps, save, out_f32=False, planes=False):
    """Channel LayerNorm of a bf16 tensor -> (y, mean, rstd); y bf16 (the next GEMM's operand plane) or, out_f32, fp32 -- with `planes`
    additionally the bf16 plane of the same values: (y fp32, mean, rstd, y bf16)."""
    B, H, W, C = x.shape
    M = B * H * W
    y = torch.empty((B, H, W, C), device=x.device, dtype=torch.float32 if out_f32 else BF)
    yb = torch.empty((B, H, W, C), device=x.device, dtype=BF) if (out_f32 and planes) else None
    mean = torch.empty((M,), device=x.device, dtype=torch.float32) if save else None
    rstd = torch.empty((M,), device=x.device, dtype=torch.float32) if save else None
    rt.lib().cdf_layernorm_c_fwd_io(P(x), ld_of(x), P(y) if out_f32 else 0, C, P(g), P(b), P(mean), P(rstd), M, C, eps,
                                    P(yb) if out_f32 else P(y), 0, 0 if (out_f32 and not planes) else C, 1, rt.stream(x))
    return (y, mean, rstd, yb) if (out_f32 and planes) else (y, mean, rstd)


def layernorm_bwd_bf(dy, x, g_param, b_param, mean, rstd, add=None):
    """dx (bf16) of the channel LayerNorm of the bf16 tensor x; dy bf16 (ConvNeXt block) or fp32 with a bf16 `add` (attention block:
    dx = grad + add).  Accumulates the g / b gradients."""
    L = rt.lib()
    B, H, W, C = x.shape
    M = B * H * W
    io = 7 if is_bf(dy) else 14
    assert is_bf(x) and (io == 14 or add is None) and (add is None or is_bf(add))
    dx = torch.empty((B, H, W, C), device=x.device, dtype=BF)
    part = torch.empty((L.cdf_layernorm_blocks(M, C) * 2 * C,), device=x.device, dtype=torch.float32)
    L.cdf_layernorm_c_bwd_io(P(dy), ld_of(dy), P(x), ld_of(x), P(g_param), P(mean), P(rstd), P(dx), C, P(add),
                             0 if add is None else ld_of(add), P(grad_of(g_param)), P(grad_of(b_param)), P(part), M, C, 0, 1, io, rt.stream(x))
    return dx

"""Tap tables that express every dense convolution of the two UNets (forward, data gradient,
weight gradient) as the one gather-GEMM the HIP library implements (see csrc/k_conv.hip).

A *phase descriptor* is the flat int list ``[oy, ox, ntaps, (dy, dx, wi) * ntaps]`` per output
phase; a *wgrad tap descriptor* is ``(day, dax, dby, dbx) * ntaps``.  Tap index ``wi = ky*KW+kx``.
"""
import ctypes
from functools import lru_cache


def _c_int_array(vals):
    return (ctypes.c_int * len(vals))(*vals)


class GemmPlan:
    """Geometry of one cdf_conv_gemm call."""
    __slots__ = ("H", "W", "OH", "OW", "QH", "QW", "os", "istride", "nphase", "desc", "ntaps_w")

    def __init__(self, H, W, OH, OW, QH, QW, os, istride, phases, ntaps_w):
        self.H, self.W, self.OH, self.OW, self.QH, self.QW = H, W, OH, OW, QH, QW
        self.os, self.istride = os, istride
        self.nphase = len(phases)
        flat = []
        for (oy, ox, taps) in phases:
            flat += [oy, ox, len(taps)]
            for t in taps:
                flat += list(t)
        self.desc = _c_int_array(flat)
        self.ntaps_w = ntaps_w


@lru_cache(maxsize=None)
def conv_fwd(H, W, KH, KW, stride, pad_t, pad_l, pad_b, pad_r):
    """y = conv2d(x) with explicit (possibly asymmetric) zero padding."""
    OH = (H + pad_t + pad_b - KH) // stride + 1
    OW = (W + pad_l + pad_r - KW) // stride + 1
    taps = [(ky - pad_t, kx - pad_l, ky * KW + kx) for ky in range(KH) for kx in range(KW)]
    return GemmPlan(H, W, OH, OW, OH, OW, 1, stride, [(0, 0, taps)], KH * KW)


def _transposed_phases(s, KH, KW, pad_t, pad_l):
    """Output-parity phases of a stride-s transposed gather: out coordinate o = q*s + p reads
    source coordinate q + (p + pad - k)/s for the taps k with (p + pad - k) % s == 0."""
    phases = []
    for py in range(s):
        for px in range(s):
            taps = []
            for ky in range(KH):
                if (py + pad_t - ky) % s:
                    continue
                for kx in range(KW):
                    if (px + pad_l - kx) % s:
                        continue
                    taps.append(((py + pad_t - ky) // s, (px + pad_l - kx) // s, ky * KW + kx))
            phases.append((py, px, taps))
    return phases


@lru_cache(maxsize=None)
def conv_dgrad(H, W, KH, KW, stride, pad_t, pad_l, pad_b, pad_r):
    """dX of conv_fwd(H, W, ...): gathers from dY [OH, OW]; output grid is the input image."""
    OH = (H + pad_t + pad_b - KH) // stride + 1
    OW = (W + pad_l + pad_r - KW) // stride + 1
    assert H % stride == 0 and W % stride == 0
    phases = _transposed_phases(stride, KH, KW, pad_t, pad_l)
    return GemmPlan(OH, OW, H, W, H // stride, W // stride, stride, 1, phases, KH * KW)


@lru_cache(maxsize=None)
def convT_fwd(H, W, KH, KW, stride, pad):
    """y = conv_transpose2d(x, stride, padding=pad); OH = (H-1)*stride - 2*pad + KH."""
    OH = (H - 1) * stride - 2 * pad + KH
    OW = (W - 1) * stride - 2 * pad + KW
    assert OH % stride == 0 and OW % stride == 0
    phases = _transposed_phases(stride, KH, KW, pad, pad)
    return GemmPlan(H, W, OH, OW, OH // stride, OW // stride, stride, 1, phases, KH * KW)


@lru_cache(maxsize=None)
def convT_dgrad(H, W, KH, KW, stride, pad):
    """dX of convT_fwd: a regular strided conv over dY [OH, OW] onto the input grid [H, W]."""
    OH = (H - 1) * stride - 2 * pad + KH
    OW = (W - 1) * stride - 2 * pad + KW
    taps = [(ky - pad, kx - pad, ky * KW + kx) for ky in range(KH) for kx in range(KW)]
    return GemmPlan(OH, OW, H, W, H, W, 1, stride, [(0, 0, taps)], KH * KW)


class WgradPlan:
    __slots__ = ("QH", "QW", "HA", "WA", "sa", "HB", "WB", "sb", "ntaps", "desc", "same_b", "same3x3")

    def __init__(self, QH, QW, HA, WA, sa, HB, WB, sb, taps):
        self.QH, self.QW, self.HA, self.WA, self.sa = QH, QW, HA, WA, sa
        self.HB, self.WB, self.sb = HB, WB, sb
        self.ntaps = len(taps)
        self.same_b = len(taps) >= 2 and all(t[2:] == taps[0][2:] for t in taps)   # every tap reads B at one offset
        # 3 x 3 stride-1 same-size convolution: A shifted by (-1..1, -1..1) in row-major tap order, B read in place
        self.same3x3 = (len(taps) == 9 and sa == 1 and sb == 1 and (HA, WA, HB, WB) == (QH, QW, QH, QW) and
                        [tuple(t) for t in taps] == [(dy, dx, 0, 0) for dy in (-1, 0, 1) for dx in (-1, 0, 1)])
        flat = []
        for t in taps:
            flat += list(t)
        self.desc = _c_int_array(flat)


@lru_cache(maxsize=None)
def conv_wgrad(H, W, KH, KW, stride, pad_t, pad_l, pad_b, pad_r):
    """dW[tap][ci][co] = sum_m X[m*stride - pad + k][ci] * dY[m][co] over the output grid."""
    OH = (H + pad_t + pad_b - KH) // stride + 1
    OW = (W + pad_l + pad_r - KW) // stride + 1
    taps = [(ky - pad_t, kx - pad_l, 0, 0) for ky in range(KH) for kx in range(KW)]
    return WgradPlan(OH, OW, H, W, stride, OH, OW, 1, taps)


@lru_cache(maxsize=None)
def convT_wgrad(H, W, KH, KW, stride, pad):
    """dW[tap][ci][co] = sum_m X[m][ci] * dY[m*stride - pad + k][co] over the input grid."""
    OH = (H - 1) * stride - 2 * pad + KH
    OW = (W - 1) * stride - 2 * pad + KW
    taps = [(0, 0, ky - pad, kx - pad) for ky in range(KH) for kx in range(KW)]
    return WgradPlan(H, W, H, W, 1, OH, OW, stride, taps)

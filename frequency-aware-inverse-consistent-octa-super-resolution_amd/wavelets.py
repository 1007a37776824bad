"""2-D Haar DWT / IDWT modules on fused HIP kernels, behind the interface of the reference's
vendored pytorch_wavelets (pytorch_wavelets/pytorch_wavelets/dwt/transform2d.py:7-148,
lowlevel.py:312-365,647-694).

Scope: wave 'haar'/'db1', even H and W, the padding modes for which the
2-tap bank needs no padding at even sizes ('zero', 'symmetric', 'reflect', 'periodic':
lowlevel.py:153-154 gives p = 0).  Anything else raises -- the OCTA code never reaches it.
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function

from . import ops

_S = 1.0 / math.sqrt(2.0)
_MODES = {"zero": 0, "symmetric": 1, "per": 2, "periodization": 2, "constant": 3, "reflect": 4, "replicate": 5, "periodic": 6}
_NATIVE_MODES = (0, 1, 4, 6)


def mode_to_int(mode):
    """lowlevel.py:274-290."""
    if mode not in _MODES:
        raise ValueError("Unkown pad type: {}".format(mode))
    return _MODES[mode]


def int_to_mode(mode):
    """lowlevel.py:293-309."""
    for k, v in (("zero", 0), ("symmetric", 1), ("periodization", 2), ("constant", 3), ("reflect", 4), ("replicate", 5), ("periodic", 6)):
        if v == mode:
            return k
    raise ValueError("Unkown pad type: {}".format(mode))


def _check_haar(wave):
    if isinstance(wave, str):
        if wave not in ("haar", "db1"):
            raise NotImplementedError("only the Haar wavelet is built (the OCTA path uses wave='haar', model.py:140,190); got %r" % (wave,))
        return
    taps = [list(map(float, torch.as_tensor(w).flatten().tolist())) for w in wave]
    if any(len(t) != 2 or abs(abs(t[0]) - _S) > 1e-6 or abs(abs(t[1]) - _S) > 1e-6 for t in taps):
        raise NotImplementedError("only 2-tap Haar filter banks are built")


def _check_geometry(x, mode):
    if mode not in _NATIVE_MODES:
        raise NotImplementedError("padding mode %r is outside the built Haar path" % int_to_mode(mode))
    if x.shape[-1] % 2 or x.shape[-2] % 2:
        raise NotImplementedError("odd sizes need boundary padding; the OCTA path only transforms even sizes")


class AFB2D(Function):
    """lowlevel.py:312-365: one analysis level; ``apply(x, h0_row, h1_row, h0_col, h1_col, mode_int) -> (low, highs)``.
    The filter tensors are accepted for signature compatibility (they are the Haar taps)."""

    @staticmethod
    def forward(ctx, x, h0_row, h1_row, h0_col, h1_col, mode):
        _check_geometry(x, mode)
        ll, hi = ops._HaarAFB2D.forward(ctx, x)
        return ll, hi

    @staticmethod
    def backward(ctx, low, highs):
        return ops._HaarAFB2D.backward(ctx, low, highs), None, None, None, None, None


class SFB2D(Function):
    """lowlevel.py:647-694: one synthesis level; ``apply(low, highs, g0_row, g1_row, g0_col, g1_col, mode_int) -> y``."""

    @staticmethod
    def forward(ctx, low, highs, g0_row, g1_row, g0_col, g1_col, mode):
        if mode not in _NATIVE_MODES:
            raise NotImplementedError("padding mode %r is outside the built Haar path" % int_to_mode(mode))
        return ops._HaarSFB2D.forward(ctx, low, highs)

    @staticmethod
    def backward(ctx, dy):
        dl, dh = ops._HaarSFB2D.backward(ctx, dy)
        return dl, dh, None, None, None, None, None


class DWTForward(nn.Module):
    """transform2d.py:7-74.  forward(x) -> (yl, [yh_0 .. yh_{J-1}]), yh_j of shape (N, C, 3, H/2^{j+1}, W/2^{j+1})
    with band order LH, HL, HH; buffers h0_col, h1_col, h0_row, h1_row as registered by the reference."""

    def __init__(self, J=1, wave="db1", mode="zero"):
        super().__init__()
        _check_haar(wave)
        # prep_filt_afb2d reverses the decomposition taps (lowlevel.py:925-953): dec_lo [s,s], dec_hi [-s,s] -> [s,-s]
        self.register_buffer("h0_col", torch.tensor([_S, _S]).reshape(1, 1, 2, 1))
        self.register_buffer("h1_col", torch.tensor([_S, -_S]).reshape(1, 1, 2, 1))
        self.register_buffer("h0_row", torch.tensor([_S, _S]).reshape(1, 1, 1, 2))
        self.register_buffer("h1_row", torch.tensor([_S, -_S]).reshape(1, 1, 1, 2))
        self.J = J
        self.mode = mode

    def forward(self, x):
        yh = []
        ll = x
        mode = mode_to_int(self.mode)
        for _ in range(self.J):
            ll, high = AFB2D.apply(ll, self.h0_col, self.h1_col, self.h0_row, self.h1_row, mode)
            yh.append(high)
        return ll, yh


class DWTInverse(nn.Module):
    """transform2d.py:77-148.  forward((yl, yh)) -> x; a ``None`` entry in yh stands for zero bands."""

    def __init__(self, wave="db1", mode="zero"):
        super().__init__()
        _check_haar(wave)
        self.register_buffer("g0_col", torch.tensor([_S, _S]).reshape(1, 1, 2, 1))
        self.register_buffer("g1_col", torch.tensor([_S, -_S]).reshape(1, 1, 2, 1))
        self.register_buffer("g0_row", torch.tensor([_S, _S]).reshape(1, 1, 1, 2))
        self.register_buffer("g1_row", torch.tensor([_S, -_S]).reshape(1, 1, 1, 2))
        self.mode = mode

    def forward(self, coeffs):
        yl, yh = coeffs
        ll = yl
        mode = mode_to_int(self.mode)
        for h in yh[::-1]:
            if h is None:
                h = torch.zeros(ll.shape[0], ll.shape[1], 3, ll.shape[-2], ll.shape[-1], device=ll.device, dtype=ll.dtype)
            if ll.shape[-2] > h.shape[-2]:
                ll = ll[..., :-1, :]
            if ll.shape[-1] > h.shape[-1]:
                ll = ll[..., :-1]
            ll = SFB2D.apply(ll, h, self.g0_col, self.g1_col, self.g0_row, self.g1_row, mode)
        return ll

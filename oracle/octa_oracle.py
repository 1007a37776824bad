"""CPU oracle for the frequency-aware OCTA super-resolution train step.

TEST INFRASTRUCTURE ONLY.  This file is a CPU restatement (plain torch-CPU /
numpy arithmetic, fp32 unless a caller asks for fp64) of the algorithm the
reference executes on its hot path.  Only ``tests/``, ``__graft_entry__.smoke``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package
never does.  It is pinned against the reference itself by
``oracle/gen_golden.py`` (which imports the reference's own modules in the build
container and stores their outputs under ``tests/golden/``) and by
``tests/test_oracle_golden.py``.

Reference locations restated here (paths relative to /root/reference):
  * Haar DWT / IDWT      pytorch_wavelets/pytorch_wavelets/dwt/transform2d.py:44-74,111-148
                         pytorch_wavelets/pytorch_wavelets/dwt/lowlevel.py:91-172,226-271,312-365,647-694
  * FFT Gaussian split   utils.py:71-117
  * SSIM                 ssim.py:7-73
  * generators           model.py:238-298,403-506
  * discriminators       model.py:86-235
  * train step           train.py:73-126 (construction), train.py:166-269 (step)
  * replay buffer, init  utils.py:31-69,165-176

The dense arithmetic (conv, batch-norm, FFT) is PyTorch ATen on CPU -- the same
third-party library the reference itself calls (requirements: torch>=1.0.0); it
is not re-derived here.  Everything else is written out explicitly.
"""
from __future__ import annotations

import math
import random
import zlib
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# Haar DWT (J levels), mode='reflect', even H/W  -- closed form
# --------------------------------------------------------------------------

def haar_dwt2_level(x):
    """One analysis level.  For even H, W and the 2-tap Haar bank the reflect
    padding is empty (lowlevel.py:153-154,166: p = 2*(outsize-1) - N + L = 0), so
    AFB2D.forward (lowlevel.py:336-347) is the 2x2 block butterfly below.  Band
    order LH, HL, HH follows lowlevel.py:344-346 / transform2d.py:56-57: the row
    (width) filter runs first and its lo/hi outputs are then column filtered, so
    sub-band 1 (LH) is lo along W / hi along H."""
    a = x[..., 0::2, 0::2]
    b = x[..., 0::2, 1::2]
    c = x[..., 1::2, 0::2]
    d = x[..., 1::2, 1::2]
    ll = (a + b + c + d) * 0.5
    lh = (a + b - c - d) * 0.5
    hl = (a - b + c - d) * 0.5
    hh = (a - b - c + d) * 0.5
    return ll, torch.stack([lh, hl, hh], dim=2)


def haar_dwt2(x, J=1):
    """DWTForward(J, 'haar', 'reflect').forward (transform2d.py:44-74)."""
    yh = []
    ll = x
    for _ in range(J):
        ll, hi = haar_dwt2_level(ll)
        yh.append(hi)
    return ll, yh


def haar_idwt2_level(ll, hi):
    """One synthesis level (SFB2D.forward lowlevel.py:670-681 with Haar taps)."""
    lh, hl, hh = hi[:, :, 0], hi[:, :, 1], hi[:, :, 2]
    a = (ll + lh + hl + hh) * 0.5
    b = (ll + lh - hl - hh) * 0.5
    c = (ll - lh + hl - hh) * 0.5
    d = (ll - lh - hl + hh) * 0.5
    N, C, h, w = ll.shape
    y = ll.new_empty(N, C, 2 * h, 2 * w)
    y[..., 0::2, 0::2] = a
    y[..., 0::2, 1::2] = b
    y[..., 1::2, 0::2] = c
    y[..., 1::2, 1::2] = d
    return y


def haar_idwt2(yl, yh):
    """DWTInverse('haar','reflect').forward (transform2d.py:131-148); ``None``
    high bands are zeros (transform2d.py:137-139)."""
    ll = yl
    for h in yh[::-1]:
        if h is None:
            h = torch.zeros(ll.shape[0], ll.shape[1], 3, ll.shape[-2], ll.shape[-1], dtype=ll.dtype)
        if ll.shape[-2] > h.shape[-2]:
            ll = ll[..., :-1, :]
        if ll.shape[-1] > h.shape[-1]:
            ll = ll[..., :-1]
        ll = haar_idwt2_level(ll, h)
    return ll


# --------------------------------------------------------------------------
# FFT Gaussian frequency split
# --------------------------------------------------------------------------

def gauss_mask(rows, cols, radius, high):
    """utils.py:71-91 without the Python double loop: d^2 from the centre
    (int(rows/2), int(cols/2)); low-pass exp(-d^2/(2 r^2)), high-pass 1 - that.
    Computed in float64 then cast to float32 like ``torch.from_numpy(mask).float()``."""
    i = np.arange(rows, dtype=np.float64)[:, None] - int(rows / 2)
    j = np.arange(cols, dtype=np.float64)[None, :] - int(cols / 2)
    g = np.exp(-0.5 * (i * i + j * j) / (radius ** 2))
    m = 1.0 - g if high else g
    return torch.from_numpy(m).float()


#: "vectorised" (default) builds the mask with numpy broadcasting; "loop" builds it the way the reference does on EVERY call
#: (utils.py:71-91: a Python double loop over rows x cols into a complex array) -- same values, ~1 s per 256x256 train step.
#: bench.py times both forms of the CPU baseline.
MASK_STYLE = "vectorised"


def gauss_mask_loop(rows, cols, radius, high):
    """utils.py:71-80 (guais_low_pass) / 82-91 (guais_high_pass) as written: per-element Python arithmetic, complex mask."""
    center = int(rows / 2), int(cols / 2)
    mask = np.zeros((rows, cols), dtype=complex)
    for i in range(rows):
        for j in range(cols):
            d = (i - center[0]) ** 2 + (j - center[1]) ** 2
            g = np.exp(-0.5 * d / (radius ** 2))
            mask[i, j] = 1 - g if high else g
    return torch.from_numpy(mask).float()               # (discards the zero imaginary part, as the reference's .float() does)


def _mask(rows, cols, radius, high):
    return gauss_mask_loop(rows, cols, radius, high) if MASK_STYLE == "loop" else gauss_mask(rows, cols, radius, high)


def high_pass(timg, i=4):
    """utils.py:93-103: timg is (1,H,W); returns (H,W)."""
    f = torch.fft.fftshift(torch.fft.fft2(timg[0]))
    f = f * _mask(f.shape[0], f.shape[1], i, True).to(f.real.dtype)
    return torch.abs(torch.fft.ifft2(torch.fft.ifftshift(f)))


def low_pass(timg, i=10):
    """utils.py:105-117 (note the final ``* -1``)."""
    f = torch.fft.fftshift(torch.fft.fft2(timg[0]))
    f = f * _mask(f.shape[0], f.shape[1], i, False).to(f.real.dtype)
    return torch.abs(torch.fft.ifft2(torch.fft.ifftshift(f))) * -1


def freq_split(x, r_hp, r_lp):
    """train.py:173-175 applied per sample (the reference only supports batch 1:
    it filters sample 0 and broadcasts -- SURVEY fact 3).  x: (B,1,H,W).
    Returns hf = (high_pass(x_b)+x_b)/2 and lf = low_pass(x_b), both (B,1,H,W)."""
    hfs, lfs = [], []
    for b in range(x.shape[0]):
        hfs.append(high_pass(x[b], r_hp))
        lfs.append(low_pass(x[b], r_lp))
    hf = torch.stack(hfs).unsqueeze(1)
    lf = torch.stack(lfs).unsqueeze(1)
    return (hf + x) / 2.0, lf


def circulant_lowpass_matrix(n, radius, dtype=torch.float64):
    """The separable structure the HIP path exploits, stated here so tests can
    check it against the FFT form: the shifted Gaussian mask factorises as
    g(u) g(v), so ifft2(ifftshift(fftshift(fft2 x) * mask)) == C_H x C_W^T with
    C[a,b] = (1/n) sum_k g'(k) cos(2 pi k (a-b)/n), g'(k) = exp(-dist(k)^2/(2 r^2)),
    dist(k) = k for k < ceil(n/2)... i.e. the ifftshift of the centred taps."""
    k = np.arange(n)
    centred = np.exp(-0.5 * (k - int(n / 2)) ** 2 / radius ** 2)
    g = np.fft.ifftshift(centred)  # g'[k]
    col = np.real(np.fft.ifft(g))  # first column of the circulant
    idx = (k[:, None] - k[None, :]) % n
    return torch.from_numpy(col[idx]).to(dtype)


# --------------------------------------------------------------------------
# SSIM
# --------------------------------------------------------------------------

def ssim_window_1d(window_size=11, sigma=1.5):
    """ssim.py:7-9."""
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def ssim(img1, img2, window_size=11, size_average=True):
    """ssim.py:17-37 with the window of ssim.py:11-15 (outer product of the
    normalised 1-D taps, one copy per channel, zero padding window_size//2)."""
    ch = img1.shape[1]
    w1 = ssim_window_1d(window_size).to(img1.dtype)
    w2 = torch.outer(w1, w1)[None, None].expand(ch, 1, window_size, window_size).contiguous()
    p = window_size // 2
    mu1 = F.conv2d(img1, w2, padding=p, groups=ch)
    mu2 = F.conv2d(img2, w2, padding=p, groups=ch)
    s11 = F.conv2d(img1 * img1, w2, padding=p, groups=ch) - mu1 * mu1
    s22 = F.conv2d(img2 * img2, w2, padding=p, groups=ch) - mu2 * mu2
    s12 = F.conv2d(img1 * img2, w2, padding=p, groups=ch) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


# --------------------------------------------------------------------------
# Parameter specs (state_dict keys and shapes) and the name-keyed init rule
# --------------------------------------------------------------------------

def _bn(spec, p, c):
    spec[p + ".weight"] = ((c,), "bn_w")
    spec[p + ".bias"] = ((c,), "bn_b")
    spec[p + ".running_mean"] = ((c,), "bn_rm")
    spec[p + ".running_var"] = ((c,), "bn_rv")
    spec[p + ".num_batches_tracked"] = ((), "bn_nbt")


def _conv(spec, p, cout, cin, k, bias):
    spec[p + ".weight"] = ((cout, cin, k, k), "conv_w")
    if bias:
        spec[p + ".bias"] = ((cout,), ("conv_b", cin * k * k))


def _convT(spec, p, cin, cout, k, bias):
    # ConvTranspose2d weight is (Cin, Cout, k, k); torch's default bias bound uses
    # fan_in computed from weight.size(1)*k*k = Cout*k*k
    spec[p + ".weight"] = ((cin, cout, k, k), "conv_w")
    if bias:
        spec[p + ".bias"] = ((cout,), ("conv_b", cout * k * k))


def _unet_block(spec, p, outer, inner, input_nc=None, sub=None, outermost=False, innermost=False, dropout=True):
    """Key layout of model.py:336-400 (the U-Net is dead code on the hot path --
    model.py:262-268 never calls it -- but its tensors exist in the state_dict)."""
    if input_nc is None:
        input_nc = outer
    if outermost:
        _conv(spec, p + ".model.0", inner, input_nc, 4, True)
        sub(p + ".model.1")
    elif innermost:
        _conv(spec, p + ".model.1", inner, input_nc, 4, True)
        _convT(spec, p + ".model.3", inner, outer, 4, True)
        _bn(spec, p + ".model.4", outer)
    else:
        _conv(spec, p + ".model.1", inner, input_nc, 4, True)
        _bn(spec, p + ".model.2", inner)
        sub(p + ".model.3")
        _convT(spec, p + ".model.5", inner * 2, outer, 4, True)
        _bn(spec, p + ".model.6", outer)


def _unet(spec, p, input_nc=64, output_nc=64, num_downs=7, ngf=64):
    """model.py:302-326."""
    def inner(q):
        _unet_block(spec, q, ngf * 8, ngf * 8, innermost=True)
    blk = inner
    for _ in range(num_downs - 5):
        prev = blk
        blk = (lambda pv: (lambda q: _unet_block(spec, q, ngf * 8, ngf * 8, sub=pv)))(prev)
    for (o, i) in ((ngf * 4, ngf * 8), (ngf * 2, ngf * 4), (ngf, ngf * 2)):
        prev = blk
        blk = (lambda pv, o=o, i=i: (lambda q: _unet_block(spec, q, o, i, sub=pv)))(prev)
    _unet_block(spec, p + ".model", output_nc, ngf, input_nc=input_nc, sub=blk, outermost=True)


def _resnet_generator(spec, p, input_nc, output_nc=64, ngf=64, n_blocks=8):
    """model.py:444-476: indices inside the Sequential."""
    _conv(spec, p + ".model.1", ngf, input_nc, 7, False)
    _bn(spec, p + ".model.2", ngf)
    _conv(spec, p + ".model.4", ngf * 2, ngf, 3, False)
    _bn(spec, p + ".model.5", ngf * 2)
    _conv(spec, p + ".model.7", ngf * 4, ngf * 2, 3, False)
    _bn(spec, p + ".model.8", ngf * 4)
    for i in range(n_blocks):
        q = "%s.model.%d.conv_block" % (p, 10 + i)
        _conv(spec, q + ".0", ngf * 4, ngf * 4, 3, False)
        _bn(spec, q + ".1", ngf * 4)
        _conv(spec, q + ".3", ngf * 4, ngf * 4, 3, False)
        _bn(spec, q + ".4", ngf * 4)
    b = 10 + n_blocks
    _convT(spec, "%s.model.%d" % (p, b), ngf * 4, ngf * 2, 3, False)
    _bn(spec, "%s.model.%d" % (p, b + 1), ngf * 2)
    _convT(spec, "%s.model.%d" % (p, b + 3), ngf * 2, ngf, 3, False)
    _bn(spec, "%s.model.%d" % (p, b + 4), ngf)
    _conv(spec, "%s.model.%d" % (p, b + 7), output_nc, ngf, 7, True)


def _shallow_net_up(spec, p, in_dim=128):
    """model.py:423-439 with up=True."""
    _convT(spec, p + ".model.1", in_dim, 64, 4, False)
    _bn(spec, p + ".model.2", 64)
    for i in (3, 4, 5):
        q = "%s.model.%d.conv_block" % (p, i)
        _conv(spec, q + ".0", 64, 64, 3, False)
        _bn(spec, q + ".1", 64)
        _conv(spec, q + ".3", 64, 64, 3, False)
        _bn(spec, q + ".4", 64)
    _conv(spec, p + ".model.7", 1, 64, 3, False)


def _shallow_frequency(spec, p):
    """model.py:242-246 / 275-279."""
    _conv(spec, p + ".0", 64, 1, 4, False)
    _conv(spec, p + ".2", 128, 64, 3, False)
    _bn(spec, p + ".3", 128)
    _conv(spec, p + ".5", 64, 128, 3, False)
    _bn(spec, p + ".6", 64)


def spec_network_a2b():
    """model.py:239-260."""
    s = OrderedDict()
    _unet(s, "unet")
    _shallow_frequency(s, "shallow_frequency")
    _shallow_net_up(s, "shallow_up")
    _conv(s, "skip.1", 64, 128, 3, False)
    _bn(s, "skip.2", 64)
    _convT(s, "unet_up.1", 128, 64, 4, False)
    _bn(s, "unet_up.2", 64)
    _conv(s, "A2B_input.0", 64, 1, 4, False)
    _resnet_generator(s, "resnet", 64)
    return s


def spec_network_b2a():
    """model.py:272-287."""
    s = OrderedDict()
    _shallow_frequency(s, "shallow_frequency")
    _shallow_net_up(s, "shallow_up")
    _conv(s, "skip.1", 64, 128, 3, False)
    _bn(s, "skip.2", 64)
    _resnet_generator(s, "resnet", 128)
    _conv(s, "B2A_input.0", 128, 1, 4, False)
    return s


def _patch_discriminator(spec, p, input_nc, ndf=64, n_layers=5):
    """model.py:86-123."""
    _conv(spec, p + ".model.0", ndf, input_nc, 4, True)
    idx, prev = 2, 1
    for n in range(1, n_layers):
        mult = min(2 ** n, 8)
        _conv(spec, "%s.model.%d" % (p, idx), ndf * mult, ndf * prev, 4, True)
        _bn(spec, "%s.model.%d" % (p, idx + 1), ndf * mult)
        idx, prev = idx + 3, mult
    mult = min(2 ** n_layers, 8)
    _conv(spec, "%s.model.%d" % (p, idx), ndf * mult, ndf * prev, 4, True)
    _bn(spec, "%s.model.%d" % (p, idx + 1), ndf * mult)
    _conv(spec, "%s.model.%d" % (p, idx + 3), 1, ndf * mult, 4, True)


def spec_fs_discriminator(cs):
    """model.py:132-152 (cs='sum') / 182-204 (cs='cat')."""
    s = OrderedDict()
    for k, shape in (("h0_col", (1, 1, 2, 1)), ("h1_col", (1, 1, 2, 1)), ("h0_row", (1, 1, 1, 2)), ("h1_row", (1, 1, 1, 2))):
        s["DWT2." + k] = (shape, "dwt_" + k[:2])
    _patch_discriminator(s, "net", 1)
    _patch_discriminator(s, "net_dwt", 1 if cs == "sum" else 3)
    return s


DEAD_PREFIXES = {
    "A2B": ("unet.", "unet_up."),          # model.py:241,254-257 never run by forward (model.py:262-268)
    "B2A": ("skip.",),                     # model.py:281-284 never run by forward (model.py:290-298)
}


def make_state(spec, tag, seed=0, dtype=torch.float32):
    """Name-keyed deterministic init: every tensor is drawn from its own
    generator seeded by crc32(tag/key)+seed, with the distributions of
    utils.py:63-69 (Conv*: weight ~ N(0, 0.02), bias left at torch's default
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)); BatchNorm2d: weight ~ N(1, 0.02), bias 0)."""
    out = OrderedDict()
    s = 1.0 / math.sqrt(2.0)
    for key, (shape, kind) in spec.items():
        g = torch.Generator().manual_seed((zlib.crc32((tag + "/" + key).encode()) + seed) & 0x7FFFFFFF)
        if kind == "conv_w":
            t = torch.empty(shape, dtype=torch.float32).normal_(0.0, 0.02, generator=g)
        elif isinstance(kind, tuple) and kind[0] == "conv_b":
            bound = 1.0 / math.sqrt(kind[1])
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        elif kind == "bn_w":
            t = torch.empty(shape, dtype=torch.float32).normal_(1.0, 0.02, generator=g)
        elif kind in ("bn_b", "bn_rm"):
            t = torch.zeros(shape)
        elif kind == "bn_rv":
            t = torch.ones(shape)
        elif kind == "bn_nbt":
            out[key] = torch.zeros((), dtype=torch.long)
            continue
        elif kind == "dwt_h0":
            # prep_filt_afb2d reverses the taps (lowlevel.py:925-953); Haar dec_lo = [s, s]
            t = torch.tensor([s, s]).reshape(shape)
        elif kind == "dwt_h1":
            # dec_hi = [-s, s] reversed -> [s, -s]
            t = torch.tensor([s, -s]).reshape(shape)
        else:
            raise ValueError(kind)
        out[key] = t.to(dtype)
    return out


def make_eval_state(spec, tag, seed=0):
    """``make_state`` with the state a trained checkpoint has where a fresh one is trivial: BatchNorm biases ~ N(0, 0.1),
    running means ~ N(0, 0.1), running variances ~ U(0.5, 1.5), each from its own name-keyed generator.  Used by the
    eval-mode fixtures (oracle/gen_golden.gen_eval: utils.py:186 `model.eval()`), where fresh statistics (0, 1) would make
    the BatchNorm fold a no-op."""
    out = make_state(spec, tag, seed)
    for key, (shape, kind) in spec.items():
        if kind not in ("bn_b", "bn_rm", "bn_rv"):
            continue
        g = torch.Generator().manual_seed((zlib.crc32((tag + "/eval/" + key).encode()) + seed) & 0x7FFFFFFF)
        if kind == "bn_rv":
            out[key] = torch.rand(shape, generator=g, dtype=torch.float32) + 0.5
        else:
            out[key] = torch.empty(shape, dtype=torch.float32).normal_(0.0, 0.1, generator=g)
    return out


def synthetic_batch(B, H, seed=1234, rank=0):
    """uniform [-1,1) images, the range of Normalize(0.5,0.5) (train.py:133,138)."""
    g = torch.Generator().manual_seed(seed + rank)
    a = torch.rand(B, 1, H, H, generator=g) * 2 - 1
    b = torch.rand(B, 1, H, H, generator=g) * 2 - 1
    return a, b


# --------------------------------------------------------------------------
# Functional networks over a flat {key: tensor} state
# --------------------------------------------------------------------------

class Net:
    """A flat parameter store with BatchNorm running-stat updates."""

    def __init__(self, state, train=True, momentum=0.1, eps=1e-5):
        self.s = state
        self.train = train
        self.momentum = momentum
        self.eps = eps

    def conv(self, x, p, stride=1, pad=0):
        return F.conv2d(x, self.s[p + ".weight"], self.s.get(p + ".bias"), stride=stride, padding=pad)

    def convT(self, x, p, stride, pad, opad=0):
        return F.conv_transpose2d(x, self.s[p + ".weight"], self.s.get(p + ".bias"), stride=stride, padding=pad, output_padding=opad)

    def bn(self, x, p):
        """nn.BatchNorm2d training mode: batch statistics, momentum 0.1, eps 1e-5."""
        if self.train:
            self.s[p + ".num_batches_tracked"] += 1
        return F.batch_norm(x, self.s[p + ".running_mean"], self.s[p + ".running_var"], self.s[p + ".weight"],
                            self.s[p + ".bias"], self.train, self.momentum, self.eps)


def resnet_generator(n, p, x, n_blocks=8):
    """model.py:450-480."""
    x = F.relu(n.bn(n.conv(F.pad(x, (3, 3, 3, 3), mode="reflect"), p + ".model.1"), p + ".model.2"))
    x = F.relu(n.bn(n.conv(x, p + ".model.4", 2, 1), p + ".model.5"))
    x = F.relu(n.bn(n.conv(x, p + ".model.7", 2, 1), p + ".model.8"))
    for i in range(n_blocks):
        q = "%s.model.%d.conv_block" % (p, 10 + i)
        y = F.relu(n.bn(n.conv(x, q + ".0", 1, 1), q + ".1"))
        x = x + n.bn(n.conv(y, q + ".3", 1, 1), q + ".4")           # model.py:503-506
    b = 10 + n_blocks
    x = F.relu(n.bn(n.convT(x, "%s.model.%d" % (p, b), 2, 1, 1), "%s.model.%d" % (p, b + 1)))
    x = F.relu(n.bn(n.convT(x, "%s.model.%d" % (p, b + 3), 2, 1, 1), "%s.model.%d" % (p, b + 4)))
    return n.conv(F.pad(x, (3, 3, 3, 3), mode="reflect"), "%s.model.%d" % (p, b + 7))   # no tanh (model.py:474)


def shallow_frequency(n, p, x):
    """model.py:242-246."""
    x = F.leaky_relu(n.conv(x, p + ".0", 2, 1), 0.2)
    x = F.relu(n.bn(n.conv(x, p + ".2", 1, 1), p + ".3"))
    return n.bn(n.conv(x, p + ".5", 1, 1), p + ".6")


def shallow_up(n, p, x):
    """model.py:431-442 (up=True) with ResnetBlock model.py:403-421."""
    x = n.bn(n.convT(F.relu(x), p + ".model.1", 2, 1), p + ".model.2")
    for i in (3, 4, 5):
        q = "%s.model.%d.conv_block" % (p, i)
        y = F.relu(n.bn(n.conv(x, q + ".0", 1, 1), q + ".1"))
        x = x + n.bn(n.conv(y, q + ".3", 1, 1), q + ".4")
    return torch.tanh(n.conv(F.relu(x), p + ".model.7", 1, 1))


def network_a2b(n, lf, hf):
    """model.py:262-268 -> (lf_feature, hf_feature, out)."""
    lf_feature = shallow_frequency(n, "shallow_frequency", lf)
    hin = n.conv(hf, "A2B_input.0", 2, 1)
    cat = torch.cat([hin, resnet_generator(n, "resnet", hin)], 1)
    hf_feature = n.bn(n.conv(F.relu(cat), "skip.1", 1, 1), "skip.2")
    return lf_feature, hf_feature, shallow_up(n, "shallow_up", torch.cat([lf_feature, hf_feature], 1))


def network_b2a(n, hf, lf):
    """model.py:290-298 -> (hf_feature, lf_feature, out)."""
    hf_feature = shallow_frequency(n, "shallow_frequency", hf)
    lf_feature = resnet_generator(n, "resnet", n.conv(lf, "B2A_input.0", 2, 1))
    return hf_feature, lf_feature, shallow_up(n, "shallow_up", torch.cat([hf_feature, lf_feature], 1))


def patch_discriminator(n, p, x, n_layers=5):
    """model.py:102-127."""
    x = F.leaky_relu(n.conv(x, p + ".model.0", 2, 1), 0.2)
    idx = 2
    for _ in range(1, n_layers):
        x = F.leaky_relu(n.bn(n.conv(x, "%s.model.%d" % (p, idx), 2, 1), "%s.model.%d" % (p, idx + 1)), 0.2)
        idx += 3
    x = F.leaky_relu(n.bn(n.conv(x, "%s.model.%d" % (p, idx), 1, 1), "%s.model.%d" % (p, idx + 1)), 0.2)
    return n.conv(x, "%s.model.%d" % (p, idx + 3), 1, 1)


def fs_discriminator(n, x, cs):
    """model.py:154-179 (cs='sum': wavelet branch sees LL) and model.py:207-235
    (cs='cat': wavelet branch sees cat(LH,HL,HH)*0.5+0.5)."""
    ll, yh = haar_dwt2(x, 1)
    if cs == "sum":
        dwt = ll
    else:
        dwt = torch.cat([yh[0][:, :, 0], yh[0][:, :, 1], yh[0][:, :, 2]], 1) * 0.5 + 0.5
    xd = patch_discriminator(n, "net", x).mean(dim=(2, 3)).view(x.shape[0], -1)
    dd = patch_discriminator(n, "net_dwt", dwt).mean(dim=(2, 3)).view(x.shape[0], -1)
    return torch.flatten(0.7 * xd + 0.3 * dd)


# --------------------------------------------------------------------------
# Replay buffer, AdamW, the train step
# --------------------------------------------------------------------------

class ReplayBuffer:
    """utils.py:31-51 (python ``random`` draws, one uniform then optionally one randint per element)."""

    def __init__(self, max_size=50, rng=None):
        self.max_size = max_size
        self.data = []
        self.rng = rng or random

    def push_and_pop(self, data):
        out = []
        for element in data.detach():
            element = element.unsqueeze(0)
            if len(self.data) < self.max_size:
                self.data.append(element)
                out.append(element)
            elif self.rng.uniform(0, 1) > 0.5:
                i = self.rng.randint(0, self.max_size - 1)
                out.append(self.data[i].clone())
                self.data[i] = element
            else:
                out.append(element)
        return torch.cat(out)


class AdamW:
    """torch.optim.AdamW arithmetic (train.py:102-103: lr 1.3e-4, betas (0.9,0.999),
    eps 1e-8, decoupled weight_decay 0.01); parameters whose grad is None are skipped."""

    def __init__(self, params, lr=1.3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, betas[0], betas[1], eps, weight_decay
        self.state = {}

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        for p in self.params:
            if p.grad is None:
                continue
            st = self.state.setdefault(id(p), {"t": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
            st["t"] += 1
            t = st["t"]
            p.mul_(1 - self.lr * self.wd)
            st["m"].mul_(self.b1).add_(p.grad, alpha=1 - self.b1)
            st["v"].mul_(self.b2).addcmul_(p.grad, p.grad, value=1 - self.b2)
            bc1 = 1 - self.b1 ** t
            bc2 = 1 - self.b2 ** t
            denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(st["m"], denom, value=-self.lr / bc1)


LOSS_WEIGHTS = dict(beta1=0.25, beta2=10.0, beta3=2.0, beta4=0.5, beta5=0.5)   # train.py:50-54


def _is_float_param(kind):
    return kind in ("conv_w", "bn_w", "bn_b") or (isinstance(kind, tuple) and kind[0] == "conv_b")


class StepOracle:
    """State + one training iteration following train.py:73-126 and 166-269."""

    def __init__(self, seed=0, lr=1.3e-4, dtype=torch.float32, weights=LOSS_WEIGHTS, ssim_weight=0.0, whf_weight=0.0, dwt_levels=1,
                 states=None):
        self.dtype = dtype
        self.w = dict(weights)
        self.ssim_weight, self.whf_weight, self.dwt_levels = ssim_weight, whf_weight, dwt_levels
        specs = {"A2B": spec_network_a2b(), "B2A": spec_network_b2a(), "D_A": spec_fs_discriminator("sum"), "D_B": spec_fs_discriminator("cat")}
        self.specs = specs
        self.state = states or {k: make_state(v, k, seed, dtype) for k, v in specs.items()}
        self.params = {}
        for k, spec in specs.items():
            ps = []
            for key, (_, kind) in spec.items():
                if _is_float_param(kind):
                    self.state[k][key].requires_grad_(True)
                    ps.append(self.state[k][key])
            self.params[k] = ps
        self.nets = {k: Net(self.state[k]) for k in specs}
        self.opt_G = AdamW(self.params["A2B"] + self.params["B2A"], lr=lr)          # train.py:102
        self.opt_D = AdamW(self.params["D_A"] + self.params["D_B"], lr=lr)          # train.py:103
        self.fake_A_buffer, self.fake_B_buffer = ReplayBuffer(), ReplayBuffer()    # train.py:125-126

    def set_requires_grad(self, names, flag):
        """utils.py:165-176."""
        for k in names:
            for p in self.params[k]:
                p.requires_grad_(flag)

    def forward_generators(self, real_A, real_B):
        """train.py:173-214."""
        nA, nB = self.nets["A2B"], self.nets["B2A"]
        o = {}
        hf, lf = freq_split(real_A, 10, 8)                                    # train.py:173-175
        lf_feature_A, hf_feature_A, o["fake_B"] = network_a2b(nA, lf, hf)     # train.py:176
        _, _, o["idt_A"] = network_b2a(nB, hf, lf)                            # train.py:180
        o["hf_feature_A"] = hf_feature_A.detach()                             # train.py:183-186
        hf, lf = freq_split(o["fake_B"], 5, 14)                               # train.py:189-191
        o["hf_feature_recovered_A"], _, o["recovered_A"] = network_b2a(nB, hf, lf)   # train.py:193
        hf, lf = freq_split(real_B, 5, 14)                                    # train.py:197-199
        hf_feature_B, _, o["fake_A"] = network_b2a(nB, hf, lf)                # train.py:200
        _, _, o["idt_B"] = network_a2b(nA, lf, hf)                            # train.py:203
        o["hf_feature_B"] = hf_feature_B.detach()                             # train.py:205-208
        hf, lf = freq_split(o["fake_A"], 10, 8)                               # train.py:211-213
        _, o["hf_feature_recovered_B"], o["recovered_B"] = network_a2b(nA, lf, hf)   # train.py:214
        return o

    def generator_loss(self, o, real_A, real_B):
        """train.py:221-236."""
        w = self.w
        ones = torch.ones(real_A.shape[0], dtype=self.dtype)                  # target_real train.py:119
        L = {}
        L["loss_GAN_A2B"] = F.mse_loss(fs_discriminator(self.nets["D_B"], o["fake_B"], "cat"), ones) * w["beta4"]
        L["loss_GAN_B2A"] = F.mse_loss(fs_discriminator(self.nets["D_A"], o["fake_A"], "sum"), ones) * w["beta5"]
        # BCEWithLogits(input=detached feature, target=recovered feature): grad flows through the target
        L["loss_cycle_ABA"] = F.l1_loss(o["recovered_A"], real_A) * w["beta3"] + \
            F.binary_cross_entropy_with_logits(o["hf_feature_A"], o["hf_feature_recovered_A"])
        L["loss_cycle_BAB"] = F.l1_loss(o["recovered_B"], real_B) * w["beta3"] + \
            w["beta1"] * F.binary_cross_entropy_with_logits(o["hf_feature_B"], o["hf_feature_recovered_B"])
        L["loss_idt"] = F.l1_loss(real_A, o["idt_A"]) * w["beta2"] + F.l1_loss(real_B, o["idt_B"]) * w["beta2"]
        loss_G = L["loss_GAN_A2B"] + L["loss_GAN_B2A"] + L["loss_cycle_ABA"] + L["loss_cycle_BAB"] + L["loss_idt"]
        # opt-in extensions (weight 0 => exact reference behaviour): train.py:234 (commented SSIM term)
        if self.ssim_weight:
            L["loss_ssim"] = self.ssim_weight * ((1 - ssim(o["recovered_A"], real_A)) + (1 - ssim(o["recovered_B"], real_B)))
            loss_G = loss_G + L["loss_ssim"]
        if self.whf_weight:
            t = 0
            for rec, real in ((o["recovered_A"], real_A), (o["recovered_B"], real_B)):
                _, yh_r = haar_dwt2(rec, self.dwt_levels)
                _, yh_t = haar_dwt2(real, self.dwt_levels)
                t = t + sum(F.l1_loss(a, b) for a, b in zip(yh_r, yh_t))
            L["loss_whf"] = self.whf_weight * t
            loss_G = loss_G + L["loss_whf"]
        L["loss_G"] = loss_G
        return L

    def discriminator_losses(self, o, real_A, real_B):
        """train.py:247-267 (the two backward calls are issued by the caller)."""
        ones = torch.ones(real_A.shape[0], dtype=self.dtype)
        zeros = torch.zeros(real_A.shape[0], dtype=self.dtype)
        fake_A = self.fake_A_buffer.push_and_pop(o["fake_A"])
        la = (F.mse_loss(fs_discriminator(self.nets["D_A"], real_A, "sum"), ones) +
              F.mse_loss(fs_discriminator(self.nets["D_A"], fake_A.detach(), "sum"), zeros)) * 0.5
        yield "loss_D_A", la
        fake_B = self.fake_B_buffer.push_and_pop(o["fake_B"])
        lb = (F.mse_loss(fs_discriminator(self.nets["D_B"], real_B, "cat"), ones) +
              F.mse_loss(fs_discriminator(self.nets["D_B"], fake_B.detach(), "cat"), zeros)) * 0.5
        yield "loss_D_B", lb

    def train_step(self, real_A, real_B, keep=False, grad_hook=None):
        """One iteration of train.py:166-269.  ``grad_hook(phase, params)`` is called
        between backward and the optimizer step (the build inserts its gradient
        all-reduce there)."""
        real_A = real_A.to(self.dtype)
        real_B = real_B.to(self.dtype)
        o = self.forward_generators(real_A, real_B)
        self.set_requires_grad(["D_A", "D_B"], False)                         # train.py:218
        self.opt_G.zero_grad()
        L = self.generator_loss(o, real_A, real_B)
        L["loss_G"].backward()                                                # train.py:238
        if grad_hook:
            grad_hook("G", self.opt_G.params)
        self.opt_G.step()                                                     # train.py:239
        self.set_requires_grad(["D_A", "D_B"], True)                          # train.py:242
        self.opt_D.zero_grad()
        for name, l in self.discriminator_losses(o, real_A, real_B):
            l.backward()                                                      # train.py:255,267
            L[name] = l
        if grad_hook:
            grad_hook("D", self.opt_D.params)
        self.opt_D.step()                                                     # train.py:269
        out = {k: float(v.detach()) for k, v in L.items()}
        if keep:
            out["tensors"] = {k: v.detach() for k, v in o.items()}
        return out

    def grad_norms(self):
        r = {}
        for k, ps in self.params.items():
            sq = 0.0
            for p in ps:
                if p.grad is not None:
                    sq += float((p.grad.double() ** 2).sum())
            r[k] = math.sqrt(sq)
        return r


def psnr(y, gt, data_range=2.0):
    """skimage.metrics.peak_signal_noise_ratio(y, gt, data_range=2) = 10 log10(4/MSE) (utils.py:209)."""
    mse = float(((y.double() - gt.double()) ** 2).mean())
    return 10.0 * math.log10(data_range ** 2 / mse)


# ------------------------------------------------------------------------------------------------
# input transforms (train.py:129-140) -- the loader's tensor arithmetic, for the device-side pipeline
# ------------------------------------------------------------------------------------------------
def transform_A(img_u8, top, left, size_A=128):
    """transforms_A on one decoded grayscale image (uint8 [H, W]) with the RandomCrop offsets given: ToTensor (/255),
    crop, Resize((2*size_A, 2*size_A), BICUBIC), Normalize(0.5, 0.5).  torchvision is absent offline; its tensor ``Resize`` is
    ``torch.nn.functional.interpolate(mode='bicubic', align_corners=False)`` (torchvision/transforms/_functional_tensor.py ``resize``;
    antialiasing does not act on an upscale), which is what runs here."""
    x = torch.as_tensor(img_u8).to(torch.float32).div(255.0)[None, None]
    x = x[:, :, top:top + size_A, left:left + size_A]
    x = torch.nn.functional.interpolate(x, size=(2 * size_A, 2 * size_A), mode="bicubic", align_corners=False)
    return ((x - 0.5) / 0.5)[0]


def transform_B(img_u8, top, left, size_B=256):
    """transforms_B: ToTensor, Normalize(0.5, 0.5), RandomCrop(size_B)."""
    x = torch.as_tensor(img_u8).to(torch.float32).div(255.0)[None]
    x = (x - 0.5) / 0.5
    return x[:, top:top + size_B, left:left + size_B]


# --------------------------------------------------------------------------
# evaluation metrics (utils.py:209-212).  skimage is absent offline: these follow its PUBLISHED definitions (scikit-image
# metrics module: peak_signal_noise_ratio, structural_similarity with its defaults, mean_squared_error,
# normalized_mutual_information) and are pinned by closed-form cases only -- "parity unpinned" for this row.
# --------------------------------------------------------------------------

def skimage_mse(y, gt):
    return float(np.mean((np.asarray(y, np.float64) - np.asarray(gt, np.float64)) ** 2))


def skimage_psnr(y, gt, data_range=2.0):
    err = skimage_mse(y, gt)
    return float("inf") if err == 0 else 10.0 * math.log10(data_range ** 2 / err)


def skimage_nmi(a, b, bins=100):
    """(H(a) + H(b)) / H(a, b) on the joint ``bins`` x ``bins`` histogram (numpy.histogram2d over each image's [min, max])."""
    # float64 samples: the bin edges are then linspace(min, max, bins + 1) in float64, which is what the NumPy 1.x of the
    # reference's era computed for float32 images too (NumPy 2 keeps float32 edges for float32 samples: a few pixels per
    # image change bins, NMI moves by ~2e-6)
    h, _, _ = np.histogram2d(np.ravel(a).astype(np.float64), np.ravel(b).astype(np.float64), bins=bins)

    def ent(p):
        p = p[p > 0] / p.sum()
        return float(-(p * np.log(p)).sum())
    hj = ent(h.ravel())
    return (ent(h.sum(1)) + ent(h.sum(0))) / hj if hj > 0 else 1.0


def skimage_ssim(a, b, data_range=2.0, win=7):
    """structural_similarity defaults for 2-D float images: 7x7 uniform window, K1 0.01, K2 0.03, sample covariance, mean over
    the map cropped by (win-1)/2."""
    from scipy.ndimage import uniform_filter
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    NP = win * win
    cov_norm = NP / (NP - 1.0)
    ux, uy = uniform_filter(a, win), uniform_filter(b, win)
    uxx, uyy, uxy = uniform_filter(a * a, win), uniform_filter(b * b, win), uniform_filter(a * b, win)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    p = (win - 1) // 2
    return float(S[p:-p, p:-p].mean())

"""SSIM behind the reference's ``ssim.py`` interface (ssim.py:7-73) on one fused separable HIP kernel."""
from math import exp

import torch

from . import ops


def gaussian(window_size, sigma):
    """ssim.py:7-9."""
    gauss = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return gauss / gauss.sum()


def create_window(window_size, channel):
    """ssim.py:11-15 (the kernel uses the separable 1-D taps; this 2-D window is kept for API parity)."""
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def _check(window_size):
    if window_size != 11:
        raise NotImplementedError("the fused kernel is built for the reference's 11-tap, sigma 1.5 window (ssim.py:40)")


def ssim(img1, img2, window_size=11, size_average=True):
    """ssim.py:65-73."""
    _check(window_size)
    return ops.ssim(img1, img2, size_average)


class SSIM(torch.nn.Module):
    """ssim.py:39-63."""

    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        _check(window_size)
        self.window_size, self.size_average, self.channel = window_size, size_average, 1
        self.window = create_window(window_size, self.channel)

    def forward(self, img1, img2):
        return ops.ssim(img1, img2, self.size_average)

"""Inference / evaluation path of the reference (utils.py:182-242 `eval`, `eval_6m`).

``super_resolve`` is the reference's inference recipe (utils.py:202-205): frequency split with radii (10, 8), generator
forward in eval mode (BatchNorm on running statistics, executed by the HIP kernels).  ``evaluate_pairs`` reproduces the
metric loop: like the reference it moves each output to the host and scores it there; skimage is not available offline, so
the four skimage metrics are restated in numpy from their published definitions (defaults of the calls at utils.py:209-212).
Dataset / PNG IO is out of scope: the caller supplies tensors.
"""
import math

import numpy as np
import torch

from . import ops


@torch.no_grad()
def super_resolve(model, lr_img, r_hp=10, r_lp=8):
    """lr_img (B,1,H,W) in [-1,1] -> super-resolved (B,1,H,W).  Leaves the model in eval mode, as utils.eval does."""
    model.eval()
    hf, lf = ops.freq_split(lr_img, r_hp, r_lp)
    return model(lf, hf)[2]


def psnr(y, gt, data_range=2.0):
    """skimage.metrics.peak_signal_noise_ratio = 10 log10(data_range^2 / MSE)."""
    err = float(np.mean((np.asarray(y, np.float64) - np.asarray(gt, np.float64)) ** 2))
    return 10.0 * math.log10(data_range ** 2 / err)


def mse(y, gt):
    return float(np.mean((np.asarray(y, np.float64) - np.asarray(gt, np.float64)) ** 2))


def nmi(a, b, bins=100):
    """skimage.metrics.normalized_mutual_information: (H(a) + H(b)) / H(a, b), entropies (natural log... base cancels) of the
    joint ``bins`` x ``bins`` histogram and its marginals."""
    h, _, _ = np.histogram2d(np.ravel(a), np.ravel(b), bins=bins)
    def ent(p):
        p = p[p > 0] / p.sum()
        return float(-(p * np.log(p)).sum())
    return (ent(h.sum(1)) + ent(h.sum(0))) / ent(h.ravel())


def ssim_skimage(a, b, data_range=2.0, win=7):
    """skimage.metrics.structural_similarity defaults for 2-D float images: 7x7 uniform window, K1 0.01, K2 0.03, sample
    covariance, mean over the map cropped by (win-1)/2."""
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


def evaluate_pairs(model, pairs):
    """pairs: iterable of (lr (1,1,H,W), hr (1,1,H,W)) device tensors.  Returns mean PSNR / SSIM / MSE / NMI like the print
    at utils.py:214,242."""
    tot = {"psnr": 0.0, "ssim": 0.0, "mse": 0.0, "nmi": 0.0}
    n = 0
    for lr, hr in pairs:
        y = super_resolve(model, lr).cpu().numpy().squeeze(0).squeeze(0)
        g = hr.cpu().numpy().squeeze(0).squeeze(0)
        tot["psnr"] += psnr(y, g, 2.0)
        tot["ssim"] += ssim_skimage(y, g, 2.0)
        tot["mse"] += mse(y, g)
        tot["nmi"] += nmi(y, g)
        n += 1
    return {k: v / max(n, 1) for k, v in tot.items()}

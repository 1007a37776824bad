"""Inference / evaluation path of the reference (utils.py:182-242 `eval`, `eval_6m`), on the device.

``super_resolve`` is the reference's inference recipe (utils.py:202-205): frequency split with radii (10, 8), generator forward
in eval mode.  `model.eval()` turns every BatchNorm2d into a per-channel affine map; the forward folds it into the preceding
convolution's weights (``faoctasr_bn_fold``), so the inference generator is convolutions + activations only.
``image_metrics`` / ``evaluate_pairs`` score outputs with the four skimage metrics of utils.py:209-212 -- PSNR, SSIM (7x7 uniform
window), MSE, NMI (100 x 100 joint histogram) -- computed by HIP kernels (``faoctasr_eval_metrics``); the reference copies every
image to the host for skimage.  Only the N x 4 results travel.  Dataset / PNG IO is out of scope: the caller supplies tensors.
"""
import torch

from . import _lib, ops
from ._lib import call, ptr, stream_ptr


@torch.no_grad()
def super_resolve(model, lr_img, r_hp=10, r_lp=8):
    """lr_img (B,1,H,W) in [-1,1] -> super-resolved (B,1,H,W).  Leaves the model in eval mode, as utils.eval does."""
    model.eval()
    hf, lf = ops.freq_split(lr_img, r_hp, r_lp)
    return model(lf, hf)[2]


@torch.no_grad()
def image_metrics(y, gt, data_range=2.0, bins=100):
    """y, gt: (N,1,H,W) or (N,H,W) device tensors -> float64 tensor (N,4) on the device: PSNR, SSIM, MSE, NMI per image pair
    (skimage.metrics.peak_signal_noise_ratio(data_range=2) / structural_similarity / mean_squared_error /
    normalized_mutual_information with their defaults, utils.py:209-212)."""
    y, gt = ops._c(y), ops._c(gt)
    if y.shape != gt.shape:
        raise _lib.KernelError("image_metrics: shapes differ: %s vs %s" % (tuple(y.shape), tuple(gt.shape)))
    if y.dim() == 4:
        if y.shape[1] != 1:
            raise _lib.KernelError("image_metrics: single-channel images expected")
        N, _, H, W = y.shape
    else:
        N, H, W = y.shape
    out = torch.empty((N, 4), dtype=torch.float64, device=y.device)
    nbytes = _lib.load().faoctasr_eval_workspace_bytes(N, bins)
    ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=y.device)
    call("eval_metrics", ptr(y), ptr(gt), out.data_ptr(), ws.data_ptr(), N, H, W, float(data_range), int(bins), stream_ptr())
    return out


def evaluate_pairs(model, pairs):
    """pairs: iterable of (lr (B,1,H,W), hr (B,1,H,W)) device tensors.  Returns mean PSNR / SSIM / MSE / NMI like the print at
    utils.py:214,242; one host read at the end."""
    acc, n = None, 0
    for lr, hr in pairs:
        m = image_metrics(super_resolve(model, lr), hr).sum(0)
        acc = m if acc is None else acc + m
        n += lr.shape[0]
    if acc is None:
        return {"psnr": 0.0, "ssim": 0.0, "mse": 0.0, "nmi": 0.0}
    vals = (acc / n).tolist()
    return dict(zip(("psnr", "ssim", "mse", "nmi"), vals))

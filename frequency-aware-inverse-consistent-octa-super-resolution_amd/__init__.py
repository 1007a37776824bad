"""MI355X-native train step for frequency-aware inverse-consistent OCTA super-resolution.

Drop-in for the reference's hot path (its ``model.py`` / ``utils.py`` / ``ssim.py`` /
``pytorch_wavelets`` callables and the loop body of ``train.py``), executed by hand-written
HIP kernels for gfx950 behind the C ABI of ``include/faoctasr.h``.
"""
from . import _lib, ops
from ._lib import KernelError
from .model import (Discriminator, FS_DiscriminatorA, FS_DiscriminatorB, NetworkA2B, NetworkB2A, ResidualBlock, ResnetBlock,
                    ResnetGenerator, TVLoss, UnetGenerator, UnetSkipConnectionBlock, shallowNet)
from .evaluate import evaluate_pairs, image_metrics, super_resolve
from . import ssim                # stays the MODULE: the reference does `import ssim; ssim.SSIM()` (train.py:24,97)
from .ssim import SSIM
from .ssim import ssim as ssim_fn  # the function ssim.py:65-73; not exported under the submodule's name
from .data import GpuTransformA, GpuTransformB, crop_resize_normalize, random_crop_offsets
from .train import GraphedTrainStep, ParamArena, TrainStep, live_parameters
from .utils import (DeviceReplayBuffer, LambdaLR, ReplayBuffer, frequency_split, high_pass, low_pass, psnr, set_requires_grad, weights_init_normal)
from .wavelets import AFB2D, SFB2D, DWTForward, DWTInverse

__all__ = [n for n in dir() if not n.startswith("_")]

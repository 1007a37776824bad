"""Device-side input pipeline: the tensor transforms of the reference's loader, train.py:129-140, fused into
one HIP pass so decoded images can feed the train step at GPU rate.

  transforms_A = [ToTensor, RandomCrop(size_A), Resize(2*size_A, BICUBIC), Normalize(0.5, 0.5)]      -> ``GpuTransformA``
  transforms_B = [ToTensor, Normalize(0.5, 0.5), RandomCrop(size_B)]                                  -> ``GpuTransformB``

Decoding (PIL ``Image.open(..).convert('L')``, dataset.py:24-31) stays on the host and out of scope; the transforms take the decoded
uint8 planes as one device tensor [N, H, W].  Crop offsets are drawn like torchvision's ``RandomCrop.get_params``
(``torch.randint(0, h - th + 1, (1,))`` for the top, then the left, per image) so a seeded run picks the reference's crops.
"""
import torch

from . import _lib
from ._lib import call, ptr, stream_ptr


def random_crop_offsets(n, h, w, size, generator=None):
    """torchvision.transforms.RandomCrop.get_params per image: top then left, one ``torch.randint`` each."""
    tops, lefts = [], []
    for _ in range(n):
        if h == size and w == size:
            tops.append(0); lefts.append(0)
            continue
        tops.append(int(torch.randint(0, h - size + 1, size=(1,), generator=generator)))
        lefts.append(int(torch.randint(0, w - size + 1, size=(1,), generator=generator)))
    return tops, lefts


def crop_resize_normalize(img_u8, tops, lefts, crop, out_size, mean=0.5, std=0.5):
    """img_u8 [N, H, W] uint8 on the GPU -> [N, 1, out_size, out_size] fp32 = (bicubic(crop(img)/255) - mean) / std."""
    if not img_u8.is_cuda:
        raise _lib.KernelError("crop_resize_normalize needs a CUDA/HIP tensor (no CPU fallback), got %s" % img_u8.device)
    if img_u8.dtype != torch.uint8 or img_u8.dim() != 3:
        raise ValueError("expected a uint8 tensor [N, H, W], got %s %s" % (img_u8.dtype, tuple(img_u8.shape)))
    img_u8 = img_u8.contiguous()
    n, h, w = img_u8.shape
    if crop > h or crop > w:
        raise ValueError("Required crop size (%d, %d) is larger than input image size (%d, %d)" % (crop, crop, h, w))
    t = torch.as_tensor(tops, dtype=torch.int32).to(img_u8.device)
    l = torch.as_tensor(lefts, dtype=torch.int32).to(img_u8.device)
    if t.numel() != n or l.numel() != n:
        raise ValueError("need one crop offset per image")
    out = torch.empty((n, 1, out_size, out_size), dtype=torch.float32, device=img_u8.device)
    call("prep_crop_resize", img_u8.data_ptr(), t.data_ptr(), l.data_ptr(), ptr(out), n, h, w, crop, out_size, float(mean), float(std), stream_ptr())
    return out


class GpuTransformA:
    """transforms_A of train.py:129-134 for a decoded uint8 batch: random ``size_A`` crop, bicubic x2, normalise to [-1, 1]."""

    def __init__(self, size_A=128, generator=None):
        self.size, self.generator = size_A, generator

    def __call__(self, img_u8):
        n, h, w = img_u8.shape
        tops, lefts = random_crop_offsets(n, h, w, self.size, self.generator)
        return crop_resize_normalize(img_u8, tops, lefts, self.size, 2 * self.size)


class GpuTransformB:
    """transforms_B of train.py:136-140: normalise, random ``size_B`` crop (the two commute)."""

    def __init__(self, size_B=256, generator=None):
        self.size, self.generator = size_B, generator

    def __call__(self, img_u8):
        n, h, w = img_u8.shape
        tops, lefts = random_crop_offsets(n, h, w, self.size, self.generator)
        return crop_resize_normalize(img_u8, tops, lefts, self.size, self.size)

"""Training utilities behind the reference's ``utils.py`` interface (utils.py:31-117,165-176).

``high_pass`` / ``low_pass`` keep the reference signatures (``timg`` of shape (1,H,W) -> (H,W)) but
run as circulant MFMA GEMMs with cached matrices instead of a per-call Python mask loop + FFT
(DESIGN.md "frequency split"); gradients flow through them as in the reference.
"""
import random

import torch

from . import ops


class ReplayBuffer:
    """utils.py:31-51: 50-image history; per element one ``random.uniform`` draw and, when it exceeds
    0.5 on a full buffer, one ``random.randint`` draw.  Images stay on the device."""

    def __init__(self, max_size=50):
        assert max_size > 0, "Empty buffer or trying to create a black hole. Be careful."
        self.max_size = max_size
        self.data = []

    def push_and_pop(self, data):
        to_return = []
        for element in data.detach():
            element = torch.unsqueeze(element, 0)
            if len(self.data) < self.max_size:
                self.data.append(element)
                to_return.append(element)
            elif random.uniform(0, 1) > 0.5:
                i = random.randint(0, self.max_size - 1)
                to_return.append(self.data[i].clone())
                self.data[i] = element
            else:
                to_return.append(element)
        return torch.cat(to_return)


class DeviceReplayBuffer:
    """``ReplayBuffer`` with the history in ONE device tensor (the replay buffer of a hipGraph-captured step).

    The decisions are the reference's (utils.py:37-51): the same ``random.uniform`` / ``random.randint`` draws in the same order,
    made on the host by ``plan``; ``apply`` is then a fixed-shape gather + scatter driven by two index tensors, so the device
    work is identical from call to call and can be captured.  Row ``max_size`` of the store is a trash row for elements that are
    not kept.  Index convention of ``plan``: source ``i < max_size`` = history row ``i``; ``max_size + 1 + j`` = element ``j``
    of the incoming batch (an earlier element of the same call that was just written to the row being read)."""

    def __init__(self, max_size=50):
        assert max_size > 0, "Empty buffer or trying to create a black hole. Be careful."
        self.max_size = max_size
        self.count = 0
        self.store = None

    @classmethod
    def adopt(cls, host_buffer):
        """Continue a host-side ``ReplayBuffer`` (its images move into the device store)."""
        if isinstance(host_buffer, cls):
            return host_buffer
        b = cls(host_buffer.max_size)
        if host_buffer.data:
            b._ensure(host_buffer.data[0])
            b.store[:len(host_buffer.data)].copy_(torch.cat(host_buffer.data))
            b.count = len(host_buffer.data)
        return b

    def _ensure(self, like):
        if self.store is None:
            self.store = torch.zeros((self.max_size + 1,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)

    def plan(self, batch):
        """Host decisions for one call: (source indices, destination rows), two lists of length ``batch``."""
        src, writer = [], {}
        for j in range(batch):
            own = self.max_size + 1 + j
            if self.count < self.max_size:
                writer[self.count] = j
                self.count += 1
                src.append(own)
            elif random.uniform(0, 1) > 0.5:
                i = random.randint(0, self.max_size - 1)
                src.append(self.max_size + 1 + writer[i] if i in writer else i)
                writer[i] = j
            else:
                src.append(own)
        dst = [self.max_size] * batch                       # trash row unless this element is the last writer of a row
        for row, j in writer.items():
            dst[j] = row
        return src, dst

    def apply(self, data, src, dst):
        """Device part: ``src`` / ``dst`` int64 tensors of length batch (see ``plan``)."""
        data = data.detach()
        self._ensure(data)
        pool = torch.cat([self.store, data])
        out = pool.index_select(0, src)
        self.store.index_copy_(0, dst, data)
        return out

    def push_and_pop(self, data):
        src, dst = self.plan(data.shape[0])
        dev = data.device
        return self.apply(data, torch.tensor(src, dtype=torch.long, device=dev), torch.tensor(dst, dtype=torch.long, device=dev))

    @property
    def data(self):
        """The history as the reference's list of (1, C, H, W) tensors."""
        return [] if self.store is None else [self.store[i:i + 1] for i in range(self.count)]


class LambdaLR:
    """utils.py:53-61."""

    def __init__(self, n_epochs, offset, decay_start_epoch):
        assert (n_epochs - decay_start_epoch) > 0, "Decay must start before the training session ends!"
        self.n_epochs, self.offset, self.decay_start_epoch = n_epochs, offset, decay_start_epoch

    def step(self, epoch):
        return 1.0 - max(0, epoch + self.offset - self.decay_start_epoch) / (self.n_epochs - self.decay_start_epoch)


def weights_init_normal(m):
    """utils.py:63-69: Conv*: weight ~ N(0, 0.02) (bias untouched); BatchNorm2d: weight ~ N(1, 0.02), bias 0."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1 and hasattr(m, "weight"):
        torch.nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find("BatchNorm2d") != -1:
        torch.nn.init.normal_(m.weight.data, 1.0, 0.02)
        torch.nn.init.constant_(m.bias.data, 0.0)


def set_requires_grad(nets, requires_grad=False):
    """utils.py:165-176."""
    if not isinstance(nets, list):
        nets = [nets]
    for net in nets:
        if net is not None:
            for param in net.parameters():
                param.requires_grad = requires_grad


def _as_batch(timg):
    if timg.dim() != 3:
        raise ValueError("expected a (C,H,W) image like the reference's `real_A[0]`, got shape %s" % (tuple(timg.shape),))
    return timg[0].reshape(1, 1, timg.shape[1], timg.shape[2])


def high_pass(timg, i=4):
    """utils.py:93-103: |ifft2(ifftshift(fftshift(fft2(timg[0])) * (1 - gauss_i)))| -> (H,W)."""
    x = _as_batch(timg)
    hf, _ = ops.freq_split(x, i, i)           # hf = (|x - low| + x) / 2
    return ops.axpby(hf, x, 2.0, -1.0)[0, 0]


def low_pass(timg, i=10):
    """utils.py:105-117: -|ifft2(ifftshift(fftshift(fft2(timg[0])) * gauss_i))| -> (H,W)."""
    _, lf = ops.freq_split(_as_batch(timg), i, i)
    return lf[0, 0]


def frequency_split(x, r_hp, r_lp):
    """Batched form of train.py:173-175: per sample hf = (high_pass(x_b, r_hp) + x_b)/2, lf = low_pass(x_b, r_lp)."""
    return ops.freq_split(x, r_hp, r_lp)


def psnr(y, gt, data_range=2.0):
    """skimage.metrics.peak_signal_noise_ratio(y, gt, data_range=2) = 10 log10(4 / MSE) (utils.py:209)."""
    mse = ops.mse_loss(y, gt)
    return 10.0 * torch.log10(data_range ** 2 / mse)

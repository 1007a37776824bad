"""Layer modules over the HIP operators, parameter-compatible with ``torch.nn``.

Parameter / buffer names and shapes equal those of the ``torch.nn`` layers the reference's
``model.py`` instantiates, so ``state_dict`` keys match and reference ``.pth`` files load
(checkpoint row).  Class names contain 'Conv' / 'BatchNorm2d' on purpose:
``utils.weights_init_normal`` dispatches on ``__class__.__name__`` (utils.py:63-69).
"""
import math

import torch
import torch.nn as nn

from . import ops
from ._lib import call, ptr, stream_ptr


def folded_conv_params(conv, bn):
    """(weight, bias) of ``conv`` with the eval-mode BatchNorm ``bn`` folded in (``faoctasr_bn_fold``): what `model.eval()`
    (utils.py:186) turns conv -> BN into.  Cached on the conv module until its weights or the BN's statistics change."""
    key = (ops._ver(conv.weight), ops._ver(bn.weight), ops._ver(bn.bias), bn._stat_epoch, None if conv.bias is None else ops._ver(conv.bias))
    hit = getattr(conv, "_folded", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    w = conv.weight.detach()
    wf = torch.empty_like(w)
    bf = torch.empty(bn.num_features, dtype=torch.float32, device=w.device)
    transposed = isinstance(conv, ConvTranspose2d)
    M = conv.out_channels
    kk = conv.kernel_size * conv.kernel_size
    call("bn_fold", ptr(w), ptr(conv.bias.detach()) if conv.bias is not None else None, ptr(bn.weight.detach()), ptr(bn.bias.detach()),
         ptr(bn.running_mean), ptr(bn.running_var), bn.eps, ptr(wf), ptr(bf), M, kk if transposed else conv.in_channels * kk,
         1 if transposed else 0, conv.in_channels if transposed else 1, stream_ptr())
    conv._folded = (key, wf, bf)
    return wf, bf


class Conv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # torch.nn.Conv2d default: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias
        bound = 1.0 / math.sqrt(self.in_channels * self.kernel_size * self.kernel_size)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def forward(self, x, act=None, slope=0.2, reflect_pad=0, params=None, link=None):
        w, b = params if params is not None else (self.weight, self.bias)
        if reflect_pad:
            return ops.conv2d(x, w, b, self.stride, reflect_pad, True, act, slope, link)
        return ops.conv2d(x, w, b, self.stride, self.padding, False, act, slope, link)

    def extra_repr(self):
        return "%d, %d, kernel_size=%d, stride=%d, padding=%d, bias=%s" % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding, self.bias is not None)


class ConvTranspose2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding, self.output_padding = kernel_size, stride, padding, output_padding
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        bound = 1.0 / math.sqrt(out_channels * kernel_size * kernel_size)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def forward(self, x, act=None, slope=0.2, params=None):
        w, b = params if params is not None else (self.weight, self.bias)
        return ops.conv_transpose2d(x, w, b, self.stride, self.padding, self.output_padding, act, slope)


class BatchNorm2d(nn.Module):
    """nn.BatchNorm2d (affine, track_running_stats).  Training: batch statistics, fused activation / residual; eval: running
    statistics (the inference path).  ``num_batches_tracked`` is counted on the host and written into the buffer when the
    state_dict is taken (one tiny device kernel per BN call otherwise: 243 per train step)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._pending_batches = 0
        self._stat_epoch = 0          # bumped whenever the running statistics change (they change through raw pointers)

    def _flush_counter(self):
        if self._pending_batches:
            self.num_batches_tracked += self._pending_batches
            self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_counter()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending_batches = 0
        self._stat_epoch += 1
        super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x, act=None, slope=0.2, residual=None, link=None):
        if not self.training:
            y = ops.batchnorm_eval(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, None if residual is not None else act,
                                   slope)
            if residual is not None:
                y = ops.add(y, residual)
                if act:
                    y = ops.activation(y, act, slope)
            return y
        d = self.__dict__                     # (plain counters: nn.Module.__setattr__ costs ~2.5 us a write, 486 writes per step)
        d["_pending_batches"] += 1
        d["_stat_epoch"] += 1
        return ops.batchnorm_train(x, self.weight, self.bias, self.running_mean, self.running_var, self.momentum, self.eps, act, slope,
                                   residual, link)


class InstanceNorm2d(nn.Module):
    """Not used by the reference (model.py:134,184 ignore norm_layer='Instance'); provided because north_star names it."""

    def __init__(self, num_features, eps=1e-5, affine=False):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.weight = nn.Parameter(torch.ones(num_features)) if affine else None
        self.bias = nn.Parameter(torch.zeros(num_features)) if affine else None

    def forward(self, x, act=None, slope=0.2):
        return ops.instance_norm(x, self.weight, self.bias, self.eps, act, slope)


class ReLU(nn.Module):
    def __init__(self, inplace=False):
        super().__init__()

    act, slope = "relu", 0.0

    def forward(self, x):
        return ops.activation(x, "relu")


class LeakyReLU(nn.Module):
    def __init__(self, negative_slope=0.01, inplace=False):
        super().__init__()
        self.slope = negative_slope

    act = "lrelu"

    def forward(self, x):
        return ops.activation(x, "lrelu", self.slope)


class Tanh(nn.Module):
    act, slope = "tanh", 0.0

    def forward(self, x):
        return ops.activation(x, "tanh")


class ReflectionPad2d(nn.Module):
    """Only exists fused into the following convolution's gather (model.py:450-451,472-473)."""

    def __init__(self, padding):
        super().__init__()
        self.padding = padding

    def forward(self, x):
        raise NotImplementedError("ReflectionPad2d is folded into the next Conv2d by FusedSequential")


_ACTS = (ReLU, LeakyReLU, Tanh)


class FusedSequential(nn.Sequential):
    """nn.Sequential with the same child indices (hence the same state_dict keys) whose forward
    runs peephole-fused kernels: [ReflectionPad2d] Conv [act] | BatchNorm [act] | ResBlock [ReLU]."""

    def forward(self, x, start=0):
        mods = list(self)
        i, n = start, len(mods)
        while i < n:
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < n else None
            if isinstance(m, ReflectionPad2d) and isinstance(nxt, Conv2d) and nxt.padding == 0:
                after = mods[i + 2] if i + 2 < n else None
                if isinstance(after, BatchNorm2d) and not after.training and not torch.is_grad_enabled():
                    act = mods[i + 3] if i + 3 < n else None
                    if isinstance(act, _ACTS):
                        x = nxt(x, act=act.act, slope=act.slope, reflect_pad=m.padding, params=folded_conv_params(nxt, after))
                        i += 4
                    else:
                        x = nxt(x, reflect_pad=m.padding, params=folded_conv_params(nxt, after))
                        i += 3
                    continue
                if isinstance(after, (LeakyReLU, Tanh)):
                    x = nxt(x, act=after.act, slope=after.slope, reflect_pad=m.padding)
                    i += 3
                else:
                    x = nxt(x, reflect_pad=m.padding)
                    i += 2
            elif isinstance(m, (Conv2d, ConvTranspose2d)) and isinstance(nxt, BatchNorm2d) and not nxt.training and not torch.is_grad_enabled():
                # inference: conv -> BN(eval) [-> act] is ONE convolution with folded weights (utils.py:186 `model.eval()`)
                after = mods[i + 2] if i + 2 < n else None
                if isinstance(after, _ACTS):
                    x = m(x, act=after.act, slope=after.slope, params=folded_conv_params(m, nxt))
                    i += 3
                else:
                    x = m(x, params=folded_conv_params(m, nxt))
                    i += 2
            elif isinstance(m, (Conv2d, ConvTranspose2d)):
                if isinstance(nxt, _ACTS):
                    x = m(x, act=nxt.act, slope=nxt.slope)
                    i += 2
                else:
                    x = m(x)
                    i += 1
            elif isinstance(m, (BatchNorm2d, InstanceNorm2d)):
                if isinstance(nxt, _ACTS):
                    x = m(x, act=nxt.act, slope=nxt.slope)
                    i += 2
                else:
                    x = m(x)
                    i += 1
            elif getattr(m, "is_residual_block", False) and isinstance(nxt, ReLU):
                x = m(x, post_act="relu")
                i += 2
            else:
                x = m(x)
                i += 1
        return x

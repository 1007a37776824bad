"""Generators and discriminators of the frequency-aware OCTA super-resolution CycleGAN on the
MI355X kernels, behind the reference's ``model.py`` interface.

Same class names, constructor arguments, forward signatures, return structures and
``state_dict`` keys as /root/reference/model.py (cited per class); every layer executes as a
hand-written HIP kernel (see ops.py).  Quirks of the reference are kept because parity is
judged against it: BatchNorm2d everywhere (``norm_layer='Instance'`` is ignored, model.py:134,184),
8 ResidualBlocks + 3 ResnetBlocks, 5-layer 4x4 PatchGAN with an image and a Haar-DWT branch,
dead ``unet`` / ``unet_up`` / ``skip`` parameters that exist only as state_dict entries.
"""
import torch
import torch.nn as nn

from . import ops
from .layers import (BatchNorm2d, Conv2d, ConvTranspose2d, FusedSequential, LeakyReLU, ReflectionPad2d, ReLU, Tanh, folded_conv_params)
from .wavelets import DWTForward


class TVLoss(nn.Module):
    """model.py:17-33.  A dead value on the train step (train.py:178 computes it, nothing reads it);
    kept as host-side tensor plumbing, not a kernel."""

    def __init__(self, TVLoss_weight=1):
        super().__init__()
        self.TVLoss_weight = TVLoss_weight

    def forward(self, x):
        b, c, h, w = x.shape
        count_h, count_w = c * (h - 1) * w, c * h * (w - 1)
        h_tv = ((x[:, :, 1:, :] - x[:, :, :h - 1, :]) ** 2).sum()
        w_tv = ((x[:, :, :, 1:] - x[:, :, :, :w - 1]) ** 2).sum()
        return self.TVLoss_weight * 2 * (h_tv / count_h + w_tv / count_w) / b


class Discriminator(nn.Module):
    """PatchGAN of model.py:86-127: 4x4 convs with bias, five stride-2 stages then two stride-1,
    BatchNorm2d + LeakyReLU(0.2) from the second conv on."""

    def __init__(self, input_nc=1, ndf=64, n_layers=5, norm_layer=BatchNorm2d):
        super().__init__()
        seq = [Conv2d(input_nc, ndf, 4, 2, 1), LeakyReLU(0.2, True)]
        prev = 1
        for n in range(1, n_layers):
            mult = min(2 ** n, 8)
            seq += [Conv2d(ndf * prev, ndf * mult, 4, 2, 1, bias=True), norm_layer(ndf * mult), LeakyReLU(0.2, True)]
            prev = mult
        mult = min(2 ** n_layers, 8)
        seq += [Conv2d(ndf * prev, ndf * mult, 4, 1, 1, bias=True), norm_layer(ndf * mult), LeakyReLU(0.2, True)]
        seq += [Conv2d(ndf * mult, 1, 4, 1, 1)]
        self.model = FusedSequential(*seq)

    def forward(self, input):
        return self.model(input)


class _FSDiscriminator(nn.Module):
    """Shared body of FS_DiscriminatorA / FS_DiscriminatorB (model.py:132-235)."""

    def __init__(self, cs):
        super().__init__()
        self.wgan = False
        self.DWT2 = DWTForward(J=1, wave="haar", mode="reflect")
        self.filter = self.filter_wavelet
        self.cs = cs
        self.net = Discriminator(input_nc=1)
        self.net_dwt = Discriminator(input_nc=1 if cs == "sum" else 3)
        self.out_net = nn.Softmax()      # constructed and unused in the reference too (model.py:152,204)

    def forward(self, x, y=None):
        dwt, ximg = self.filter(x)
        # global average pool of both heads and the 0.7/0.3 mix in one kernel (model.py:158-164)
        return ops.mean_mix(self.net(ximg), self.net_dwt(dwt), 0.7, 0.3)

    def _bands(self, x, norm=True):
        ll, hc = self.DWT2(x)
        lh, hl, hh = hc[0][:, :, 0], hc[0][:, :, 1], hc[0][:, :, 2]
        if norm:
            lh, hl, hh = lh * 0.5 + 0.5, hl * 0.5 + 0.5, hh * 0.5 + 0.5
        return ll, lh, hl, hh


class FS_DiscriminatorA(_FSDiscriminator):
    """model.py:132-179.  The first positional argument is ``recursions`` (train.py:75 passes
    input_nc there); with cs='sum' the wavelet branch sees the raw LL band."""

    def __init__(self, recursions=1, stride=1, kernel_size=5, wgan=False, highpass=True, D_arch="FSD", norm_layer="Instance",
                 filter_type="gau", cs="sum"):
        super().__init__(cs)

    def filter_wavelet(self, x, norm=True):
        c = self.cs.lower()
        if c == "sum" and x.shape[1] == 1:
            return ops.haar_dfront(x, 0), x                       # fused LL front end
        if c == "cat" and norm and x.shape[1] == 1:
            return ops.haar_dfront(x, 1), x
        ll, lh, hl, hh = self._bands(x, norm)
        if c == "sum":
            return ll, x
        if c == "each":
            return ll, lh, hl, hh, x
        if c == "cat":
            return torch.cat((lh, hl, hh), 1), x
        raise NotImplementedError("Wavelet format [{:s}] not recognized".format(self.cs))


class FS_DiscriminatorB(_FSDiscriminator):
    """model.py:182-235.  With cs='cat' the wavelet branch sees cat(LH,HL,HH)*0.5+0.5 (3 channels);
    with cs='sum' this variant returns HH (model.py:226-227)."""

    def __init__(self, recursions=1, stride=1, kernel_size=5, wgan=False, highpass=True, D_arch="FSD", norm_layer="Instance",
                 filter_type="gau", cs="cat"):
        super().__init__(cs)

    def filter_wavelet(self, x, norm=True):
        c = self.cs.lower()
        if c == "cat" and norm and x.shape[1] == 1:
            return ops.haar_dfront(x, 1), x                       # fused band-concat + normalise front end
        ll, lh, hl, hh = self._bands(x, norm)
        if c == "sum":
            return hh, x
        if c == "each":
            return ll, lh, hl, hh, x
        if c == "cat":
            return torch.cat((lh, hl, hh), 1), x
        raise NotImplementedError("Wavelet format [{:s}] not recognized".format(self.cs))


def _shallow_frequency(use_bias):
    """model.py:242-246 / 275-279."""
    return FusedSequential(Conv2d(1, 64, 4, 2, 1, bias=use_bias), LeakyReLU(0.2, True),
                           Conv2d(64, 128, 3, 1, 1, bias=use_bias), BatchNorm2d(128), ReLU(True),
                           Conv2d(128, 64, 3, 1, 1, bias=use_bias), BatchNorm2d(64))


def _skip(use_bias):
    """model.py:249-252 / 281-284."""
    return FusedSequential(ReLU(True), Conv2d(128, 64, 3, 1, 1, bias=use_bias), BatchNorm2d(64))


class NetworkA2B(nn.Module):
    """model.py:238-268.  forward(lf, hf) -> (lf_feature, hf_feature, out)."""

    def __init__(self, use_bias=False):
        super().__init__()
        self.unet = UnetGenerator(input_nc=64, output_nc=64, num_downs=7)          # never called (model.py:262-268)
        self.shallow_frequency = _shallow_frequency(use_bias)
        self.shallow_up = shallowNet(up=True)
        self.skip = _skip(use_bias)
        self.unet_up = FusedSequential(ReLU(True), ConvTranspose2d(128, 64, 4, 2, 1, bias=use_bias), BatchNorm2d(64))   # never called
        self.A2B_input = FusedSequential(Conv2d(1, 64, 4, 2, 1, bias=use_bias))
        self.resnet = ResnetGenerator(input_nc=64, output_nc=64, n_blocks=8)

    def forward(self, lf, hf):
        lf_feature = self.shallow_frequency(lf)
        hf_feature_input = self.A2B_input(hf)
        # skip = ReLU -> conv -> BN on cat([input, resnet(input)]): the cat and the ReLU are one kernel
        hf_feature = self.skip(ops.cat2_act(hf_feature_input, self.resnet(hf_feature_input), "relu"), start=1)
        return lf_feature, hf_feature, self.shallow_up(ops.cat2_act(lf_feature, hf_feature, "relu"), start=1)


class NetworkB2A(nn.Module):
    """model.py:271-298.  forward(hf, lf) -> (hf_feature, lf_feature, out); ``skip`` is dead."""

    def __init__(self, use_bias=False):
        super().__init__()
        self.shallow_frequency = _shallow_frequency(use_bias)
        self.shallow_up = shallowNet(up=True)
        self.skip = _skip(use_bias)
        self.resnet = ResnetGenerator(input_nc=128, output_nc=64, n_blocks=8)
        self.B2A_input = FusedSequential(Conv2d(1, 128, 4, 2, 1, bias=use_bias))

    def forward(self, hf, lf):
        hf_feature = self.shallow_frequency(hf)
        lf_feature = self.resnet(self.B2A_input(lf))
        return hf_feature, lf_feature, self.shallow_up(ops.cat2_act(hf_feature, lf_feature, "relu"), start=1)


class UnetSkipConnectionBlock(nn.Module):
    """model.py:336-400.  Parameter skeleton + forward; not on the hot path."""

    def __init__(self, outer_nc, inner_nc, input_nc=None, submodule=None, outermost=False, innermost=False, norm_layer=BatchNorm2d,
                 use_dropout=True):
        super().__init__()
        self.outermost = outermost
        if input_nc is None:
            input_nc = outer_nc
        downconv = Conv2d(input_nc, inner_nc, 4, 2, 1, bias=True)
        downrelu, downnorm = LeakyReLU(0.2, True), norm_layer(inner_nc)
        uprelu, upnorm = ReLU(True), norm_layer(outer_nc)
        if outermost:
            model = [downconv, submodule]                                  # the reference drops `up` here (model.py:375)
        elif innermost:
            model = [downrelu, downconv, uprelu, ConvTranspose2d(inner_nc, outer_nc, 4, 2, 1, bias=True), upnorm]
        else:
            model = [downrelu, downconv, downnorm, submodule, uprelu, ConvTranspose2d(inner_nc * 2, outer_nc, 4, 2, 1, bias=True), upnorm]
            if use_dropout:
                model.append(nn.Dropout(0.5))
        self.model = FusedSequential(*model)

    def forward(self, x):
        if self.outermost:
            return self.model(x)
        return ops.cat2_act(x, self.model(x), None)


class UnetGenerator(nn.Module):
    """model.py:302-332."""

    def __init__(self, input_nc=1, output_nc=1, num_downs=8, ngf=64, norm_layer=BatchNorm2d, use_dropout=False):
        super().__init__()
        blk = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=None, norm_layer=norm_layer, innermost=True)
        for _ in range(num_downs - 5):
            blk = UnetSkipConnectionBlock(ngf * 8, ngf * 8, submodule=blk, norm_layer=norm_layer, use_dropout=use_dropout)
        blk = UnetSkipConnectionBlock(ngf * 4, ngf * 8, submodule=blk, norm_layer=norm_layer)
        blk = UnetSkipConnectionBlock(ngf * 2, ngf * 4, submodule=blk, norm_layer=norm_layer)
        blk = UnetSkipConnectionBlock(ngf, ngf * 2, submodule=blk, norm_layer=norm_layer)
        self.model = UnetSkipConnectionBlock(output_nc, ngf, input_nc=input_nc, submodule=blk, outermost=True, norm_layer=norm_layer)

    def forward(self, B):
        return self.model(B)


class _ResBlock(nn.Module):
    """x + conv_block(x) with conv_block = conv3x3, BN, ReLU, conv3x3, BN (model.py:403-421 and 483-506).
    The residual add (and a trailing ReLU when the parent Sequential has one) is fused into the second BN."""
    is_residual_block = True

    def __init__(self, dim, norm_layer, use_bias):
        super().__init__()
        self.conv_block = FusedSequential(Conv2d(dim, dim, 3, 1, 1, bias=use_bias), norm_layer(dim), ReLU(True),
                                          Conv2d(dim, dim, 3, 1, 1, bias=use_bias), norm_layer(dim))

    def forward(self, x, post_act=None):
        cb = self.conv_block
        if not cb[1].training and not torch.is_grad_enabled():       # inference: both BatchNorms folded into their convolutions
            y = cb[0](x, act="relu", params=folded_conv_params(cb[0], cb[1]))
            y = ops.add(cb[3](y, params=folded_conv_params(cb[3], cb[4])), x)
            return ops.activation(y, post_act) if post_act else y
        # x receives two gradients: the skip's (from the last BatchNorm's backward) and the first convolution's input gradient.  The
        # shared dict lets that convolution add the former in its own epilogue instead of autograd launching an elementwise add.
        link = {} if (ops.fuse_residual_grad and x.requires_grad and torch.is_grad_enabled() and cb[4].training) else None
        y = cb[1](cb[0](x, link=link), act="relu")
        return cb[4](cb[3](y), act=post_act, residual=x, link=link)


class ResnetBlock(_ResBlock):
    """model.py:403-421 (64 channels, full resolution; ``use_bias`` is forced False at model.py:408)."""

    def __init__(self, dim=64, norm_layer=BatchNorm2d, use_bias=False):
        super().__init__(dim, norm_layer, False)

    def build_conv_block(self, dim=64, norm_layer=BatchNorm2d, use_bias=False):
        return _ResBlock(dim, norm_layer, use_bias).conv_block


class ResidualBlock(_ResBlock):
    """model.py:483-506 (256 channels at 1/4 resolution)."""

    def __init__(self, dim, padding_type="reflect", norm_layer=BatchNorm2d, use_dropout=False, use_bias=False):
        if use_dropout:
            raise NotImplementedError("the reference never enables dropout here (model.py:444)")
        super().__init__(dim, norm_layer, use_bias)


class shallowNet(nn.Module):
    """model.py:423-442: ReLU, (ConvTranspose 4x4 s2 | Conv 3x3), BN, 3 ResnetBlocks, ReLU, Conv 3x3 -> out_dim, Tanh."""

    def __init__(self, in_dim=128, out_dim=1, up=False):
        super().__init__()
        first = ConvTranspose2d(in_dim, 64, 4, 2, 1, bias=False) if up else Conv2d(in_dim, 64, 3, 1, 1, bias=False)
        self.model = FusedSequential(ReLU(True), first, BatchNorm2d(64), ResnetBlock(), ResnetBlock(), ResnetBlock(), ReLU(True),
                                     Conv2d(64, out_dim, 3, 1, 1, bias=False), Tanh())

    def forward(self, x, start=0):
        return self.model(x, start=start)


class ResnetGenerator(nn.Module):
    """model.py:444-480: reflect-pad 7x7, two stride-2 3x3, n_blocks ResidualBlocks, two ConvTranspose 3x3 s2
    (output_padding 1), reflect-pad 7x7 with bias; BatchNorm2d + ReLU; no final tanh (model.py:474)."""

    def __init__(self, input_nc=64, output_nc=64, ngf=64, norm_layer=BatchNorm2d, use_dropout=False, n_blocks=8, padding_type="reflect"):
        assert n_blocks >= 0
        super().__init__()
        use_bias = False     # `norm_layer == nn.InstanceNorm2d` is False in the reference (model.py:448)
        model = [ReflectionPad2d(3), Conv2d(input_nc, ngf, 7, 1, 0, bias=use_bias), norm_layer(ngf), ReLU(True)]
        for i in range(2):
            mult = 2 ** i
            model += [Conv2d(ngf * mult, ngf * mult * 2, 3, 2, 1, bias=use_bias), norm_layer(ngf * mult * 2), ReLU(True)]
        for _ in range(n_blocks):
            model += [ResidualBlock(ngf * 4, padding_type=padding_type, norm_layer=norm_layer, use_dropout=use_dropout, use_bias=use_bias)]
        for i in range(2):
            mult = 2 ** (2 - i)
            model += [ConvTranspose2d(ngf * mult, ngf * mult // 2, 3, 2, 1, output_padding=1, bias=use_bias), norm_layer(ngf * mult // 2),
                      ReLU(True)]
        model += [ReflectionPad2d(3), Conv2d(ngf, output_nc, 7, 1, 0)]
        self.model = FusedSequential(*model)

    def forward(self, input):
        return self.model(input)

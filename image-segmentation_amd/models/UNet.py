"""UNet / LargeUNet with the reference's API and state_dict (reference: models/UNet.py)."""
import torch
import torch.nn as nn

from hipseg import ops
from models.processing_blocks import ConvBlock, ConvBlockDownsample, ConvBlockUpsampleSkip



_NO_STEM_ALIAS = bool(__import__("os").environ.get("HIPSEG_NO_STEM_ALIAS"))
_NO_ENC_ALIAS = bool(__import__("os").environ.get("HIPSEG_NO_ENC_ALIAS"))  # A/B switch: autograd sums the skip gradients


def _stem(conv, x, two=False):
    """1x1 stem on the NCHW image -> NHWC activations in the active precision.  `two`: (activations, alias) for the two
    consumers of the stem output, see ops.StemFn."""
    ops._require_gpu(x)
    if x.shape[1] != conv.in_channels:
        raise ValueError(f"expected {conv.in_channels} input channels, got {x.shape[1]}")
    if _NO_STEM_ALIAS and two:  # A/B switch: one tensor for both consumers (autograd sums the gradients)
        y = ops.StemFn.apply(x.float().contiguous(), conv.weight, conv.bias, ops.precision(), False)
        return y, y
    return ops.StemFn.apply(x.float().contiguous(), conv.weight, conv.bias, ops.precision(), two)


def _head(conv, x):
    return ops.HeadFn.apply(x, conv.weight, conv.bias)


def _hooked(*mods):
    """True if a forward / backward hook is registered on one of the modules (or globally): the fused nodes below call
    forward_from_up / forward_up instead of Module.__call__, which is where hooks fire -- such a model takes the plain
    module-by-module path"""
    import torch.nn.modules.module as M

    if M._global_forward_hooks or M._global_forward_pre_hooks or M._global_backward_hooks or M._global_backward_pre_hooks:
        return True
    return any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks for m in mods)


class _UNetBase(nn.Module):
    _enc = ()
    _bott = ()
    _dec = ()

    def __init__(self, in_channels=3, out_channels=3, activation=nn.Identity()):
        super().__init__()
        self.input = nn.Conv2d(in_channels, 32, kernel_size=1, padding=0)
        for k, (ci, co) in enumerate(self._enc, 1):
            setattr(self, f"enc{k}", ConvBlockDownsample(ci, co))
        self.bottleneck = ConvBlock(*self._bott)
        for k, (ci, co) in enumerate(self._dec, 1):
            setattr(self, f"dec{k}", ConvBlockUpsampleSkip(ci, co))
        self.out = nn.Conv2d(32, out_channels, kernel_size=1, padding=0)
        self.activation = activation

    def _trunk(self, x, fuse=None):
        div = 2 ** len(self._enc)
        if x.dim() != 4 or x.shape[2] % div or x.shape[3] % div:
            raise ValueError(f"input must be (B,C,H,W) with H, W divisible by {div}; got {tuple(x.shape)}")
        ops._require_gpu(x)
        ops.prepack(self, ops.precision())  # all conv / ConvT operands of this step, one launch
        h, h_skip = _stem(self.input, x, two=True)  # one alias per consumer: enc1 below, the last decoder block's skip
        skips = [h_skip]
        for k in range(1, len(self._enc) + 1):
            # one alias per consumer (next block / the decoder's skip input): no gradient-summing pass in backward
            enc = getattr(self, f"enc{k}")
            if _NO_ENC_ALIAS or not torch.is_grad_enabled():
                h = h_skip = enc(h)
            else:
                h, h_skip = enc.forward_two(h)
            skips.append(h_skip)
        # each ConvBlock runs as ONE autograd node with the consumer of its output -- the next decoder block's
        # ConvTranspose2d, or the 1x1 head after the last block (ops.ConvBlockFn): u = the up-sampled tensor handed on
        decs = [getattr(self, f"dec{k}") for k in range(1, len(self._dec) + 1)]
        if _hooked(self.bottleneck, *decs):
            h = self.bottleneck(h) if fuse is None else fuse(h, skips)
            for k, dec in enumerate(decs):
                h = dec(h, skips[-(k + 1)])
            return _head(self.out, h)
        if fuse is None:
            u = self.bottleneck.forward_up(h, decs[0].up)
        else:
            u = decs[0].up_only(fuse(h, skips))
        for k, dec in enumerate(decs[:-1]):
            u = dec.forward_from_up(u, skips[-(k + 1)], next_up=decs[k + 1].up)
        return decs[-1].forward_from_up(u, skips[-len(decs)], head=self.out)

    @torch.compiler.disable
    def forward(self, x):
        return self.activation(self._trunk(x))


class UNet(_UNetBase):
    """3-level U-Net 32-64-128-256, bottleneck 512 (reference: models/UNet.py:7-76).  Note the
    reference's geometry: bottleneck and enc3 are both at H/8, so dec1 up-samples to H/4 and
    bilinearly resizes back to H/8."""
    _enc = ((32, 64), (64, 128), (128, 256))
    _bott = (256, 512)
    _dec = ((512, 256), (256, 128), (128, 64), (64, 32))


class LargeUNet(_UNetBase):
    """4-level variant, bottleneck 1024 (reference: models/UNet.py:78-148)."""
    _enc = ((32, 64), (64, 128), (128, 256), (256, 512))
    _bott = (512, 1024)
    _dec = ((1024, 512), (512, 256), (256, 128), (128, 64), (64, 32))

"""HIP-backed building blocks with the reference's API (reference: models/processing_blocks.py).

The torch.nn layers created in the constructors are PARAMETER CONTAINERS only: they give the
modules the reference's state_dict keys, default initialisation and the attributes that
models/helperFunctions.py:45-78 introspects.  forward() never calls them; it runs the fused HIP
pipeline in hipseg.ops on NHWC activations.
"""
import math
import random

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import cat  # noqa: F401  (the reference module exports `cat`; star-importers may use it)

from hipseg import ops



def _double_conv(seq, x, skip, pool, two=False, head=None, up=None):
    """seq = (conv, bn, relu, conv, bn, relu) parameter container of one ConvBlock.
    two: (output, alias of the output) for a block whose output has two consumers (see ops.ConvBlockFn).
    head: the 1x1 nn.Conv2d that consumes the block's output; its NCHW fp32 logits are returned instead.
    up: the nn.ConvTranspose2d(k2, s2) that consumes the block's output; the up-sampled tensor is returned instead."""
    c1, n1, _, c2, n2, _ = seq
    td = ops._tdtype(ops.precision())
    x = ops.as_nhwc(x, td)
    if skip is not None:
        skip = ops.as_nhwc(skip, td)
        if skip.shape[0] != x.shape[0] or skip.shape[2:] != x.shape[2:]:
            raise ValueError(f"skip shape {tuple(skip.shape)} does not match {tuple(x.shape)}")
    cin = x.shape[1] + (skip.shape[1] if skip is not None else 0)
    if cin != c1.in_channels:
        raise ValueError(f"expected {c1.in_channels} input channels, got {cin}")
    train = n1.training
    stats = (n1.running_mean, n1.running_var, n1.num_batches_tracked, n2.running_mean, n2.running_var,
             n2.num_batches_tracked)
    if up is not None:
        if up.kernel_size != (2, 2) or up.stride != (2, 2) or up.in_channels != c2.out_channels or head is not None:
            raise ValueError(f"up must be a ConvTranspose2d(k2, s2) over {c2.out_channels} channels")
        return ops.ConvBlockFn.apply(x, skip, c1.weight, c1.bias, n1.weight, n1.bias, c2.weight, c2.bias, n2.weight,
                                     n2.bias, *stats, train, pool, not torch.is_grad_enabled(), two, None, None, up.weight,
                                     up.bias)
    if head is not None:
        if head.kernel_size != (1, 1) or head.in_channels != c2.out_channels:
            raise ValueError(f"head must be a 1x1 convolution over {c2.out_channels} channels")
        return ops.ConvBlockFn.apply(x, skip, c1.weight, c1.bias, n1.weight, n1.bias, c2.weight, c2.bias, n2.weight,
                                     n2.bias, *stats, train, pool, not torch.is_grad_enabled(), two, head.weight, head.bias)
    return ops.ConvBlockFn.apply(x, skip, c1.weight, c1.bias, n1.weight, n1.bias, c2.weight, c2.bias, n2.weight,
                                 n2.bias, *stats, train, pool, not torch.is_grad_enabled(), two)


def _conv_seq(cin, cout, k, pad):
    if k != 3 or pad != 1:
        raise NotImplementedError("the HIP ConvBlock implements the reference's 3x3 / padding 1 configuration")
    return nn.Sequential(nn.Conv2d(cin, cout, k, padding=pad), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                         nn.Conv2d(cout, cout, k, padding=pad), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class ConvBlock(nn.Module):
    """conv3x3-BN-ReLU twice (reference: processing_blocks.py:21-52)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1):
        super().__init__()
        self.conv = _conv_seq(in_channels, out_channels, kernel_size, padding)

    @torch.compiler.disable
    def forward(self, x):
        return _double_conv(self.conv, x, None, False)

    @torch.compiler.disable
    def forward_up(self, x, up):
        """up(self(x)) for the ConvTranspose2d(k2, s2) `up` that consumes this block's output (reference:
        models/UNet.py:66-67, bottleneck -> dec1.up) as ONE autograd node: the ConvTranspose2d's data gradient is this
        block's output gradient and, where the shape has such a kernel, also reduces the BatchNorm-backward sums of the
        block's last layer (ops.ConvBlockFn)."""
        if ops._NO_UP_FUSE:
            return _upsample(up, self.forward(x))
        return _double_conv(self.conv, x, None, False, up=up)


class ConvBlockDownsample(nn.Module):
    """ConvBlock then MaxPool2d(2,2); only the pooled tensor is returned
    (reference: processing_blocks.py:54-77).  The pool is fused into the last BN-apply kernel."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.block = nn.Sequential(ConvBlock(in_channels, out_channels), nn.MaxPool2d(kernel_size=2, stride=2))

    @torch.compiler.disable
    def forward(self, x):
        return _double_conv(self.block[0].conv, x, None, True)

    @torch.compiler.disable
    def forward_two(self, x):
        """(pooled output, alias of it): for callers that feed the output to TWO consumers (the U-Nets: next encoder
        block + a decoder's skip input) -- each consumer's gradient then reaches the block's backward on its own
        instead of through an elementwise sum by autograd (see ops.ConvBlockFn)."""
        return _double_conv(self.block[0].conv, x, None, True, two=True)


def _upsample(up, x):
    td = ops._tdtype(ops.precision())
    return ops.ConvT2x2Fn.apply(ops.as_nhwc(x, td), up.weight, up.bias)


class ConvBlockUpsampleSkip(nn.Module):
    """ConvTranspose2d(k2,s2) -> bilinear resize to the skip's size (align_corners=True) ->
    cat([x, skip]) -> ConvBlock (reference: processing_blocks.py:79-109).  The concat is never
    materialised (dual-source conv); the resize is skipped when it is the identity."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2)
        self.conv = ConvBlock(out_channels * 2, out_channels)

    @torch.compiler.disable
    def forward(self, x, skip):
        return self.forward_from_up(_upsample(self.up, x), skip)

    @torch.compiler.disable
    def up_only(self, x):
        """self.up(x): the first half of forward(), for callers that hand the result to forward_from_up()"""
        return _upsample(self.up, x)

    @torch.compiler.disable
    def forward_from_up(self, u, skip, next_up=None, head=None):
        """the block behind `self.up`: u = self.up(x) (already computed, e.g. by the previous block's node) -> bilinear
        resize to the skip's size -> cat -> ConvBlock.  `next_up` / `head`: the ConvTranspose2d of the NEXT decoder block /
        the 1x1 output convolution that consumes this block's output (reference: models/UNet.py:68-73); the block and that
        consumer then run as one autograd node (see ops.ConvBlockFn) and the consumer's result is returned."""
        if u.shape[2:] != skip.shape[2:]:
            u = ops.BilinearFn.apply(u, skip.shape[2], skip.shape[3])
        if head is not None and ops._NO_HEAD_FUSE:  # A/B switch: two autograd nodes, every launch on its own
            return ops.HeadFn.apply(_double_conv(self.conv.conv, u, skip, False), head.weight, head.bias)
        if next_up is not None and ops._NO_UP_FUSE:
            return _upsample(next_up, _double_conv(self.conv.conv, u, skip, False))
        return _double_conv(self.conv.conv, u, skip, False, head=head, up=next_up)

    @torch.compiler.disable
    def forward_head(self, x, skip, head):
        """head(self(x, skip)) for the 1x1 output convolution `head` that follows the LAST decoder block (reference:
        models/UNet.py:72-73), as NCHW fp32 logits.  One autograd node: in training the block's final BatchNorm + ReLU
        runs in the head kernel's load path and the head's backward reduces that layer's BatchNorm-backward sums
        (see ops.ConvBlockFn)."""
        return self.forward_from_up(_upsample(self.up, x), skip, head=head)


class ConvBlockUpsample(nn.Module):
    """ConvTranspose2d(k2,s2) -> ConvBlock, no skip (reference: processing_blocks.py:111-133)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2)
        self.conv = ConvBlock(out_channels, out_channels)

    @torch.compiler.disable
    def forward(self, x):
        return _double_conv(self.conv.conv, _upsample(self.up, x), None, False)


class CrossAttentionFusion(nn.Module):
    """Reference: processing_blocks.py:287-322.  Keys and values are ONE CLIP vector repeated H*W
    times, so the softmax is uniform and the result is out_proj(v_proj(clip)) broadcast over all
    pixels, independent of the query features.  That affine map is evaluated directly (two tiny
    GEMMs on (B, E)); q/k projections receive exactly zero gradient, as in the reference up to
    rounding.  nn.MultiheadAttention is kept as the parameter container (same state_dict keys)."""

    def __init__(self, resnet_channels, num_heads=4):
        super().__init__()
        self.cross_attn = nn.MultiheadAttention(embed_dim=resnet_channels, num_heads=num_heads)

    def forward(self, resnet_feats, clip_feats):
        B, C, H, W = resnet_feats.shape
        a = self.cross_attn
        wv, bv = a.in_proj_weight[2 * C:], a.in_proj_bias[2 * C:]
        v = torch.nn.functional.linear(clip_feats.float(), wv, bv)
        o = torch.nn.functional.linear(v, a.out_proj.weight, a.out_proj.bias)
        td = ops._tdtype(ops.precision())
        return o.to(td)[:, None, None, :].expand(B, H, W, C).contiguous().permute(0, 3, 1, 2)


class CustomClipPreprocessor(nn.Module):
    """Resize to 224x224 + CLIP normalisation, batched (reference: processing_blocks.py:136-170
    loops per image through torchvision transforms; off the kernel path)."""

    def __init__(self, mean, std, target_size=(224, 224)):
        super().__init__()
        self.target_size = tuple(target_size)
        self.register_buffer("mean", torch.tensor(mean).view(1, -1, 1, 1), persistent=False)
        self.register_buffer("std", torch.tensor(std).view(1, -1, 1, 1), persistent=False)

    def forward(self, images):
        if tuple(images.shape[2:]) != self.target_size:
            images = torch.nn.functional.interpolate(images, size=self.target_size, mode="bilinear",
                                                     align_corners=False, antialias=True)
        return (images - self.mean) / self.std


class ClipFeatureExtractor(nn.Module):
    """Frozen CLIP ViT-B/32 image tower on PyTorch-ROCm (reference: processing_blocks.py:173-233).
    Out of scope for hand-written kernels (frozen, no_grad, pretrained weights are a network
    fetch).  `clip_model` may be injected; otherwise the pretrained model is loaded by name as in
    the reference, or -- with HIPSEG_CLIP_RANDOM_INIT=1 -- a random-init CLIPModel(CLIPConfig())
    (the default config IS ViT-B/32) for offline benchmarking."""

    def __init__(self, train=False, clip_model=None):
        super().__init__()
        import os

        mean = [0.48145466, 0.4578275, 0.40821073]
        std = [0.26862954, 0.26130258, 0.27577711]
        self.custom_preprocessor = CustomClipPreprocessor(mean=mean, std=std)
        if clip_model is None:
            from transformers import CLIPConfig, CLIPModel

            if os.environ.get("HIPSEG_CLIP_RANDOM_INIT", "0") == "1":
                clip_model = CLIPModel(CLIPConfig())
            else:
                clip_model = CLIPModel.from_pretrained("openai/clip-vit-base-patch32")
        self.clip_model = clip_model
        self.train_clip = train
        self.set_train(train)

    def set_train(self, value: bool):
        assert isinstance(value, bool), "Value must be a boolean"
        for p in self.clip_model.parameters():
            p.requires_grad = value
        self.train_clip = value

    def _lowp_shadow(self, dtype):
        """Frozen tower under autocast: torch re-casts every fp32 Linear / Conv weight and bias to the autocast dtype on
        EVERY forward (its cast cache only serves parameters that require grad, and is dropped when the autocast region
        ends): 146 cast kernels moving 528 MB per step for ViT-B/32 -- 0.98 ms of the 7.9-ms ClipUnet step.  The casts
        of frozen parameters are constants, so they are made ONCE (same values: same arithmetic as the reference's
        autocast forward) and swapped in for the duration of the call; LayerNorm / embedding parameters, which autocast
        keeps in fp32, stay as they are.  The fp32 parameters remain the module's state (state_dict, optimizer, .to());
        the shadow is rebuilt whenever one of them changes (load_state_dict, a device move)."""
        roots = [getattr(self.clip_model, a) for a in ("vision_model", "visual_projection") if hasattr(self.clip_model, a)]
        roots = roots or [self.clip_model]  # (an injected model without those attributes: every Linear / Conv of it)
        mods = [m for r in roots for m in r.modules() if isinstance(m, (nn.Linear, nn.Conv2d))]
        src = [(m, n, getattr(m, n)) for m in mods for n in ("weight", "bias") if getattr(m, n, None) is not None]
        key = (dtype, tuple((p._version, p.data_ptr()) for _, _, p in src))
        cached = getattr(self, "_shadow", None)
        if cached is None or cached[0] != key:
            with torch.no_grad():
                cached = (key, [(m, n, p, nn.Parameter(p.detach().to(dtype), requires_grad=False)) for m, n, p in src])
            object.__setattr__(self, "_shadow", cached)  # (not a submodule / buffer: never part of the state_dict)
        return cached[1]

    def _branch_norms(self):
        """the LayerNorms at the head of an encoder layer's attention / MLP branch (`layer_norm1`, `layer_norm2` of the
        HF CLIPEncoderLayer): their output feeds Linear layers only, never the residual stream"""
        vm = getattr(self.clip_model, "vision_model", None)
        layers = getattr(getattr(vm, "encoder", None), "layers", None) or ()
        return [m for l in layers for m in (getattr(l, "layer_norm1", None), getattr(l, "layer_norm2", None))
                if isinstance(m, nn.LayerNorm)]

    def forward(self, X):
        inputs = self.custom_preprocessor(X)
        env = __import__("os").environ
        shadow, norms = (), ()
        if not self.train_clip and inputs.is_cuda and torch.is_autocast_enabled() and not env.get("HIPSEG_NO_CLIP_SHADOW"):
            lowp = torch.get_autocast_dtype("cuda")
            shadow = self._lowp_shadow(lowp)
            if not env.get("HIPSEG_NO_CLIP_NORM_CAST"):
                # autocast keeps LayerNorm in fp32 and then casts its output once PER CONSUMER: q_proj, k_proj and v_proj
                # each re-cast the same tensor (24 redundant cast launches per ViT-B/32 forward).  Casting once at the
                # LayerNorm gives every consumer the identical values.
                norms = self._branch_norms()
        try:
            for m, n, _, lo in shadow:
                m._parameters[n] = lo
            for m in norms:
                m.forward = (lambda x, _m=m: nn.LayerNorm.forward(_m, x).to(lowp))
            with torch.set_grad_enabled(self.train_clip):
                feats = self.clip_model.get_image_features(pixel_values=inputs)
        finally:
            for m, n, p, _ in shadow:
                m._parameters[n] = p
            for m in norms:
                m.__dict__.pop("forward", None)
        return feats if torch.is_tensor(feats) else feats.pooler_output


class ResNet34FeatureExtractor(nn.Module):
    """ImageNet ResNet-34 trunk without avgpool/fc, optionally frozen (reference: processing_blocks.py:236-285).
    Off the hot path (used by ClipResSegmentationModel only) and a torchvision model: kept on PyTorch-ROCm.  Raises a
    clear ImportError when torchvision is not installed instead of failing at module import."""

    def __init__(self, train=False):
        super().__init__()
        try:
            import torchvision.models as tvm
        except ImportError as e:  # pragma: no cover - depends on the environment
            raise ImportError("ResNet34FeatureExtractor needs torchvision (reference: processing_blocks.py:262 "
                              "`models.resnet34(weights='IMAGENET1K_V1')`)") from e
        resnet = tvm.resnet34(weights="IMAGENET1K_V1")
        self.model = nn.Sequential(*list(resnet.children())[:-2])
        self.set_train(train)
        self.train_res = train

    def set_train(self, value: bool):
        assert isinstance(value, bool), "Value must be a boolean"
        for p in self.model.parameters():
            p.requires_grad = value
        self.train_res = value

    def forward(self, X):
        with torch.set_grad_enabled(self.train_res):
            return self.model(X)


# ------------------------------------------------------------------------------------------------
# On-device augmentation (reference: processing_blocks.py:324-451).  The reference composes kornia modules; here the
# host only SAMPLES the random parameters (torch RNG on the device, no host sync) and one fused HIP pipeline
# (csrc/augment.hip) applies flip + rotation to image||mask[||prompt] and colour jitter + blur to the image.
# kornia 0.8.0 is absent: parameter ranges / probabilities follow its documented defaults, PARITY UNPINNED.
class _AugmentorBase(nn.Module):
    flip_p = 0.5            # K.RandomHorizontalFlip() default p
    rotate_p = 0.5          # K.RandomRotation default p
    degrees = 90.0          # K.RandomRotation(90): angle ~ U[-90, 90]
    brightness = 0.4        # K.ColorJitter(brightness=0.4): factor ~ U[0.6, 1.4]
    contrast = 0.3          # factor ~ U[0.7, 1.3]
    saturation = 0.2        # factor ~ U[0.8, 1.2]
    hue = 0.2               # shift ~ U[-0.2, 0.2] turns
    sigma = (0.1, 2.0)      # K.RandomGaussianBlur(kernel_size=(5,5), sigma=(0.1, 2.0), p=1.0)

    def __init__(self, augmentations_per_datapoint):
        super().__init__()
        self.augmentations_per_datapoint = augmentations_per_datapoint
        self.last_params = None  # (params, order) of the most recent call: lets tests replay it on the CPU oracle

    def sample_params(self, B, device):
        """(B, AUG_NPARAM) fp32 table + int32[4] op order, sampled on `device` (see include/hipseg.h): two torch RNG
        launches (8 uniforms per sample, the op permutation) and one tiny kernel that derives the table."""
        from hipseg import _lib as L

        ops._require_gpu(torch.empty(0, device=device))
        u = torch.rand(B, 8, device=device)
        p = torch.empty(B, L.AUG_NPARAM, device=device)
        L.augment_params(ops.ptr(u), ops.ptr(p), B, self.augmentations_per_datapoint + 1, self.flip_p, self.rotate_p,
                         self.degrees, self.brightness, self.contrast, self.saturation, self.hue, self.sigma[0],
                         self.sigma[1], ops._stream())
        order = torch.randperm(4, device=device).to(torch.int32)
        return p, order


class DataAugmentor(_AugmentorBase):
    """forward(images (B,3,H,W), masks (B,H,W)) -> (images, masks.long()); every (aug+1)-th sample untouched
    (reference: processing_blocks.py:324-384)."""

    @torch.no_grad()
    def forward(self, images, masks):
        params, order = self.sample_params(images.shape[0], images.device)
        self.last_params = (params, order)
        out, om, _ = ops.augment(images, masks, None, params, order)
        return out, om


class DataAugmentorPrompt(_AugmentorBase):
    """forward(images, masks (B,H,W) or (B,1,H,W), prompts (B,1,H,W)) -> (images, masks (B,H,W) long, prompts)
    (reference: processing_blocks.py:386-451)."""

    @torch.no_grad()
    def forward(self, images, masks, prompts):
        if masks.dim() == 4:
            masks = masks[:, 0]
        if prompts.dim() == 3:
            prompts = prompts.unsqueeze(1)
        params, order = self.sample_params(images.shape[0], images.device)
        self.last_params = (params, order)
        out, om, oe = ops.augment(images, masks, prompts, params, order)
        return out, om, oe


# ------------------------------------------------------------------------------------------------
# Robustness perturbations of TestWrapper.test_robustness (reference: processing_blocks.py:454-592,
# model_wrappers.py:740-764).  Test-time only, a few elementwise passes per evaluation batch: plain torch ops.
class GaussianPixelNoise(nn.Module):
    """img + N(0, (std/255)^2), clamped to [0,1] (reference :454-474)."""

    def __init__(self, std):
        super().__init__()
        self.std = std

    def forward(self, img):
        return torch.clamp(img + torch.randn_like(img) * (self.std / 255.0), 0.0, 1.0)


class RepeatedBlur(nn.Module):
    """`times` x 3x3 box blur (reference :477-496 uses kornia.filters.box_blur: normalised, reflect border)."""

    def __init__(self, times):
        super().__init__()
        self.times = times

    def forward(self, img):
        for _ in range(self.times):
            img = F.avg_pool2d(F.pad(img, (1, 1, 1, 1), mode="reflect"), kernel_size=3, stride=1)
        return img


class ContrastChange(nn.Module):
    """img * factor, clamped (reference :499-518)."""

    def __init__(self, factor):
        super().__init__()
        self.factor = factor

    def forward(self, img):
        return torch.clamp(img * self.factor, 0.0, 1.0)


class BrightnessChange(nn.Module):
    """img + offset/255, clamped (reference :521-539)."""

    def __init__(self, offset):
        super().__init__()
        self.offset = offset / 255.0

    def forward(self, img):
        return torch.clamp(img + self.offset, 0.0, 1.0)


class Occlusion(nn.Module):
    """zero one random size x size square per image, IN PLACE as the reference does (reference :542-563).
    The draw ORDER is part of the contract and is kept on purpose: per image first the column, then the row, both from
    Python's `random` module with the reference's bounds, so a caller that seeds `random` (the robustness sweeps of
    TestWrapper) gets the reference's squares; everything else here is this module's own."""

    def __init__(self, size):
        super().__init__()
        self.size = size

    def forward(self, img):
        b, c, h, w = img.shape
        for i in range(b):
            x = random.randint(0, w - self.size) if w > self.size else 0
            y = random.randint(0, h - self.size) if h > self.size else 0
            img[i, :, y:y + self.size, x:x + self.size] = 0.0
        return img


class SaltAndPepper(nn.Module):
    """one uniform draw per pixel: < amount/2 -> salt (1), > 1 - amount/2 -> pepper (0) (reference :565-592)."""

    def __init__(self, amount):
        super().__init__()
        self.amount = amount

    def forward(self, img):
        half = self.amount / 2
        # one draw per pixel, shared by the channels: the lowest `half` of the unit interval turns the pixel white, the
        # highest `half` black, everything between keeps the image
        u = torch.rand(img.shape[0], 1, img.shape[2], img.shape[3], device=img.device)
        out = torch.where(u < half, torch.ones_like(img), img)
        return torch.where(u > 1 - half, torch.zeros_like(img), out)

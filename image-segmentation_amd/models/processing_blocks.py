"""HIP-backed building blocks with the reference's API (reference: models/processing_blocks.py).

The torch.nn layers created in the constructors are PARAMETER CONTAINERS only: they give the
modules the reference's state_dict keys, default initialisation and the attributes that
models/helperFunctions.py:45-78 introspects.  forward() never calls them; it runs the fused HIP
pipeline in hipseg.ops on NHWC activations.
"""
import torch
import torch.nn as nn

from hipseg import ops

__all__ = ["ConvBlock", "ConvBlockDownsample", "ConvBlockUpsampleSkip", "ConvBlockUpsample",
           "CrossAttentionFusion", "CustomClipPreprocessor", "ClipFeatureExtractor"]


def _double_conv(seq, x, skip, pool):
    """seq = (conv, bn, relu, conv, bn, relu) parameter container of one ConvBlock."""
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
    return ops.ConvBlockFn.apply(x, skip, c1.weight, c1.bias, n1.weight, n1.bias, c2.weight, c2.bias, n2.weight,
                                 n2.bias, *stats, train, pool)


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


class ConvBlockDownsample(nn.Module):
    """ConvBlock then MaxPool2d(2,2); only the pooled tensor is returned
    (reference: processing_blocks.py:54-77).  The pool is fused into the last BN-apply kernel."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.block = nn.Sequential(ConvBlock(in_channels, out_channels), nn.MaxPool2d(kernel_size=2, stride=2))

    @torch.compiler.disable
    def forward(self, x):
        return _double_conv(self.block[0].conv, x, None, True)


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
        x = _upsample(self.up, x)
        if x.shape[2:] != skip.shape[2:]:
            x = ops.BilinearFn.apply(x, skip.shape[2], skip.shape[3])
        return _double_conv(self.conv.conv, x, skip, False)


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

    def forward(self, X):
        inputs = self.custom_preprocessor(X)
        with torch.set_grad_enabled(self.train_clip):
            feats = self.clip_model.get_image_features(pixel_values=inputs)
        return feats if torch.is_tensor(feats) else feats.pooler_output

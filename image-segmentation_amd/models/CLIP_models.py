"""ClipUnet with the reference's API and state_dict (reference: models/CLIP_models.py:63-134)."""
import torch
import torch.nn as nn

from hipseg import ops

from models.processing_blocks import *  # noqa: F401,F403  (as the reference module does, CLIP_models.py:4)
from models.processing_blocks import (ClipFeatureExtractor, ConvBlock, ConvBlockDownsample,  # noqa: F401
                                      ConvBlockUpsample, ConvBlockUpsampleSkip, CrossAttentionFusion,
                                      ResNet34FeatureExtractor)
from models.UNet import UNet, _head, _stem


class _DeadBranchZeroGrads(torch.autograd.Function):
    """identity on `y`; in backward every listed parameter receives an all-zero gradient (written straight into its
    HipDDP bucket slot when it has one)."""

    @staticmethod
    def forward(ctx, y, *params):
        ctx.params = params
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        return (g,) + tuple(ops.grad_out(p).zero_() for p in ctx.params)


class ClipUnet(UNet):
    """UNet trunk whose bottleneck output is REPLACED by CrossAttentionFusion(bottleneck, clip)
    (reference: CLIP_models.py:115-134; `activation` is stored but never applied there).
    Because the fusion output does not depend on the bottleneck features (degenerate attention:
    every key is the same CLIP token, so the softmax is uniform whatever the query), the
    bottleneck ConvBlock contributes nothing to the output, and its true gradient is zero.  The
    reference still runs it, which has two visible effects on the state a checkpoint holds:
      * in train mode its BatchNorm running statistics and `num_batches_tracked` move;
      * its parameters receive a gradient tensor (zero up to ~1e-11 rounding residue of the softmax backward, measured
        on the 224 x 224 golden), NOT None -- so Adam's `weight_decay` (1e-4 in the reference's loops,
        model_wrappers.py:43) keeps acting on them and they shrink by ~lr per step.
    With `run_dead_bottleneck` (default True) the block's FORWARD is therefore executed in train mode under no_grad
    (nothing saved, nothing run in backward) and its parameters receive exact-zero gradients, so the optimizer treats
    them as the reference's does; in eval mode it changes no state and is skipped.  Set it to False to drop the
    ~4 % of step time it costs (the parameters then get no gradient at all)."""

    run_dead_bottleneck = True

    def __init__(self, out_channels=3, in_channels=3, activation=nn.Identity(), clip_feature_extractor=None):
        nn.Module.__init__(self)
        self.clip_feature_extractor = (clip_feature_extractor if clip_feature_extractor is not None
                                       else ClipFeatureExtractor(train=False))
        self.cross_attention_fusion = CrossAttentionFusion(512, num_heads=1)
        # same registration order as the reference after the two modules above
        base = UNet(in_channels=in_channels, out_channels=out_channels, activation=activation)
        for name, mod in base.named_children():
            setattr(self, name, mod)

    @torch.compiler.disable
    def forward(self, X):
        clip_features = self.clip_feature_extractor(X)
        return self._trunk(X, fuse=lambda h, skips: _fuse_clip(self, h, clip_features))


def _fuse_clip(model, h, clip_features):
    """CrossAttentionFusion(bottleneck(h), clip) of ClipUnet / ClipUnetPrompt: the bottleneck's output never reaches
    the result (see ClipUnet's docstring); its train-mode bookkeeping and zero gradients are kept."""
    dead = model.run_dead_bottleneck and model.bottleneck.training
    if dead:
        with torch.no_grad():
            model.bottleneck(h)  # BatchNorm bookkeeping only
    B, _, H, W = h.shape
    ref = h.new_empty((B, 512, H, W), device="meta")
    y = model.cross_attention_fusion(ref, clip_features)
    if dead and torch.is_grad_enabled() and y.requires_grad:
        y = _DeadBranchZeroGrads.apply(y, *[p for p in model.bottleneck.parameters() if p.requires_grad])
    return y


class ClipAutoencoder(nn.Module):
    """CLIP vector -> Linear(512, 64*16*16) -> three ConvBlockUpsample -> ConvBlockUpsampleSkip with the 1x1-stem image
    features -> 1x1 head (reference: CLIP_models.py:136-188; `activation` is stored but not applied there).  Same
    blocks, same kernels as the U-Nets; the coupler is a plain torch Linear."""

    def __init__(self, out_channels=3, in_channels=3, activation=nn.Identity(), clip_feature_extractor=None):
        super().__init__()
        self.clip_feature_extractor = (clip_feature_extractor if clip_feature_extractor is not None
                                       else ClipFeatureExtractor(train=False))
        self.input = nn.Conv2d(in_channels, 32, kernel_size=1, padding=0)
        self.coupler = nn.Linear(512, 16384)
        self.dec1 = ConvBlockUpsample(64, 64)
        self.dec2 = ConvBlockUpsample(64, 64)
        self.dec3 = ConvBlockUpsample(64, 32)
        self.dec4 = ConvBlockUpsampleSkip(32, 32)
        self.out = nn.Conv2d(32, out_channels, kernel_size=1, padding=0)
        self.activation = activation

    @torch.compiler.disable
    def forward(self, X):
        clip_features = self.clip_feature_extractor(X)
        ops.prepack(self, ops.precision())  # all conv / ConvT operands of this step, one launch
        inp = _stem(self.input, X)
        bottleneck = self.coupler(clip_features.float()).view(-1, 64, 16, 16)
        d = self.dec3(self.dec2(self.dec1(bottleneck)))
        return self.dec4.forward_head(d, inp, self.out)  # last block + 1x1 head: one autograd node


class ClipResSegmentationModel(nn.Module):
    """ResNet-34 features fused with the CLIP vector, five ConvBlockUpsample, ConvBlock on cat([dec5, X]) (reference:
    CLIP_models.py:8-61).  Needs torchvision's pretrained ResNet-34 (ResNet34FeatureExtractor raises ImportError
    without it); the decoder runs on the HIP blocks."""

    def __init__(self, out_channels=3, in_channels=3, activation=nn.Identity()):
        super().__init__()
        self.clip_feature_extractor = ClipFeatureExtractor(train=False)
        self.encoder = ResNet34FeatureExtractor(train=False)
        self.cross_attention_fusion = CrossAttentionFusion(512, num_heads=4)
        self.dec1 = ConvBlockUpsample(512, 256)
        self.dec2 = ConvBlockUpsample(256, 128)
        self.dec3 = ConvBlockUpsample(128, 64)
        self.dec4 = ConvBlockUpsample(64, 32)
        self.dec5 = ConvBlockUpsample(32, 16)
        self.out = ConvBlock(in_channels=19, out_channels=out_channels)

    @torch.compiler.disable
    def forward(self, X):
        clip_features = self.clip_feature_extractor(X)
        ops.prepack(self, ops.precision())
        attn = self.cross_attention_fusion(self.encoder(X), clip_features)
        d = self.dec5(self.dec4(self.dec3(self.dec2(self.dec1(attn)))))
        return self.out(torch.cat([d.float(), X.float()], dim=1)).float()

"""ClipUnet with the reference's API and state_dict (reference: models/CLIP_models.py:63-134)."""
import torch
import torch.nn as nn

from models.processing_blocks import (ClipFeatureExtractor, ConvBlock, ConvBlockDownsample,  # noqa: F401
                                      ConvBlockUpsampleSkip, CrossAttentionFusion)
from models.UNet import UNet



class ClipUnet(UNet):
    """UNet trunk whose bottleneck output is REPLACED by CrossAttentionFusion(bottleneck, clip)
    (reference: CLIP_models.py:115-134; `activation` is stored but never applied there).
    Because the fusion output does not depend on the bottleneck features (degenerate attention),
    the bottleneck ConvBlock contributes nothing to the output or to any gradient.  The reference
    still runs it, which in train mode updates its BatchNorm running statistics and
    `num_batches_tracked` -- state that ends up in checkpoints.  With `run_dead_bottleneck` (default
    True) the block's FORWARD is therefore executed in train mode, under no_grad (nothing is saved,
    nothing runs in backward, its parameters receive no gradient -- the reference's are exactly zero
    up to rounding), so a saved state_dict matches the reference's; in eval mode it changes no state
    and is skipped.  Set it to False to drop the ~4 % of step time it costs."""

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

        def fuse(h, skips):
            if self.run_dead_bottleneck and self.bottleneck.training:
                with torch.no_grad():
                    self.bottleneck(h)  # BatchNorm bookkeeping only (see the class docstring)
            B, _, H, W = h.shape
            ref = h.new_empty((B, 512, H, W), device="meta")
            return self.cross_attention_fusion(ref, clip_features)

        return self._trunk(X, fuse=fuse)

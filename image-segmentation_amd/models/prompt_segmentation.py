"""ClipUnetPrompt / PromptEncoder with the reference's API and state_dict (reference: models/prompt_segmentation.py:16-95).

ClipUnetPrompt is the model scripts/prompt_train.py:55-58 trains with HybridLossBinary (Dice + BCE): the ClipUnet trunk,
a second encoder over the 1-channel prompt heat map, and a 1x1 convolution fusing cat([attention output, prompt
embedding]) back to 512 channels in front of the decoder."""
import torch
import torch.nn as nn

from hipseg import ops

from models.processing_blocks import *  # noqa: F401,F403  (as the reference module does, prompt_segmentation.py:10)
from models.processing_blocks import (ClipFeatureExtractor, ConvBlock, ConvBlockDownsample, ConvBlockUpsampleSkip,
                                      CrossAttentionFusion)
from models.CLIP_models import _fuse_clip
from models.UNet import _head, _stem


class PromptEncoder(nn.Module):
    """1-channel heat map -> three ConvBlockDownsample (32, 64, 128) -> ConvBlock(128, out_channels) at 1/8 resolution
    (reference: prompt_segmentation.py:16-30)."""

    def __init__(self, out_channels=512):
        super().__init__()
        self.enc1 = ConvBlockDownsample(1, 32)
        self.enc2 = ConvBlockDownsample(32, 64)
        self.enc3 = ConvBlockDownsample(64, 128)
        self.conv = ConvBlock(128, out_channels)

    @torch.compiler.disable
    def forward(self, x):
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] % 8 or x.shape[3] % 8:
            raise ValueError(f"prompt heat map must be (B,1,H,W) with H, W divisible by 8; got {tuple(x.shape)}")
        return self.conv(self.enc3(self.enc2(self.enc1(x))))


class ClipUnetPrompt(nn.Module):
    """reference: prompt_segmentation.py:32-95.  forward(X, prompt_heatmap) -> activation(logits), logits (B,
    out_channels, H, W) fp32.  As in ClipUnet the attention output does not depend on the bottleneck features, so the
    bottleneck block is dead weight whose train-mode state changes are kept (`run_dead_bottleneck`, see
    models/CLIP_models.py); unlike ClipUnet the reference DOES apply `activation` here (prompt_segmentation.py:95)."""

    run_dead_bottleneck = True

    def __init__(self, out_channels=1, in_channels=3, activation=nn.Identity(), clip_feature_extractor=None):
        super().__init__()
        self.clip_feature_extractor = (clip_feature_extractor if clip_feature_extractor is not None
                                       else ClipFeatureExtractor(train=False))
        self.cross_attention_fusion = CrossAttentionFusion(512, num_heads=1)
        self.input = nn.Conv2d(in_channels, 32, kernel_size=1, padding=0)
        self.enc1 = ConvBlockDownsample(32, 64)
        self.enc2 = ConvBlockDownsample(64, 128)
        self.enc3 = ConvBlockDownsample(128, 256)
        self.bottleneck = ConvBlock(256, 512)
        self.prompt_encoder = PromptEncoder(out_channels=512)
        self.prompt_fusion = nn.Conv2d(1024, 512, kernel_size=1, padding=0)
        self.dec1 = ConvBlockUpsampleSkip(512, 256)
        self.dec2 = ConvBlockUpsampleSkip(256, 128)
        self.dec3 = ConvBlockUpsampleSkip(128, 64)
        self.dec4 = ConvBlockUpsampleSkip(64, 32)
        self.out = nn.Conv2d(32, out_channels, kernel_size=1)
        self.activation = activation

    @torch.compiler.disable
    def forward(self, X, prompt_heatmap):
        if X.dim() != 4 or X.shape[2] % 8 or X.shape[3] % 8:
            raise ValueError(f"input must be (B,C,H,W) with H, W divisible by 8; got {tuple(X.shape)}")
        if prompt_heatmap.shape[0] != X.shape[0] or prompt_heatmap.shape[2:] != X.shape[2:]:
            raise ValueError(f"prompt heat map {tuple(prompt_heatmap.shape)} does not match the image {tuple(X.shape)}")
        ops._require_gpu(X)
        clip_features = self.clip_feature_extractor(X)
        ops.prepack(self, ops.precision())  # all 3x3 conv / ConvT operands of this step, one launch
        inp, inp_skip = _stem(self.input, X, two=True)  # (an alias per consumer, see ops.StemFn)
        # (enc1 / enc2 feed the next block AND a decoder's skip input: one alias per consumer, see ops.ConvBlockFn)
        grad = torch.is_grad_enabled()
        e1, enc1 = self.enc1.forward_two(inp) if grad else (self.enc1(inp),) * 2
        e2, enc2 = self.enc2.forward_two(e1) if grad else (self.enc2(e1),) * 2
        enc3 = self.enc3(e2)
        prompt_embedding = self.prompt_encoder(prompt_heatmap.float())
        attention_output = ops.as_nhwc(_fuse_clip(self, enc3, clip_features), prompt_embedding.dtype)
        # prompt_fusion(cat([attention_output, prompt_embedding], dim=1)): dual-source 1x1 conv, no concatenation
        fused = ops.Conv1x1Fn.apply(attention_output, prompt_embedding, self.prompt_fusion.weight, self.prompt_fusion.bias)
        # each decoder ConvBlock as one autograd node with the consumer of its output (next block's ConvTranspose2d / head)
        u = self.dec1.forward_from_up(self.dec1.up_only(fused), enc3, next_up=self.dec2.up)
        u = self.dec2.forward_from_up(u, enc2, next_up=self.dec3.up)
        u = self.dec3.forward_from_up(u, enc1, next_up=self.dec4.up)
        return self.activation(self.dec4.forward_from_up(u, inp_skip, head=self.out))

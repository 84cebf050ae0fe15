"""Plain-PyTorch CPU restatement of the reference's U-Net / ClipUnet hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``) -- parity PINNED by
``tests/golden`` (generated from the reference's own modules).

Functional style: every function takes the activations plus a flat
``state_dict``-shaped mapping (same keys / shapes the reference registers,
SURVEY.md section 8b) and a key prefix.  Citations are to files under
``/root/reference``.
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# (enc blocks, bottleneck, dec blocks) -- models/UNet.py:36-58 and :103-127
ARCHS = {
    "UNet": dict(enc=[(32, 64), (64, 128), (128, 256)], bott=(256, 512),
                 dec=[(512, 256), (256, 128), (128, 64), (64, 32)]),
    "LargeUNet": dict(enc=[(32, 64), (64, 128), (128, 256), (256, 512)], bott=(512, 1024),
                      dec=[(1024, 512), (512, 256), (256, 128), (128, 64), (64, 32)]),
}
ARCHS["ClipUnet"] = ARCHS["UNet"]  # models/CLIP_models.py:89-113 (same trunk + fusion)


# ----------------------------------------------------------------------------- state
def _conv_block_entries(sd, p, cin, cout):
    # models/processing_blocks.py:40-49  nn.Sequential(conv, bn, relu, conv, bn, relu)
    for idx, ci in ((0, cin), (3, cout)):
        sd[f"{p}{idx}.weight"] = torch.zeros(cout, ci, 3, 3)
        sd[f"{p}{idx}.bias"] = torch.zeros(cout)
        b = idx + 1
        sd[f"{p}{b}.weight"] = torch.ones(cout)
        sd[f"{p}{b}.bias"] = torch.zeros(cout)
        sd[f"{p}{b}.running_mean"] = torch.zeros(cout)
        sd[f"{p}{b}.running_var"] = torch.ones(cout)
        sd[f"{p}{b}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)


def make_state(arch="UNet", in_channels=3, out_channels=3):
    """Zero/identity-initialised state with the reference's key order and shapes."""
    a = ARCHS[arch]
    sd = OrderedDict()
    if arch == "ClipUnet":  # CLIP_models.py:93 CrossAttentionFusion(512, num_heads=1)
        q = "cross_attention_fusion.cross_attn."
        sd[q + "in_proj_weight"] = torch.zeros(1536, 512)
        sd[q + "in_proj_bias"] = torch.zeros(1536)
        sd[q + "out_proj.weight"] = torch.zeros(512, 512)
        sd[q + "out_proj.bias"] = torch.zeros(512)
    sd["input.weight"] = torch.zeros(32, in_channels, 1, 1)
    sd["input.bias"] = torch.zeros(32)
    for k, (ci, co) in enumerate(a["enc"], 1):
        _conv_block_entries(sd, f"enc{k}.block.0.conv.", ci, co)
    _conv_block_entries(sd, "bottleneck.conv.", *a["bott"])
    for k, (ci, co) in enumerate(a["dec"], 1):
        sd[f"dec{k}.up.weight"] = torch.zeros(ci, co, 2, 2)
        sd[f"dec{k}.up.bias"] = torch.zeros(co)
        _conv_block_entries(sd, f"dec{k}.conv.conv.", 2 * co, co)
    sd["out.weight"] = torch.zeros(out_channels, 32, 1, 1)
    sd["out.bias"] = torch.zeros(out_channels)
    return sd


def is_param(name):
    return not name.endswith(("running_mean", "running_var", "num_batches_tracked"))


# ----------------------------------------------------------------------------- blocks
def _bn(x, sd, p, train):
    if train:
        sd[p + "num_batches_tracked"] += 1  # nn.BatchNorm2d bookkeeping
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"],
                        sd[p + "bias"], train, BN_MOMENTUM, BN_EPS)


def conv_block(x, sd, p, train):
    """ConvBlock.forward -- models/processing_blocks.py:40-52."""
    x = F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], padding=1)
    x = F.relu(_bn(x, sd, p + "1.", train))
    x = F.conv2d(x, sd[p + "3.weight"], sd[p + "3.bias"], padding=1)
    return F.relu(_bn(x, sd, p + "4.", train))


def down(x, sd, p, train):
    """ConvBlockDownsample.forward -- processing_blocks.py:69-77 (returns the pooled tensor only)."""
    return F.max_pool2d(conv_block(x, sd, p + "block.0.conv.", train), 2, 2)


def up_skip(x, skip, sd, p, train):
    """ConvBlockUpsampleSkip.forward -- processing_blocks.py:100-109."""
    x = F.conv_transpose2d(x, sd[p + "up.weight"], sd[p + "up.bias"], stride=2)
    x = F.interpolate(x, size=skip.shape[2:], mode="bilinear", align_corners=True)
    return conv_block(torch.cat([x, skip], dim=1), sd, p + "conv.conv.", train)


def up(x, sd, p, train):
    """ConvBlockUpsample.forward -- processing_blocks.py:126-133."""
    x = F.conv_transpose2d(x, sd[p + "up.weight"], sd[p + "up.bias"], stride=2)
    return conv_block(x, sd, p + "conv.conv.", train)


def cross_attention_fusion(feats, clip, sd, p="cross_attention_fusion.cross_attn.", collapsed=False):
    """CrossAttentionFusion.forward -- processing_blocks.py:310-322, one head, E=512.

    Keys/values are the same CLIP vector repeated H*W times, so the softmax is
    uniform and the result is out_proj(v_proj(clip)) broadcast over pixels
    (``collapsed=True`` evaluates exactly that affine map).
    """
    B, C, H, W = feats.shape
    wq, wk, wv = sd[p + "in_proj_weight"].chunk(3, 0)
    bq, bk, bv = sd[p + "in_proj_bias"].chunk(3, 0)
    v = clip @ wv.t() + bv  # (B, C)
    if collapsed:
        o = v @ sd[p + "out_proj.weight"].t() + sd[p + "out_proj.bias"]
        return o[:, :, None, None].expand(B, C, H, W)
    q = feats.flatten(2).permute(2, 0, 1) @ wq.t() + bq  # (L, B, C)
    k = (clip @ wk.t() + bk)[None].expand(H * W, B, C)
    vv = v[None].expand(H * W, B, C)
    att = torch.softmax(torch.einsum("lbc,sbc->bls", q, k) / (C ** 0.5), dim=-1)
    o = torch.einsum("bls,sbc->lbc", att, vv) @ sd[p + "out_proj.weight"].t() + sd[p + "out_proj.bias"]
    return o.permute(1, 2, 0).reshape(B, C, H, W)


def unet_forward(x, sd, arch="UNet", train=True, clip_features=None, collapsed=False):
    """UNet.forward (models/UNet.py:60-76), LargeUNet.forward (:129-148),
    ClipUnet.forward trunk (models/CLIP_models.py:115-134, clip_features injected)."""
    a = ARCHS[arch]
    h = F.conv2d(x, sd["input.weight"], sd["input.bias"])
    skips = [h]
    for k in range(1, len(a["enc"]) + 1):
        h = down(h, sd, f"enc{k}.", train)
        skips.append(h)
    h = conv_block(h, sd, "bottleneck.conv.", train)
    if arch == "ClipUnet":
        h = cross_attention_fusion(h, clip_features, sd, collapsed=collapsed)
    for k in range(1, len(a["dec"]) + 1):
        h = up_skip(h, skips[-k], sd, f"dec{k}.", train)
    return F.conv2d(h, sd["out.weight"], sd["out.bias"])


# ----------------------------------------------------------------------------- losses / metrics
def hybrid_loss(pred, target):
    """HybridLoss.forward -- models/losses.py:13-15: cross-entropy only."""
    return F.cross_entropy(pred, target)


def dice_loss_binary_smp(y_pred, y_true, from_logits=True, smooth=0.0, eps=1e-7):
    """segmentation_models_pytorch==0.4.0 DiceLoss(mode='binary') as published
    (third-party, absent here -> PARITY UNPINNED, known-answer tests only).
    from_logits=True re-applies a sigmoid (logsigmoid().exp()); sums over
    dims (0,2) of the (B,1,-1) views; denominator clamp_min(eps); loss zeroed
    when the target is empty; mean over the single class."""
    bs = y_true.size(0)
    if from_logits:
        y_pred = F.logsigmoid(y_pred).exp()
    y_true = y_true.reshape(bs, 1, -1).to(y_pred.dtype)
    y_pred = y_pred.reshape(bs, 1, -1)
    inter = torch.sum(y_pred * y_true, dim=(0, 2))
    card = torch.sum(y_pred + y_true, dim=(0, 2))
    score = (2.0 * inter + smooth) / (card + smooth).clamp_min(eps)
    loss = (1.0 - score) * (y_true.sum(dim=(0, 2)) > 0).to(score.dtype)
    return loss.mean()


def hybrid_loss_binary(pred, target):
    """HybridLossBinary.forward -- models/losses.py:24-36 (BCE-with-logits + smp Dice on sigmoid(pred))."""
    if target.dim() == 3:
        target = target.unsqueeze(1)
    bce = F.binary_cross_entropy_with_logits(pred, target)
    return bce + dice_loss_binary_smp(torch.sigmoid(pred), target)


def iou(preds, targets, eps=1e-6):
    """IoU.forward -- models/losses.py:43-63."""
    cls = torch.argmax(torch.softmax(preds, 1), 1)
    vals = []
    for c in range(preds.shape[1]):
        p, t = (cls == c).float(), (targets == c).float()
        inter = (p * t).sum()
        vals.append((inter + eps) / (p.sum() + t.sum() - inter + eps))
    return torch.stack(vals).mean()


def pixel_accuracy(preds, targets):
    """PixelAccuracy.forward -- models/losses.py:133-154 (3 classes, absent classes skipped)."""
    cls = torch.argmax(torch.softmax(preds, 1), 1)
    accs = []
    for c in range(3):
        m = targets == c
        if m.sum() > 0:
            accs.append(((cls == targets) & m).float().sum() / m.float().sum())
    return torch.stack(accs).mean()


def iou_binary(preds, targets, eps=1e-6, threshold=0.5):
    """IoUBinary.forward -- models/losses.py:71-90 (per-sample IoU, then mean)."""
    p = (torch.sigmoid(preds) > threshold).float().squeeze(1)
    t = targets.float() if targets.dim() == 3 else targets.float().squeeze(1)
    inter = (p * t).sum(dim=[1, 2])
    union = p.sum(dim=[1, 2]) + t.sum(dim=[1, 2]) - inter
    return ((inter + eps) / (union + eps)).mean()


def pixel_accuracy_binary(preds, targets, threshold=0.5):
    """PixelAccuracyBinary.forward -- models/losses.py:161-180."""
    p = (torch.sigmoid(preds) > threshold).float().squeeze(1)
    t = (targets.squeeze(1) if targets.dim() == 4 else targets).float()
    return (p == t).float().sum() / t.numel()


# ----------------------------------------------------------------------------- train step (cpu_baseline leg)
class OracleTrainer:
    """fwd + CE + bwd + Adam(lr 1e-3, weight_decay 1e-4) on the CPU restatement --
    the loop body of models/model_wrappers.py:167-177 without autocast/GradScaler
    (fp32 on CPU)."""

    def __init__(self, arch="UNet", seed=0, lr=1e-3, weight_decay=1e-4):
        from .fill import fill_state_dict

        self.arch = arch
        self.sd = fill_state_dict(make_state(arch), seed)
        self.params = [v.requires_grad_(True) for k, v in self.sd.items() if is_param(k)]
        self.opt = torch.optim.Adam(self.params, lr=lr, weight_decay=weight_decay)

    def step(self, x, target, clip_features=None):
        self.opt.zero_grad()
        loss = hybrid_loss(unet_forward(x, self.sd, self.arch, True, clip_features), target)
        loss.backward()
        self.opt.step()
        return float(loss.detach())

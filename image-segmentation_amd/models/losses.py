"""Losses and metrics with the reference's API (reference: models/losses.py)."""
import torch
import torch.nn as nn

from hipseg import ops



class HybridLoss(nn.Module):
    """forward() is cross-entropy only, exactly as the reference (losses.py:13-15); the Dice and
    confusion terms the reference constructs there are never used by its forward."""

    def forward(self, pred, target):
        ops._require_gpu(pred)
        return ops.CrossEntropyFn.apply(pred.float().contiguous(), target.long().contiguous())


class HybridLossBinary(nn.Module):
    """BCEWithLogits (mean) + smp DiceLoss(mode='binary') applied to sigmoid(pred)
    (reference: losses.py:17-36; smp 0.4.0 defaults from_logits=True, smooth 0, eps 1e-7 --
    third-party arithmetic, parity unpinned)."""

    def forward(self, pred, target):
        ops._require_gpu(pred)
        if target.dim() == 3:
            target = target.unsqueeze(1)
        return ops.BceDiceFn.apply(pred.float().contiguous(), target.float().contiguous())


def _conf(preds, targets):
    return ops.confusion_matrix(preds, targets).double()


class IoU(nn.Module):
    """mean over classes of (inter+eps)/(union+eps), argmax predictions (reference: losses.py:38-63).
    One fused argmax + confusion-matrix kernel instead of softmax/argmax + a Python loop per class."""

    def __init__(self, eps=1e-6):
        super().__init__()
        self.eps = eps

    def forward(self, preds, targets):
        c = _conf(preds, targets)
        inter = c.diagonal()
        union = c.sum(0) + c.sum(1) - inter
        return ((inter + self.eps) / (union + self.eps)).mean().float()


class PixelAccuracy(nn.Module):
    """mean per-class recall over the classes present in the target, 3 classes
    (reference: losses.py:129-154)."""

    def forward(self, preds, targets):
        c = _conf(preds, targets)[:3, :]
        tot = c.sum(1)
        present = tot > 0
        acc = c.diagonal()[:3] / tot.clamp_min(1)
        return (acc * present).sum().div(present.sum()).float()


class IoUBinary(nn.Module):
    """per-sample IoU of (sigmoid(pred) > threshold), then mean (reference: losses.py:65-90).
    Validation-only metric: evaluated with elementwise torch ops."""

    def __init__(self, eps=1e-6, threshold=0.5):
        super().__init__()
        self.eps, self.threshold = eps, threshold

    def forward(self, preds, targets):
        p = (torch.sigmoid(preds.float()) > self.threshold).float().squeeze(1)
        t = targets.float() if targets.dim() == 3 else targets.float().squeeze(1)
        inter = (p * t).sum(dim=[1, 2])
        union = p.sum(dim=[1, 2]) + t.sum(dim=[1, 2]) - inter
        return ((inter + self.eps) / (union + self.eps)).mean()


class PixelAccuracyBinary(nn.Module):
    """(reference: losses.py:156-180)"""

    def __init__(self, threshold=0.5):
        super().__init__()
        self.threshold = threshold

    def forward(self, preds, targets):
        p = (torch.sigmoid(preds.float()) > self.threshold).float().squeeze(1)
        t = (targets.squeeze(1) if targets.dim() == 4 else targets).float()
        return (p == t).float().sum() / t.numel()


def _soft_dice_score(p, t, dims, smooth=0.0, eps=1e-7):
    """smp 0.4.0 `soft_dice_score`: (2*sum(p*t) + smooth) / clamp_min(sum(p + t) + smooth, eps) over `dims`."""
    inter = (p * t).sum(dims)
    card = (p + t).sum(dims)
    return (2.0 * inter + smooth) / (card + smooth).clamp_min(eps)


class Dice(nn.Module):
    """1 - smp.losses.DiceLoss(mode='multiclass')(softmax(preds), targets) (reference: losses.py:92-100).

    smp 0.4.0 (absent third-party arithmetic -- PARITY UNPINNED, restated from its published algorithm, see
    DESIGN.md): `from_logits=True` applies log_softmax(dim=1).exp() to its input again, so the score is computed on
    softmax(softmax(preds)); per-class soft dice over dims (batch, pixels) with smooth 0 / eps 1e-7, classes absent
    from the target contribute loss 0, mean over classes.  Validation-only metric (the wrappers construct it and
    then report 2*IoU/(1+IoU) instead, model_wrappers.py:151,211): evaluated with elementwise torch ops."""

    def __init__(self, eps=1e-6):
        super().__init__()

    def forward(self, preds, targets):
        B, C = preds.shape[0], preds.shape[1]
        p = torch.softmax(torch.softmax(preds.float(), dim=1), dim=1).reshape(B, C, -1)
        t = torch.nn.functional.one_hot(targets.reshape(B, -1).long(), C).permute(0, 2, 1).to(p.dtype)
        score = _soft_dice_score(p, t, (0, 2))
        loss = (1.0 - score) * (t.sum((0, 2)) > 0).to(p.dtype)
        return 1.0 - loss.mean()


class DiceBinary(nn.Module):
    """1 - smp.losses.DiceLoss(mode='binary')(sigmoid(preds), targets) (reference: losses.py:102-126); same smp
    semantics as the Dice term of HybridLossBinary (sigmoid applied twice), PARITY UNPINNED."""

    def __init__(self, eps=1e-6, threshold=0.5):
        super().__init__()
        self.eps, self.threshold = eps, threshold

    def forward(self, preds, targets):
        B = preds.shape[0]
        t = (targets.unsqueeze(1) if targets.dim() == 3 else targets).float().reshape(B, 1, -1)
        p = torch.sigmoid(torch.sigmoid(preds.float())).reshape(B, 1, -1)
        score = _soft_dice_score(p, t, (0, 2))
        loss = (1.0 - score) * (t.sum((0, 2)) > 0).to(p.dtype)
        return 1.0 - loss.mean()


class CombinedConfusionLoss(nn.Module):
    """Per-pixel cross-entropy, x incorrect_penalty where argmax != target, x confusion_penalty again where the
    (prediction, target) pair is one of `confusion_pairs` in either order; mean (reference: losses.py:182-214).
    Constructed by the reference's HybridLoss but never used in its forward; off the hot path, elementwise torch ops."""

    def __init__(self, incorrect_penalty=2.0, confusion_pairs=[(1, 2)], confusion_penalty=2.0):  # noqa: B006
        super().__init__()
        self.incorrect_penalty = incorrect_penalty
        self.confusion_pairs = confusion_pairs
        self.confusion_penalty = confusion_penalty

    def forward(self, pred, target):
        pred = pred.float()
        loss = torch.nn.functional.cross_entropy(pred, target, reduction="none")
        cls = torch.softmax(pred, dim=1).argmax(dim=1)
        w = torch.where(cls != target, self.incorrect_penalty, 1.0)
        for c1, c2 in self.confusion_pairs:
            conf = ((cls == c1) & (target == c2)) | ((cls == c2) & (target == c1))
            w = torch.where(conf, w * self.confusion_penalty, w)
        return (loss * w).mean()

"""CPU restatement of the on-device augmentation pipeline, given the sampled parameter table.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  PARITY UNPINNED: the reference composes kornia 0.8.0 modules
(models/processing_blocks.py:344-384,386-451) and kornia is absent from this image, with no fixture in the reference;
every function below restates kornia's published algorithm from its documentation and cites it.  The random
parameters are inputs here (table layout of include/hipseg.h `hipseg_augment`), so the HIP kernels can be checked
element-wise against this file on identical parameters.
"""
import math

import torch
import torch.nn.functional as F


def _gray(img):
    """kornia.color.rgb_to_grayscale default weights."""
    return 0.299 * img[0] + 0.587 * img[1] + 0.114 * img[2]


def _rgb_to_hsv(img, eps=1e-8):
    """kornia.color.rgb_to_hsv (h in radians [0, 2pi))."""
    mx, arg = img.max(0)
    mn = img.min(0)[0]
    dc = mx - mn
    v = mx
    s = dc / (mx + eps)
    dc = torch.where(dc == 0, torch.ones_like(dc), dc)
    rc, gc, bc = mx - img[0], mx - img[1], mx - img[2]
    h = torch.stack([bc - gc, (rc - bc) + 2.0 * dc, (gc - rc) + 4.0 * dc], 0)
    h = torch.gather(h, 0, arg[None])[0] / dc
    h = (h / 6.0) % 1.0
    return 2.0 * math.pi * h, s, v


def _hsv_to_rgb(h, s, v):
    """kornia.color.hsv_to_rgb."""
    h = h / (2.0 * math.pi)
    hi = torch.floor(h * 6.0) % 6
    f = ((h * 6.0) % 6) - hi
    p, q, t = v * (1.0 - s), v * (1.0 - f * s), v * (1.0 - (1.0 - f) * s)
    hi = hi.long()
    r = torch.stack([v, q, p, p, t, v], 0)
    g = torch.stack([t, v, v, q, p, p], 0)
    b = torch.stack([p, p, t, v, v, q], 0)
    return torch.stack([torch.gather(c, 0, hi[None])[0] for c in (r, g, b)], 0)


def colour_jitter(img, p, order):
    """kornia.augmentation.ColorJitter: the four ops in the sampled order.  img (3,H,W) fp32."""
    for op in order:
        if op == 0:    # adjust_brightness_accumulative
            img = (img * p[4]).clamp(0.0, 1.0)
        elif op == 1:  # adjust_contrast_with_mean_subtraction
            m = _gray(img).mean()
            img = ((img - m) * p[5] + m).clamp(0.0, 1.0)
        elif op == 2:  # adjust_saturation_with_gray_subtraction
            y = _gray(img)
            img = ((img - y) * p[6] + y).clamp(0.0, 1.0)
        else:          # adjust_hue: shift in radians
            h, s, v = _rgb_to_hsv(img)
            h = torch.fmod(h + p[7], 2.0 * math.pi)
            img = _hsv_to_rgb(h, s, v)
    return img


def gaussian_blur5(img, sigma):
    """kornia.filters.gaussian_blur2d((5,5), sigma, border_type='reflect'), separable."""
    x = torch.arange(5, dtype=torch.float32) - 2.0
    k = torch.exp(-x * x / (2.0 * sigma * sigma))
    k = k / k.sum()
    k2 = (k[:, None] * k[None, :])[None, None].expand(3, 1, 5, 5)
    return F.conv2d(F.pad(img[None], (2, 2, 2, 2), mode="reflect"), k2, groups=3)[0]


def source_index(p, H, W):
    """(H,W) long tensor of flat source indices for flip-then-rotate (nearest, zeros padding; -1 = outside).
    kornia RandomHorizontalFlip then RandomRotation -> warp_affine(nearest, zeros, align_corners=True) about
    ((W-1)/2, (H-1)/2)."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    cx, cy = 0.5 * (W - 1), 0.5 * (H - 1)
    c, s = float(p[2]), float(p[3])
    dx, dy = xs - cx, ys - cy
    fx = torch.tensor(c) * dx - torch.tensor(s) * dy + cx
    fy = torch.tensor(s) * dx + torch.tensor(c) * dy + cy
    sx, sy = torch.round(fx).long(), torch.round(fy).long()  # torch.round: half to even, like nearbyint
    inside = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
    if p[1] != 0:
        sx = W - 1 - sx
    return torch.where(inside, sy * W + sx, torch.full_like(sx, -1))


def augment(images, masks, extra, params, order):
    """same contract as hipseg_augment / ops.augment, on CPU tensors."""
    images = images.float()
    B, _, H, W = images.shape
    out = torch.empty_like(images)
    om = None if masks is None else torch.empty_like(masks)
    oe = None if extra is None else torch.empty_like(extra)
    order = [int(o) for o in order]
    for b in range(B):
        p = params[b]
        if p[0] != 0:
            out[b] = images[b]
            if om is not None:
                om[b] = masks[b]
            if oe is not None:
                oe[b] = extra[b]
            continue
        idx = source_index(p, H, W).reshape(-1)
        ok = idx >= 0
        safe = idx.clamp_min(0)
        g = images[b].reshape(3, -1)[:, safe] * ok
        img = colour_jitter(g.reshape(3, H, W), p, order)
        out[b] = gaussian_blur5(img, float(p[8]))
        if om is not None:
            om[b] = (masks[b].reshape(-1)[safe] * ok).reshape(H, W)
        if oe is not None:
            oe[b] = (extra[b].reshape(extra.shape[1], -1)[:, safe] * ok).reshape(-1, H, W)
    return out, om, oe

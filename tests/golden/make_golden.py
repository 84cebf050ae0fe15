#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the REFERENCE itself.

Run ONLY in the build container (needs /root/reference; it never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Recipe: import the real `transformers`/`datasets`, stub
the absent third-party packages that are touched only OFF the hot path
(torchvision, kornia, segmentation_models_pytorch), then import the
reference's `models.*`.  Weights and inputs come from `oracle.fill`
(pure functions of name/shape/seed), so fixtures hold expected OUTPUTS only.
All arithmetic: PyTorch CPU fp32, torch.set_num_threads(1) for run-to-run
stable reductions.
"""
import os
import sys
from unittest.mock import MagicMock

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import transformers, datasets  # noqa: E401,F401  (real packages first)

for _n in ["torchvision", "torchvision.models", "torchvision.transforms", "torchvision.transforms.v2",
           "kornia", "kornia.augmentation", "kornia.filters", "segmentation_models_pytorch"]:
    sys.modules[_n] = MagicMock()

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from models.UNet import UNet, LargeUNet  # noqa: E402  (reference)
from models.processing_blocks import (ConvBlock, ConvBlockDownsample, ConvBlockUpsampleSkip,  # noqa: E402
                                      ConvBlockUpsample, CrossAttentionFusion)
from models.losses import (HybridLoss, IoU, IoUBinary, PixelAccuracy, PixelAccuracyBinary,  # noqa: E402
                           CombinedConfusionLoss)
import models.CLIP_models as ref_clip  # noqa: E402

from oracle import fill  # noqa: E402

torch.set_num_threads(1)


def T(name, shape, lo=0.0, hi=1.0):
    return torch.from_numpy(fill.uniform(name, shape, lo, hi))


def npy(t):
    return t.detach().cpu().numpy().copy()


# ------------------------------------------------------------------ block-level fixtures
BLOCK_CASES = [
    # name, class, ctor args, input shapes (x[, skip])
    ("cb_4_8", "ConvBlock", (4, 8), [(2, 4, 16, 16)]),
    ("cb_32_64", "ConvBlock", (32, 64), [(2, 32, 16, 16)]),
    ("cb_8_8_odd", "ConvBlock", (8, 8), [(1, 8, 8, 24)]),
    ("down_8_16", "ConvBlockDownsample", (8, 16), [(2, 8, 16, 16)]),
    ("upskip_16_8_identity", "ConvBlockUpsampleSkip", (16, 8), [(2, 16, 8, 8), (2, 8, 16, 16)]),
    ("upskip_16_8_dec1", "ConvBlockUpsampleSkip", (16, 8), [(2, 16, 8, 8), (2, 8, 8, 8)]),
    ("upskip_64_32_dec1", "ConvBlockUpsampleSkip", (64, 32), [(1, 64, 4, 12), (1, 32, 4, 12)]),
    ("up_16_8", "ConvBlockUpsample", (16, 8), [(2, 16, 8, 8)]),
]
_CLS = dict(ConvBlock=ConvBlock, ConvBlockDownsample=ConvBlockDownsample,
            ConvBlockUpsampleSkip=ConvBlockUpsampleSkip, ConvBlockUpsample=ConvBlockUpsample)


def gen_blocks():
    out = {}
    for name, cls, args, shapes in BLOCK_CASES:
        m = _CLS[cls](*args)
        fill.fill_state_dict(m.state_dict(), prefix=name + ".")
        ins = [T(f"{name}.in{i}", s, -1.0, 1.0).requires_grad_(True) for i, s in enumerate(shapes)]
        # eval first (running stats as filled), then train
        m.eval()
        with torch.no_grad():
            out[f"{name}/eval_out"] = npy(m(*ins))
        m.train()
        y = m(*ins)
        g = T(f"{name}.gout", tuple(y.shape), -1.0, 1.0)
        (y * g).sum().backward()
        out[f"{name}/train_out"] = npy(y)
        for i, t in enumerate(ins):
            out[f"{name}/grad_in{i}"] = npy(t.grad)
        for k, p in m.named_parameters():
            out[f"{name}/grad/{k}"] = npy(p.grad)
        for k, b in m.named_buffers():
            out[f"{name}/buf/{k}"] = npy(b)
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **out)
    print("blocks.npz", len(out))


# ------------------------------------------------------------------ whole-model fixtures
def _model_case(out, tag, model, x, target, adam_steps=0, grad_keys=(), bf16_yardstick=False):
    fill.fill_state_dict(model.state_dict())
    model.eval()
    with torch.no_grad():
        out[f"{tag}/eval_logits"] = npy(model(x))
    model.train()
    logits = model(x)
    loss = HybridLoss()(logits, target)
    loss.backward()
    out[f"{tag}/train_logits"] = npy(logits)
    out[f"{tag}/ce_loss"] = npy(loss)
    if bf16_yardstick:
        # the reference's OWN reduced-precision path (torch.autocast bf16 on CPU), as a yardstick for
        # what "bf16" costs on these weights; buffers are restored so the fixtures above stay valid
        saved = {k: v.clone() for k, v in model.state_dict().items()}
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
            model.eval()
            out[f"{tag}/bf16ref_eval_logits"] = npy(model(x).float())
            model.train()
            out[f"{tag}/bf16ref_train_logits"] = npy(model(x).float())
        model.load_state_dict(saved)
    for k, p in model.named_parameters():
        g = p.grad
        out[f"{tag}/gradstat/{k}"] = np.array([float(g.double().sum()), float(g.double().abs().sum()),
                                               float(g.double().pow(2).sum())])
        if k in grad_keys:
            out[f"{tag}/grad/{k}"] = npy(g)
    for k in ("enc1.block.0.conv.1.running_mean", "enc1.block.0.conv.1.running_var",
              "bottleneck.conv.4.running_mean", "bottleneck.conv.4.running_var",
              "bottleneck.conv.4.num_batches_tracked"):
        out[f"{tag}/buf/{k}"] = npy(model.state_dict()[k])
    if adam_steps:
        # loop body of models/model_wrappers.py:167-177 (fp32, no autocast): Adam lr 1e-3 wd 1e-4
        fill.fill_state_dict(model.state_dict())
        opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
        traj = []
        for _ in range(adam_steps):
            opt.zero_grad()
            l = HybridLoss()(model(x), target)
            l.backward()
            opt.step()
            traj.append(float(l))
        out[f"{tag}/adam_traj"] = np.array(traj)


def gen_models():
    out = {}
    x = T("c1.x", (2, 3, 128, 128))
    t = torch.from_numpy(fill.randint("c1.t", (2, 128, 128), 3))
    _model_case(out, "unet_c1", UNet(), x, t, adam_steps=5, bf16_yardstick=True,
                grad_keys=("input.weight", "input.bias", "enc1.block.0.conv.0.weight",
                           "enc1.block.0.conv.1.weight", "enc1.block.0.conv.1.bias",
                           "bottleneck.conv.3.bias", "dec1.up.bias", "dec4.up.weight",
                           "dec4.conv.conv.3.weight", "out.weight", "out.bias"))
    x = T("large.x", (1, 3, 64, 64))
    t = torch.from_numpy(fill.randint("large.t", (1, 64, 64), 3))
    _model_case(out, "large_64", LargeUNet(), x, t,
                grad_keys=("input.weight", "out.weight", "dec5.up.bias"))
    # odd-ish geometry (ClipUnet config 5 is 224 -> 28x28 at the bottleneck)
    x = T("unet56.x", (1, 3, 56, 40))
    t = torch.from_numpy(fill.randint("unet56.t", (1, 56, 40), 3))
    _model_case(out, "unet_56x40", UNet(), x, t, grad_keys=("input.weight", "out.weight"))
    np.savez_compressed(os.path.join(HERE, "models.npz"), **out)
    print("models.npz", len(out))


def gen_clip():
    """ClipUnet trunk with an injected feature extractor (the real one is a network fetch)."""
    out = {}
    feats = T("clip.feats", (2, 512), -1.0, 1.0)

    class FakeExtractor(nn.Module):
        def __init__(self, train=False):
            super().__init__()

        def forward(self, x):
            return feats

    ref_clip.ClipFeatureExtractor = FakeExtractor
    m = ref_clip.ClipUnet()
    fill.fill_state_dict(m.state_dict())
    x = T("clip.x", (2, 3, 32, 32))
    t = torch.from_numpy(fill.randint("clip.t", (2, 32, 32), 3))
    m.eval()
    with torch.no_grad():
        out["clip/eval_logits"] = npy(m(x))
    m.train()
    logits = m(x)
    loss = HybridLoss()(logits, t)
    loss.backward()
    out["clip/train_logits"] = npy(logits)
    out["clip/ce_loss"] = npy(loss)
    for k, p in m.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        out[f"clip/gradstat/{k}"] = np.array([float(g.double().sum()), float(g.double().abs().sum()),
                                              float(g.double().pow(2).sum())])
    out["clip/grad/out_proj.bias"] = npy(m.cross_attention_fusion.cross_attn.out_proj.bias.grad)
    out["clip/grad/in_proj_weight_rowabs"] = npy(m.cross_attention_fusion.cross_attn.in_proj_weight.grad.abs().sum(1))
    # the fusion module alone
    caf = CrossAttentionFusion(512, num_heads=1)
    fill.fill_state_dict(caf.state_dict(), prefix="cross_attention_fusion.")
    b = T("caf.bott", (2, 512, 4, 4), -1.0, 1.0)
    with torch.no_grad():
        out["caf/out"] = npy(caf(b, feats))
    np.savez_compressed(os.path.join(HERE, "clip.npz"), **out)
    print("clip.npz", len(out))


def gen_losses():
    out = {}
    logits = T("loss.logits", (2, 3, 32, 32), -3.0, 3.0)
    tgt = torch.from_numpy(fill.randint("loss.t", (2, 32, 32), 3))
    lg = logits.clone().requires_grad_(True)
    ce = HybridLoss()(lg, tgt)
    ce.backward()
    out["ce"] = npy(ce)
    out["ce_grad"] = npy(lg.grad)
    out["iou"] = npy(IoU()(logits, tgt))
    out["pixel_accuracy"] = npy(PixelAccuracy()(logits, tgt))
    # missing-class case for PixelAccuracy (class 2 absent)
    tgt2 = torch.from_numpy(fill.randint("loss.t2", (2, 32, 32), 2))
    out["iou_2cls"] = npy(IoU()(logits, tgt2))
    out["pixel_accuracy_2cls"] = npy(PixelAccuracy()(logits, tgt2))
    bl = T("loss.blogits", (2, 1, 32, 32), -3.0, 3.0)
    bt = torch.from_numpy(fill.randint("loss.bt", (2, 32, 32), 2)).float()
    # BCE half of HybridLossBinary (models/losses.py:33); the Dice half is smp (absent) -> unpinned
    blg = bl.clone().requires_grad_(True)
    bce = nn.BCEWithLogitsLoss()(blg, bt.unsqueeze(1))
    bce.backward()
    out["bce"] = npy(bce)
    out["bce_grad"] = npy(blg.grad)
    out["iou_binary"] = npy(IoUBinary()(bl, bt))
    out["pixel_accuracy_binary"] = npy(PixelAccuracyBinary()(bl, bt))
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("losses.npz", len(out))


# ------------------------------------------------------------------ dataset-record decode
def gen_records():
    """Outputs of the reference's OWN `CustomImageDataset._deserialize_datapoint` (customDatasets/datasets.py:92-135)
    on the synthetic records of oracle.records.make_records().  The class cannot be constructed offline (its
    __init__ downloads the dataset), so the two decode methods are called on a bare instance."""
    sys.modules.setdefault("scripts.dataset_downloader", MagicMock())
    from customDatasets.datasets import CustomImageDataset  # reference
    from oracle import records
    images, masks = records.make_records()
    ds = CustomImageDataset.__new__(CustomImageDataset)
    out = {"mask_in": masks}
    lut = np.full(256, np.nan, np.float32)
    for i in range(images.shape[0]):
        img, msk = ds._deserialize_datapoint({"image": images[i].tobytes(), "mask": masks[i].tobytes()})
        img, msk = npy(img), npy(msk)
        assert img.shape == (3, 256, 256) and img.dtype == np.float32 and msk.shape == (256, 256)
        out[f"mask_out_{i}"] = msk.astype(np.uint8)          # values 0..2
        out[f"mask_dtype_{i}"] = np.array(str(msk.dtype))
        out[f"image_sum_{i}"] = img.astype(np.float64).sum((1, 2))  # per channel: pins the HWC -> CHW permutation
        out[f"image_samples_{i}"] = img[:, ::37, ::41].copy()
        # the image output is a function of the byte value alone: record the reference's value for every byte
        hwc = np.transpose(img, (1, 2, 0))
        lut[images[i].reshape(-1)] = hwc.reshape(-1)
        assert np.array_equal(lut[images[i]], hwc)
    assert not np.isnan(lut).any()
    out["image_lut"] = lut
    np.savez_compressed(os.path.join(HERE, "records.npz"), **out)
    print("records.npz", len(out))


# ------------------------------------------------------------------ round-2 fixtures (separate files: the round-1 ones stay)
TRAINED_PREFIXES = ("dec4.", "out.")  # the parameters the reference trains for the confident-logits fixture


def blob_task(tag, n, size=64, cells=8):
    """Learnable synthetic segmentation task: smooth random colour fields (cells x cells uniform noise, bilinearly
    up-sampled to size x size); the class of a pixel is its brightest channel.  Pure function of (tag, n, size)."""
    low = T(f"{tag}.low", (n, 3, cells, cells))
    x = torch.nn.functional.interpolate(low, size=(size, size), mode="bilinear", align_corners=True)
    x = (x - x.amin((1, 2, 3), keepdim=True)) / (x.amax((1, 2, 3), keepdim=True) - x.amin((1, 2, 3), keepdim=True))
    return x.contiguous(), x.argmax(1)


def gen_round2():
    out = {}
    # (1) LargeUNet at a size that reaches the 1024-channel layers with H = 8 > the 4x4 of large_64
    x = T("large128.x", (1, 3, 128, 128))
    t = torch.from_numpy(fill.randint("large128.t", (1, 128, 128), 3))
    _model_case(out, "large_128", LargeUNet(), x, t, grad_keys=("input.weight", "out.weight", "bottleneck.conv.3.bias"))
    # (2) reference-TRAINED confident logits: the reference UNet with oracle.fill weights, its last decoder block and
    # head (dec4.*, out.*: 36 K parameters, small enough to commit) trained by the reference's own loop body
    # (models/model_wrappers.py:167-177, fp32) on the blob task; every BatchNorm's running statistics move too.
    # The fp32 logits/masks of the trained reference on held-out images are what the bf16 HIP path must reproduce
    # within 1e-2 IoU (north star) -- margins here are far above bf16 noise, unlike the untrained unet_c1 fixture.
    m = UNet()
    fill.fill_state_dict(m.state_dict())
    for k, p in m.named_parameters():
        p.requires_grad_(k.startswith(TRAINED_PREFIXES))
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=3e-3, weight_decay=1e-4)
    xtr, ttr = blob_task("blob.train", 4)
    crit = HybridLoss()
    m.train()
    traj = []
    for _ in range(60):
        opt.zero_grad()
        loss = crit(m(xtr), ttr)
        loss.backward()
        opt.step()
        traj.append(float(loss))
    out["trained/loss_traj"] = np.array(traj)
    for k, v in m.state_dict().items():
        if k.startswith(TRAINED_PREFIXES) or k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            out[f"trained/state/{k}"] = npy(v)
    xte, tte = blob_task("blob.test", 4)
    m.eval()
    with torch.no_grad():
        lg = m(xte)
    out["trained/eval_logits"] = npy(lg)
    out["trained/target"] = npy(tte).astype(np.uint8)
    top2 = lg.topk(2, dim=1).values
    out["trained/median_margin"] = npy((top2[:, 0] - top2[:, 1]).median())
    out["trained/iou_vs_target"] = npy(IoU()(lg, tte))
    saved = {k: v.clone() for k, v in m.state_dict().items()}
    m.train()
    with torch.no_grad():
        out["trained/train_logits"] = npy(m(xte))
    m.load_state_dict(saved)
    # (2b) ClipUnet: the reference runs its (output-discarded) bottleneck ConvBlock, so one train-mode forward moves the
    # bottleneck's BatchNorm buffers -- checkpoint state the drop-in must reproduce
    feats = T("clip.feats", (2, 512), -1.0, 1.0)

    class FakeExtractor(nn.Module):
        def __init__(self, train=False):
            super().__init__()

        def forward(self, x):
            return feats

    ref_clip.ClipFeatureExtractor = FakeExtractor
    cm = ref_clip.ClipUnet()
    fill.fill_state_dict(cm.state_dict())
    cm.train()
    with torch.no_grad():
        cm(T("clip.x", (2, 3, 32, 32)))
    for k, v in cm.state_dict().items():
        if k.startswith("bottleneck.") and k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            out[f"clip_bn/{k}"] = npy(v)
    # ... and its parameters get a gradient TENSOR (rounding residue of a mathematically zero gradient), so the
    # reference's Adam(weight_decay=1e-4) (model_wrappers.py:43) keeps shrinking them: three of its loop-body steps in
    # fp32, then a few rows of the dead and of the live weights
    cm2 = ref_clip.ClipUnet()
    fill.fill_state_dict(cm2.state_dict())
    cm2.train()
    opt2 = torch.optim.Adam(cm2.parameters(), lr=1e-3, weight_decay=1e-4)
    xs, ts = T("clip.x", (2, 3, 32, 32)), torch.from_numpy(fill.randint("clip.t", (2, 32, 32), 3))
    for _ in range(3):
        opt2.zero_grad()
        HybridLoss()(cm2(xs), ts).backward()
        opt2.step()
    sd2 = cm2.state_dict()
    out["clip_adam3/bottleneck.conv.0.weight[:4]"] = npy(sd2["bottleneck.conv.0.weight"][:4])
    out["clip_adam3/bottleneck.conv.4.weight"] = npy(sd2["bottleneck.conv.4.weight"])
    out["clip_adam3/dec1.up.weight[:2]"] = npy(sd2["dec1.up.weight"][:2])
    out["clip_adam3/out.weight"] = npy(sd2["out.weight"])
    # (2c) ClipAutoencoder (models/CLIP_models.py:136-188): Linear coupler -> ConvBlockUpsample x3 -> ConvBlockUpsampleSkip
    ca = ref_clip.ClipAutoencoder()
    fill.fill_state_dict(ca.state_dict())
    xa = T("clipae.x", (2, 3, 32, 32))
    ta = torch.from_numpy(fill.randint("clipae.t", (2, 32, 32), 3))
    ca.eval()
    with torch.no_grad():
        out["clipae/eval_logits"] = npy(ca(xa))
    ca.train()
    lga = ca(xa)
    la = HybridLoss()(lga, ta)
    la.backward()
    out["clipae/train_logits"] = npy(lga)
    out["clipae/ce_loss"] = npy(la)
    for k in ("coupler.weight", "dec1.up.weight", "dec4.conv.conv.3.weight", "out.weight", "input.weight"):
        g_ = dict(ca.named_parameters())[k].grad
        out[f"clipae/gradstat/{k}"] = np.array([float(g_.double().sum()), float(g_.double().abs().sum()),
                                                float(g_.double().pow(2).sum())])
    # (2d) production geometries against the reference itself (logits sub-sampled 4x4 to keep the fixture small):
    # UNet 2 x 3 x 256 x 256 (BASELINE config C2's image size; 2 images = 1024 8x16 tiles, the grid at which the bf16
    # path switches to its weights-stationary / ring kernels) and ClipUnet 1 x 3 x 224 x 224 (config C5: 28 x 28 bottleneck)
    def big_case(tag, model, x, t, extra_keys=()):
        fill.fill_state_dict(model.state_dict())
        model.eval()
        with torch.no_grad():
            out[f"{tag}/eval_logits_s4"] = npy(model(x)[:, :, ::4, ::4])
        model.train()
        lg = model(x)
        ls_ = HybridLoss()(lg, t)
        ls_.backward()
        out[f"{tag}/train_logits_s4"] = npy(lg[:, :, ::4, ::4])
        out[f"{tag}/ce_loss"] = npy(ls_)
        out[f"{tag}/train_argmax_hist"] = np.bincount(npy(lg.argmax(1)).reshape(-1), minlength=3)
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            g_ = p.grad
            out[f"{tag}/gradstat/{k}"] = np.array([float(g_.double().sum()), float(g_.double().abs().sum()),
                                                   float(g_.double().pow(2).sum())])
        for k in extra_keys:
            out[f"{tag}/grad/{k}"] = npy(dict(model.named_parameters())[k].grad)
        # the reference's OWN bf16 deviation (its training loop runs under autocast): CPU autocast forward of the same
        # state and batch, relative L2 against its fp32 logits -- the yardstick for the HIP bf16 path's deviation
        for p_ in model.parameters():
            p_.grad = None
        with torch.autocast("cpu", dtype=torch.bfloat16):
            lb = model(x).float()
            lsb = HybridLoss()(lb, t)
        lsb.backward()
        d = (lb.detach() - lg.detach())[:, :, ::4, ::4]
        out[f"{tag}/ref_bf16_autocast_rel_l2"] = np.array(
            float(d.pow(2).sum().sqrt() / lg.detach()[:, :, ::4, ::4].pow(2).sum().sqrt()))
        out[f"{tag}/ref_bf16_autocast_loss"] = npy(lsb)
        for k in ("out.weight", "input.weight"):
            out[f"{tag}/ref_bf16_autocast_grad_sq/{k}"] = np.array(
                float(dict(model.named_parameters())[k].grad.double().pow(2).sum()))

    big_case("unet_256", UNet(), T("u256.x", (2, 3, 256, 256)),
             torch.from_numpy(fill.randint("u256.t", (2, 256, 256), 3)), ("out.weight", "input.weight", "dec4.up.bias"))
    feats1 = T("clip224.feats", (1, 512), -1.0, 1.0)

    class FakeExtractor1(nn.Module):
        def __init__(self, train=False):
            super().__init__()

        def forward(self, x):
            return feats1

    ref_clip.ClipFeatureExtractor = FakeExtractor1
    big_case("clip_224", ref_clip.ClipUnet(), T("clip224.x", (1, 3, 224, 224)),
             torch.from_numpy(fill.randint("clip224.t", (1, 224, 224), 3)), ("out.weight",))
    np.savez_compressed(os.path.join(HERE, "models_r2.npz"), **out)
    print("models_r2.npz", len(out), "final loss", traj[-1], "median margin", float(out["trained/median_margin"]),
          "IoU vs target", float(out["trained/iou_vs_target"]))
    # (3) CombinedConfusionLoss (models/losses.py:182-214), value and gradient
    lo = {}
    logits = T("loss.logits", (2, 3, 32, 32), -3.0, 3.0)
    tgt = torch.from_numpy(fill.randint("loss.t", (2, 32, 32), 3))
    for tag, kw in (("default", {}), ("pairs", {"incorrect_penalty": 1.5, "confusion_pairs": [(0, 1), (1, 2)],
                                                "confusion_penalty": 3.0})):
        lg = logits.clone().requires_grad_(True)
        v = CombinedConfusionLoss(**kw)(lg, tgt)
        v.backward()
        lo[f"ccl_{tag}"] = npy(v)
        lo[f"ccl_{tag}_grad"] = npy(lg.grad)
    np.savez_compressed(os.path.join(HERE, "losses_r2.npz"), **lo)
    print("losses_r2.npz", len(lo))


def gen_round3():
    """Round 3: the binary (1-channel) model step of scripts/prompt_train.py:58 -- the reference UNet(out_channels=1)
    with nn.BCEWithLogitsLoss, the PINNABLE half of HybridLossBinary (models/losses.py:21,33; the Dice half is
    segmentation_models_pytorch, absent)."""
    out = {}
    x = T("bin.x", (2, 3, 64, 64))
    t = torch.from_numpy((fill.uniform("bin.t", (2, 64, 64), 0.0, 1.0) > 0.5).astype(np.float32))
    m = UNet(out_channels=1)
    fill.fill_state_dict(m.state_dict())
    m.eval()
    with torch.no_grad():
        out["unet_bin/eval_logits"] = npy(m(x))
    m.train()
    logits = m(x)
    loss = nn.BCEWithLogitsLoss()(logits, t.unsqueeze(1))  # (B,H,W) target -> unsqueeze, models/losses.py:30-31
    loss.backward()
    out["unet_bin/train_logits"] = npy(logits)
    out["unet_bin/bce_loss"] = npy(loss)
    for k, p in m.named_parameters():
        g = p.grad
        out[f"unet_bin/gradstat/{k}"] = np.array([float(g.double().sum()), float(g.double().abs().sum()),
                                                  float(g.double().pow(2).sum())])
    for k in ("out.weight", "out.bias", "dec4.conv.conv.3.weight", "input.weight"):
        out[f"unet_bin/grad/{k}"] = npy(dict(m.named_parameters())[k].grad)
    np.savez_compressed(os.path.join(HERE, "models_r3.npz"), **out)
    print("models_r3.npz", len(out))


def gen_round3_prompt():
    """Round 3: the reference ClipUnetPrompt (models/prompt_segmentation.py:32-95; the model scripts/prompt_train.py:55
    trains) with an injected CLIP feature vector (the real extractor is a network fetch), nn.BCEWithLogitsLoss on its
    1-channel logits (the pinnable half of HybridLossBinary)."""
    import models.prompt_segmentation as ref_prompt  # reference

    out = {}
    feats = T("prompt.feats", (2, 512), -1.0, 1.0)

    class FakeExtractor(nn.Module):
        def __init__(self, train=False):
            super().__init__()

        def forward(self, x):
            return feats

    ref_prompt.ClipFeatureExtractor = FakeExtractor
    m = ref_prompt.ClipUnetPrompt()
    fill.fill_state_dict(m.state_dict())
    x = T("prompt.x", (2, 3, 64, 64))
    heat = T("prompt.heat", (2, 1, 64, 64))
    t = torch.from_numpy((fill.uniform("prompt.t", (2, 64, 64), 0.0, 1.0) > 0.5).astype(np.float32))
    out["prompt/state_keys"] = np.array(list(m.state_dict().keys()))
    m.eval()
    with torch.no_grad():
        out["prompt/eval_logits"] = npy(m(x, heat))
        pe = m.prompt_encoder(heat)
        out["prompt/eval_prompt_embedding_stat"] = np.array([float(pe.double().sum()), float(pe.double().abs().sum())])
    m.train()
    logits = m(x, heat)
    loss = nn.BCEWithLogitsLoss()(logits, t.unsqueeze(1))
    loss.backward()
    out["prompt/train_logits"] = npy(logits)
    out["prompt/bce_loss"] = npy(loss)
    for k, p in m.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        out[f"prompt/gradstat/{k}"] = np.array([float(g.double().sum()), float(g.double().abs().sum()),
                                                float(g.double().pow(2).sum())])
    params = dict(m.named_parameters())
    for k in ("prompt_fusion.bias", "prompt_encoder.enc1.block.0.conv.0.weight",
              "prompt_encoder.enc2.block.0.conv.3.weight", "out.weight"):
        out[f"prompt/grad/{k}"] = npy(params[k].grad)
    out["prompt/grad/prompt_fusion.weight[:16]"] = npy(params["prompt_fusion.weight"].grad[:16])  # (full: 2 MiB)
    for k, b in m.named_buffers():
        if k.startswith(("prompt_encoder.enc1", "bottleneck")):
            out[f"prompt/buf/{k}"] = npy(b)
    np.savez_compressed(os.path.join(HERE, "prompt_r3.npz"), **out)
    print("prompt_r3.npz", len(out))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "round3":
        gen_round3()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round3_prompt":
        gen_round3_prompt()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "records":  # regenerate one fixture file only
        gen_records()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "round2":
        gen_round2()
        sys.exit(0)
    gen_blocks()
    gen_models()
    gen_clip()
    gen_losses()
    gen_records()
    gen_round2()
    gen_round3()
    gen_round3_prompt()

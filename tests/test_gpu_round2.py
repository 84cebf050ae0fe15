"""Round-2 parity cases (need a real MI355X):
  * LargeUNet 1x3x128x128 vs the reference golden (reaches the 1024-channel layers at 8x8);
  * the reference-TRAINED confident-logits fixture: the north-star bf16 bar (masks within 1e-2 IoU of the reference's
    fp32 CPU forward) asserted against REFERENCE output, plus fp32 1e-4;
  * BASELINE configs C3 (LargeUNet 8x3x512x512) and C5 (ClipUnet 32x3x224x224) at FULL size through size-independent
    properties (the CPU oracle is too slow there);
  * the on-device DataAugmentor kernels vs oracle/augment.py on identical sampled parameters (kornia absent: parity
    unpinned) + the exact / statistical parts of its contract."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import augment as A, fill  # noqa: E402

from test_gpu_parity import M, T, iou_masks  # noqa: E402,F401  (fixture + helpers)


def test_large_unet_128_fp32_vs_reference_golden(M, golden):
    g = golden("models_r2")
    m = M.un.LargeUNet()
    fill.fill_state_dict(m.state_dict())
    m = m.cuda()
    x = T("large128.x", (1, 3, 128, 128)).cuda()
    t = torch.from_numpy(fill.randint("large128.t", (1, 128, 128), 3)).cuda()
    with M.hipseg.precision_mode("fp32"):
        m.eval()
        with torch.no_grad():
            ev = m(x)
        m.train()
        logits = m(x)
        loss = M.ls.HybridLoss()(logits, t)
        loss.backward()
    torch.cuda.synchronize()
    assert np.abs(ev.cpu().numpy() - g["large_128/eval_logits"]).max() <= 1e-4
    assert np.abs(logits.detach().cpu().numpy() - g["large_128/train_logits"]).max() <= 1e-4
    assert abs(float(loss.detach()) - float(g["large_128/ce_loss"])) <= 1e-5
    for k, p in m.named_parameters():
        if k.endswith(("conv.0.bias", "conv.3.bias")):
            continue
        s = g[f"large_128/gradstat/{k}"]
        gd = p.grad.double()
        np.testing.assert_allclose([float(gd.abs().sum()), float(gd.pow(2).sum())], s[1:], rtol=5e-3, atol=1e-6,
                                   err_msg=k)
        gk = f"large_128/grad/{k}"
        if gk in g:
            assert np.abs(p.grad.cpu().numpy() - g[gk]).max() <= 1e-2 * max(np.abs(g[gk]).max(), 1e-4), k
    # bf16 (autocast) through the 1024-channel / K = 9216 layers: finite, close to fp32 in relative L2
    with torch.autocast("cuda"), torch.no_grad():
        tb = m(x).float().cpu().numpy()
    ref = g["large_128/train_logits"]
    assert np.isfinite(tb).all() and np.sqrt(((tb - ref) ** 2).sum() / (ref ** 2).sum()) < 0.1


def _production_geometry_case(M, g, tag, m, x, t, full_grads, dead_prefix=None):
    """fp32 HIP vs the reference's own output at a BASELINE image size (logits sub-sampled 4x4 in the fixture), then
    the bf16 production kernels (weights-stationary / ring / one-tap / DMA wgrad at their real grid sizes) against the
    same reference numbers within bf16 tolerance."""
    crit = M.ls.HybridLoss()
    with M.hipseg.precision_mode("fp32"):
        m.eval()
        with torch.no_grad():
            ev = m(x)
        m.train()
        logits = m(x)
        loss = crit(logits, t)
        loss.backward()
    torch.cuda.synchronize()
    ref_t, ref_e = g[f"{tag}/train_logits_s4"], g[f"{tag}/eval_logits_s4"]
    assert np.abs(ev[:, :, ::4, ::4].cpu().numpy() - ref_e).max() <= 1e-4 * max(1.0, np.abs(ref_e).max())
    assert np.abs(logits.detach()[:, :, ::4, ::4].cpu().numpy() - ref_t).max() <= 1e-4 * max(1.0, np.abs(ref_t).max())
    assert abs(float(loss.detach()) - float(g[f"{tag}/ce_loss"])) <= 1e-5
    hist = np.bincount(logits.argmax(1).cpu().numpy().reshape(-1), minlength=3)
    assert np.abs(hist - g[f"{tag}/train_argmax_hist"]).sum() <= 1e-3 * hist.sum()
    n = 0
    for k, p in m.named_parameters():
        sk = f"{tag}/gradstat/{k}"
        if sk not in g or k.endswith(("conv.0.bias", "conv.3.bias")):  # conv bias before BN: true gradient is 0 (+noise)
            continue
        gd = p.grad.double()
        if dead_prefix and k.startswith(dead_prefix):
            # ClipUnet's bottleneck: the reference holds a gradient TENSOR of pure rounding residue (|g|_1 ~ 1e-5 over
            # 3.5 M entries); the drop-in hands the optimizer exact zeros (not None: weight decay must keep acting)
            assert g[sk][1] < 1e-4 and float(gd.abs().sum()) == 0.0, k
            continue
        np.testing.assert_allclose([float(gd.abs().sum()), float(gd.pow(2).sum())], g[sk][1:], rtol=5e-3, atol=1e-6, err_msg=k)
        n += 1
    assert n >= 30
    for k in full_grads:
        ref = g[f"{tag}/grad/{k}"]
        got = dict(m.named_parameters())[k].grad.cpu().numpy()
        # 1e-2 of the largest entry, as the LargeUNet case: the first layer's gradient sits behind 18 BN backward
        # passes, each a cancelling sum over 131k pixels, and fp32 summation ORDER differs from the CPU's
        assert np.abs(got - ref).max() <= 1e-2 * max(np.abs(ref).max(), 1e-4), k
    # bf16 production path against the same reference numbers
    for p in m.parameters():
        p.grad = None
    with torch.autocast("cuda"):
        assert M.hipseg.precision() == "bf16"
        lb = m(x)
        loss_b = crit(lb.float(), t)
    loss_b.backward()
    torch.cuda.synchronize()
    tb = lb.detach().float()[:, :, ::4, ::4].cpu().numpy()
    assert np.isfinite(tb).all()
    # yardstick: the reference's OWN CPU-autocast bf16 forward deviates this much from its fp32 logits on this batch
    # (0.070 UNet-256, 0.063 ClipUnet-224; recorded by make_golden.py) -- the HIP bf16 path must not be worse than 1.25x
    rel = float(np.sqrt(((tb - ref_t) ** 2).sum() / (ref_t ** 2).sum()))
    assert rel <= 1.25 * float(g[f"{tag}/ref_bf16_autocast_rel_l2"]) and rel < 0.1, rel
    assert abs(float(loss_b.detach()) - float(g[f"{tag}/ce_loss"])) <= 2e-2
    assert abs(float(loss_b.detach()) - float(g[f"{tag}/ref_bf16_autocast_loss"])) <= 2e-2
    # gradient energy: bf16 LOSES ~19 % of |d input.weight|^2 on the reference's own autocast backward too (0.000484 vs
    # 0.000600 fp32 for UNet-256); the HIP bf16 path is held to the reference's bf16 numbers, not to fp32
    for k in ("out.weight", "input.weight"):
        gd = dict(m.named_parameters())[k].grad.double()
        got, ref16, ref32 = float(gd.pow(2).sum()), float(g[f"{tag}/ref_bf16_autocast_grad_sq/{k}"]), g[f"{tag}/gradstat/{k}"][2]
        # within 10 % of the reference's bf16 number, OR at least as close to the reference's fp32 number as its own bf16
        # backward gets (round 4: with the 64-channel layers on the 16x16x32 kernel ClipUnet-224's |d input.weight|^2 is
        # 1.107e-3 against 1.083e-3 fp32 and 0.990e-3 reference-bf16 -- 2 % from fp32 where the reference's bf16 is 9 %)
        assert abs(got - ref16) <= 0.1 * ref16 or abs(got - ref32) <= abs(ref16 - ref32), (k, got, ref16, ref32)
        np.testing.assert_allclose(got, ref32, rtol=0.3, err_msg=k)


def test_unet_256_vs_reference_golden(M, golden):
    """UNet at BASELINE C2's image size, 2 x 3 x 256 x 256 (1024 8x16 tiles: the grid where the bf16 dispatch picks its
    weights-stationary and ring kernels), against the reference's own CPU fp32 forward/backward."""
    g = golden("models_r2")
    m = M.un.UNet()
    fill.fill_state_dict(m.state_dict())
    x = T("u256.x", (2, 3, 256, 256)).cuda()
    t = torch.from_numpy(fill.randint("u256.t", (2, 256, 256), 3)).cuda()
    _production_geometry_case(M, g, "unet_256", m.cuda(), x, t, ("out.weight", "input.weight", "dec4.up.bias"))


def test_clip_unet_224_vs_reference_golden(M, golden):
    """ClipUnet at BASELINE C5's image size, 1 x 3 x 224 x 224 (28 x 28 bottleneck, ragged tile edges), with the CLIP
    feature vector replaced by a seeded tensor on both sides (open_clip absent)."""
    g = golden("models_r2")
    feats = T("clip224.feats", (1, 512), -1.0, 1.0).cuda()

    class Fake(torch.nn.Module):
        def forward(self, x):
            return feats

    m = M.cm.ClipUnet(clip_feature_extractor=Fake())
    fill.fill_state_dict(m.state_dict())
    x = T("clip224.x", (1, 3, 224, 224)).cuda()
    t = torch.from_numpy(fill.randint("clip224.t", (1, 224, 224), 3)).cuda()
    _production_geometry_case(M, g, "clip_224", m.cuda(), x, t, ("out.weight",), dead_prefix="bottleneck.")


def test_inference_path_fused_conv_bn_relu_matches_the_unfused_eval_forward(M, golden):
    """model.eval() + torch.no_grad() (the validation loops, model_wrappers.py:193-215) takes the one-kernel
    conv -> BatchNorm(running statistics) -> ReLU path (hipseg_conv_affine_relu); with gradients enabled the same eval
    forward keeps the separate kernels (their backward needs the pre-normalisation tensor).  Both must agree -- and the
    fused one is what the reference goldens' eval logits are checked against in the other tests."""
    g = golden("models_r2")
    m = _trained_unet(M, g).eval()
    x = _blob_inputs("blob.test", 4).cuda()
    with M.hipseg.precision_mode("fp32"):
        with torch.no_grad():
            fused = m(x)
        with torch.enable_grad():
            unfused = m(x)
        assert unfused.requires_grad and not fused.requires_grad
        assert float((fused - unfused.detach()).abs().max()) <= 2e-5
        assert np.abs(fused.cpu().numpy() - g["trained/eval_logits"]).max() <= 1e-4 * max(1.0, np.abs(g["trained/eval_logits"]).max())
    with torch.autocast("cuda"):
        with torch.no_grad():
            fb = m(x).float()
        with torch.enable_grad():
            ub = m(x).float().detach()
    ref = g["trained/eval_logits"]
    # the fused form skips one bf16 rounding (of the pre-normalisation tensor) per layer: at least as close to fp32
    ef = float(np.sqrt(((fb.cpu().numpy() - ref) ** 2).sum() / (ref ** 2).sum()))
    eu = float(np.sqrt(((ub.cpu().numpy() - ref) ** 2).sum() / (ref ** 2).sum()))
    assert ef <= 1.1 * eu + 1e-3, (ef, eu)
    assert iou_masks(fb.argmax(1).cpu().numpy(), ref.argmax(1)) >= 1.0 - 1e-2
    # and nothing of the BatchNorm state moved in eval mode
    sd = m.state_dict()
    for k in sd:
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(g[f"trained/state/{k}"]), k


def _trained_unet(M, g):
    """the reference-trained state: oracle.fill weights everywhere, dec4.* / out.* and every BN buffer from the fixture"""
    m = M.un.UNet()
    sd = m.state_dict()
    fill.fill_state_dict(sd)
    n = 0
    for k in sd:
        key = f"trained/state/{k}"
        if key in g:
            sd[k].copy_(torch.from_numpy(g[key]))
            n += 1
    assert n >= 60
    return m.cuda()


def _blob_inputs(tag, n, size=64, cells=8):
    low = T(f"{tag}.low", (n, 3, cells, cells))
    x = torch.nn.functional.interpolate(low, size=(size, size), mode="bilinear", align_corners=True)
    x = (x - x.amin((1, 2, 3), keepdim=True)) / (x.amax((1, 2, 3), keepdim=True) - x.amin((1, 2, 3), keepdim=True))
    return x.contiguous()


def test_bf16_iou_vs_reference_trained_fixture(M, golden):
    """north star: 'masks within 1e-4 max-abs (fp32) and within 1e-2 IoU (bf16) of the reference PyTorch-CPU forward'
    on REFERENCE-generated logits with confident margins (median top-2 margin ~4.9; tests/golden/make_golden.py
    gen_round2 trains the reference's dec4/out on a learnable task)."""
    g = golden("models_r2")
    m = _trained_unet(M, g)
    x = _blob_inputs("blob.test", 4).cuda()
    ref = g["trained/eval_logits"]
    assert float(g["trained/median_margin"]) > 2.0
    m.eval()
    with M.hipseg.precision_mode("fp32"), torch.no_grad():
        got32 = m(x).cpu().numpy()
    assert np.abs(got32 - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), np.abs(got32 - ref).max()
    with torch.autocast("cuda"), torch.no_grad():
        assert M.hipseg.precision() == "bf16"
        got = m(x).float().cpu().numpy()
    iou = iou_masks(got.argmax(1), ref.argmax(1))
    assert iou >= 1.0 - 1e-2, f"bf16 HIP masks vs reference fp32 masks: IoU {iou}"
    # the model-vs-target IoU metric itself (the 'mask IoU vs ref' half of BASELINE's metric) moves by < 1e-2
    tgt = torch.from_numpy(g["trained/target"].astype(np.int64)).cuda()
    iou_t = float(M.ls.IoU()(torch.from_numpy(got).cuda(), tgt))
    assert abs(iou_t - float(g["trained/iou_vs_target"])) <= 1e-2
    # train-mode (batch statistics) forward of the same state
    m.train()
    with torch.autocast("cuda"), torch.no_grad():
        gt = m(x).float().cpu().numpy()
    assert iou_masks(gt.argmax(1), g["trained/train_logits"].argmax(1)) >= 1.0 - 1e-2


def _properties(M, m, x, t, probe_keys, clip=False):
    """size-independent properties of one full-size train step (as test_full_size_properties_c2)."""
    crit = M.ls.HybridLoss()

    def grads(scale, prec):
        m.zero_grad(set_to_none=True)
        with M.hipseg.precision_mode(prec):
            out = m(x)
            loss = crit(out, t)
        (loss * scale).backward()
        return out.detach(), float(loss), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    o1, l1, g1 = grads(1.0, "bf16")
    o2, l2, g2 = grads(1024.0, "bf16")
    assert np.isfinite(l1) and abs(l1 - l2) < 1e-6 * max(1.0, abs(l1))
    assert torch.equal(o1, o2), "forward is deterministic"
    for k in g1:
        if k.endswith(("conv.0.bias", "conv.3.bias")):
            continue
        a, b = g1[k].double() * 1024.0, g2[k].double()
        assert torch.isfinite(b).all(), k
        assert float((a - b).norm() / (b.norm() + 1e-30)) < 2e-2, k
    o1b, l1b, g1b = grads(1.0, "bf16")
    assert l1b == l1 and torch.equal(o1, o1b)
    for k in g1:
        assert torch.equal(g1[k], g1b[k]), f"non-deterministic gradient {k}"
    o32, l32, g32 = grads(1.0, "fp32")
    rel = float((o1.double() - o32.double()).norm() / o32.double().norm())
    assert rel < 0.1, rel
    assert abs(l1 - l32) < 2e-2
    for k, bar in probe_keys:
        r = float((g1[k].double() - g32[k].double()).norm() / g32[k].double().norm())
        assert r < bar, (k, r)
    return o1


def test_full_size_properties_c3(M):
    """BASELINE config 3: LargeUNet 8x3x512x512 (1024-channel bottleneck at 32x32, 268 MB full-resolution tensors,
    the weights-stationary kernel at 512^2)."""
    m = M.un.LargeUNet()
    fill.fill_state_dict(m.state_dict())
    m = m.cuda().train()
    torch.manual_seed(0)
    x = torch.rand(8, 3, 512, 512, device="cuda")
    t = torch.randint(0, 3, (8, 512, 512), device="cuda")
    _properties(M, m, x, t, (("out.weight", 0.02), ("dec5.conv.conv.3.weight", 0.12), ("bottleneck.conv.0.weight", 0.8)))
    assert int(m.bottleneck.conv[1].num_batches_tracked) == 4
    # eval mode is per-sample independent: batch of 8 == 2 + 6, bit for bit
    m.eval()
    with torch.no_grad(), M.hipseg.precision_mode("bf16"):
        full = m(x)
        parts = torch.cat([m(x[:2]), m(x[2:])], 0)
    assert torch.equal(full, parts)


def test_full_size_properties_c5(M):
    """BASELINE config 5: ClipUnet 32x3x224x224 (28x28 bottleneck geometry; frozen CLIP tower replaced by injected
    (B,512) features -- the pretrained weights are a network fetch)."""
    torch.manual_seed(1)
    feats = torch.randn(32, 512, device="cuda")

    class Feats(torch.nn.Module):
        off = 0

        def forward(self, X):
            return feats[self.off:self.off + X.shape[0]]

    ext = Feats()
    m = M.cm.ClipUnet(clip_feature_extractor=ext)
    fill.fill_state_dict(m.state_dict())
    m = m.cuda().train()
    x = torch.rand(32, 3, 224, 224, device="cuda")
    t = torch.randint(0, 3, (32, 224, 224), device="cuda")
    _properties(M, m, x, t, (("out.weight", 0.02), ("dec4.conv.conv.3.weight", 0.12),
                             ("cross_attention_fusion.cross_attn.out_proj.bias", 0.8)))
    # dead branch (CLIP_models.py:126): exact-zero gradient tensors (the reference's are rounding residue, not None)
    assert float(m.bottleneck.conv[0].weight.grad.abs().sum()) == 0.0
    m.eval()
    with torch.no_grad(), M.hipseg.precision_mode("bf16"):
        full = m(x)
        first = m(x[:16])
        ext.off = 16
        second = m(x[16:])
    # the HIP trunk is per-sample independent (bit-exact under a batch split in C2 / C3); here the two tiny fusion GEMMs
    # (torch.nn.functional.linear on (B,512): rocBLAS picks another kernel for M = 16 than for M = 32, measured 3e-5
    # apart in fp32) feed it bf16 inputs that differ in the last bit for some channels, which the decoder carries to
    # the logits (measured 6e-3 on a 0.83 range); masks must still agree
    parts = torch.cat([first, second], 0)
    assert float((full - parts).abs().max()) <= 2e-2 * float(full.abs().max())
    assert float((full.argmax(1) == parts.argmax(1)).float().mean()) > 0.995


# ---------------------------------------------------------------------------------------------- augmentation
def _aug_inputs(B=10, H=64, W=48, seed=3):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(B, 3, H, W, generator=g)
    msk = torch.randint(0, 3, (B, H, W), generator=g)
    return img, msk


def test_augment_kernels_vs_oracle_on_sampled_parameters(M):
    """HIP pipeline == oracle/augment.py on the SAME parameter table (flip, rotation, jitter in a random order, blur).
    Image within 2e-5; a nearest-neighbour source that sits within float rounding of a pixel boundary may resolve
    differently (FMA contraction), so <= 0.2 % of mask pixels may differ."""
    from models.processing_blocks import DataAugmentor, DataAugmentorPrompt

    img, msk = _aug_inputs()
    for trial in range(4):
        torch.manual_seed(100 + trial)
        aug = DataAugmentor(4).cuda()
        out, om = aug(img.cuda(), msk.cuda())
        params, order = aug.last_params
        ref, rm, _ = A.augment(img, msk, None, params.cpu(), order.cpu().tolist())
        assert out.shape == img.shape and out.dtype == torch.float32 and om.dtype == torch.int64
        bad = (om.cpu() != rm)
        assert float(bad.float().mean()) <= 2e-3, float(bad.float().mean())
        # compare the image away from the (few) pixels whose 5x5 window holds a differently-resolved source
        dil = torch.nn.functional.max_pool2d(bad.float()[:, None], 5, 1, 2)[:, 0] > 0
        err = ((out.cpu() - ref).abs().amax(1) * (~dil)).max()
        assert float(err) <= 2e-5, float(err)
    # prompt variant: (B,1,H,W) masks, geometric-only prompt channel, same pipeline
    torch.manual_seed(7)
    augp = DataAugmentorPrompt(1).cuda()
    prompts = (torch.rand(10, 1, 64, 48) > 0.5).float()
    out, om, op = augp(img.cuda(), msk[:, None].cuda(), prompts.cuda())
    params, order = augp.last_params
    ref, rm, rp = A.augment(img, msk, prompts, params.cpu(), order.cpu().tolist())
    assert om.shape == msk.shape and op.shape == prompts.shape
    assert float((om.cpu() != rm).float().mean()) <= 2e-3 and float((op.cpu() != rp).float().mean()) <= 2e-3


def test_augment_contract_exact_and_statistical(M):
    """What the reference's DataAugmentor.forward guarantees (processing_blocks.py:372-384) -- every (aug+1)-th sample
    untouched, image and mask moved by the SAME geometry, label set preserved, output in [0,1] -- exactly; the
    kornia-defined sampling (flip p .5, rotation p .5 in +-90 deg, jitter ranges, sigma range) statistically."""
    import math

    from models.processing_blocks import DataAugmentor

    B, H, W = 200, 32, 32
    torch.manual_seed(0)
    # image channels carry the mask value, so geometric consistency is checkable after the colour ops are disabled
    msk = torch.randint(0, 3, (B, H, W))
    img = (msk[:, None].float() / 2.0).expand(B, 3, H, W).contiguous()
    aug = DataAugmentor(4).cuda()
    aug.brightness = aug.contrast = aug.saturation = aug.hue = 0.0
    aug.sigma = (1e-3, 1e-3)
    out, om = aug(img.cuda(), msk.cuda())
    out, om = out.cpu(), om.cpu()
    p = aug.last_params[0].cpu()
    assert torch.equal(p[:, 0] != 0, torch.arange(B) % 5 == 0)
    assert torch.equal(out[::5], img[::5]) and torch.equal(om[::5], msk[::5])
    inside = out[:, 0] * 2.0  # grey image == mask value wherever the source was inside the frame
    assert torch.equal(inside.round().long(), om), "image and mask moved by the same geometry"
    assert set(om.unique().tolist()) <= {0, 1, 2}
    changed = (om != msk).flatten(1).any(1)
    assert not changed[::5].any()
    flip, cos = p[:, 1] != 0, p[:, 2]
    rot = cos < 1.0
    moved = flip | rot
    keep = p[:, 0] != 0
    ang_all = torch.atan2(p[:, 3], p[:, 2]).abs()
    assert not (changed & ~moved).any()  # a changed mask implies a flip or a rotation was drawn ...
    assert changed[~keep & (flip | (ang_all > 0.1))].all()  # ... and a flip / a real rotation does change it
    assert 0.35 < float(flip.float().mean()) < 0.65 and 0.35 < float(rot.float().mean()) < 0.65
    ang = torch.atan2(p[:, 3], p[:, 2])[rot]
    assert float(ang.abs().max()) <= math.pi / 2 + 1e-6 and float(ang.min()) < -0.5 and float(ang.max()) > 0.5
    # default ranges
    torch.manual_seed(1)
    aug2 = DataAugmentor(0).cuda()  # aug = 0: every sample is "kept" (stride 1), as in the reference
    o2, m2 = aug2(img.cuda(), msk.cuda())
    assert torch.equal(o2.cpu(), img) and torch.equal(m2.cpu(), msk)
    aug3 = DataAugmentor(4).cuda()
    x = torch.rand(B, 3, H, W)
    o3, _ = aug3(x.cuda(), msk.cuda())
    q = aug3.last_params[0].cpu()
    assert float(o3.min()) >= 0.0 and float(o3.max()) <= 1.0 + 1e-6
    for col, lo, hi in ((4, 0.6, 1.4), (5, 0.7, 1.3), (6, 0.8, 1.2), (7, -0.2 * 2 * math.pi, 0.2 * 2 * math.pi), (8, 0.1, 2.0)):
        v = q[:, col]
        assert float(v.min()) >= lo - 1e-6 and float(v.max()) <= hi + 1e-6
        assert float(v.max() - v.min()) > 0.8 * (hi - lo), col
    # blur actually smooths: total variation drops on augmented samples
    tv = lambda z: float((z[..., 1:] - z[..., :-1]).abs().mean())
    nk = q[:, 0] == 0
    assert tv(o3.cpu()[nk]) < 0.8 * tv(x[nk])


def test_clip_unet_dead_bottleneck_keeps_reference_checkpoint_state(M, golden):
    """The reference's ClipUnet still RUNS its bottleneck ConvBlock although the fusion discards the result
    (models/CLIP_models.py:125-126), so a train-mode forward moves bottleneck.conv.{1,4}.running_* and
    num_batches_tracked.  The drop-in runs that block forward-only under no_grad (default) and must leave the same
    buffers and hand its parameters ZERO gradients (the reference's are rounding residue, not None, so weight decay
    acts on them); with the switch off buffers stay untouched and the gradients are None; live gradients are unaffected
    either way."""
    g = golden("models_r2")
    feats = T("clip.feats", (2, 512), -1.0, 1.0).cuda()

    class Fake(torch.nn.Module):
        def forward(self, x):
            return feats

    x = T("clip.x", (2, 3, 32, 32)).cuda()
    t = torch.from_numpy(fill.randint("clip.t", (2, 32, 32), 3)).cuda()
    outs = {}
    for flag in (True, False):
        m = M.cm.ClipUnet(clip_feature_extractor=Fake())
        fill.fill_state_dict(m.state_dict())
        m = m.cuda().train()
        m.run_dead_bottleneck = flag
        before = {k: v.clone() for k, v in m.state_dict().items() if k.startswith("bottleneck.")}
        with M.hipseg.precision_mode("fp32"):
            loss = M.ls.HybridLoss()(m(x), t)
            loss.backward()
        outs[flag] = (float(loss), m.out.weight.grad.clone())
        sd = m.state_dict()
        for k in before:
            if not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                continue
            if flag:
                np.testing.assert_allclose(sd[k].cpu().numpy(), g[f"clip_bn/{k}"], rtol=1e-4, atol=1e-5, err_msg=k)
            else:
                assert torch.equal(sd[k], before[k]), k
        # reference: a gradient tensor of rounding residue (so Adam's weight decay acts); here exact zeros / None
        gb = m.bottleneck.conv[0].weight.grad
        assert (gb is not None and float(gb.abs().sum()) == 0.0) if flag else gb is None
    assert outs[True][0] == outs[False][0] and torch.equal(outs[True][1], outs[False][1])


@pytest.mark.parametrize("which", ["torch", "hipseg"])
def test_clip_unet_dead_bottleneck_weight_decay_matches_reference_adam(M, golden, which):
    """three loop-body steps of Adam(lr=1e-3, weight_decay=1e-4) in fp32: the reference's dead bottleneck parameters
    shrink (their gradient is a tensor of ~0, so only the decay term drives Adam); the drop-in's must land on the same
    values -- with torch.optim.Adam and with hipseg.optim.Adam -- as must the live weights."""
    g = golden("models_r2")
    feats = T("clip.feats", (2, 512), -1.0, 1.0).cuda()

    class Fake(torch.nn.Module):
        def forward(self, x):
            return feats

    m = M.cm.ClipUnet(clip_feature_extractor=Fake())
    fill.fill_state_dict(m.state_dict())
    m = m.cuda().train()
    w0 = m.bottleneck.conv[0].weight.detach().clone()
    from hipseg.optim import Adam as HipAdam

    opt = (torch.optim.Adam if which == "torch" else HipAdam)(m.parameters(), lr=1e-3, weight_decay=1e-4)
    x = T("clip.x", (2, 3, 32, 32)).cuda()
    t = torch.from_numpy(fill.randint("clip.t", (2, 32, 32), 3)).cuda()
    with M.hipseg.precision_mode("fp32"):
        for _ in range(3):
            opt.zero_grad()
            M.ls.HybridLoss()(m(x), t).backward()
            opt.step()
    sd = m.state_dict()
    ref = g["clip_adam3/bottleneck.conv.0.weight[:4]"]
    got = sd["bottleneck.conv.0.weight"][:4].cpu().numpy()
    assert np.abs(w0[:4].cpu().numpy() - ref).mean() > 1e-3  # the reference really moved them (~3 lr)
    assert np.abs(got - ref).max() <= 2e-5, np.abs(got - ref).max()
    assert np.abs(sd["bottleneck.conv.4.weight"].cpu().numpy() - g["clip_adam3/bottleneck.conv.4.weight"]).max() <= 2e-5
    ref = g["clip_adam3/out.weight"]
    assert np.abs(sd["out.weight"].cpu().numpy() - ref).max() <= 2e-5
    # dec1.up.weight sits behind a spatially constant fusion output and 2-sample BatchNorm statistics: Adam's per-entry
    # normalisation turns rounding-level gradient differences into step differences.  The reference's own trajectory
    # is that sensitive -- the CPU oracle lands a median 4.8e-5 (fp32) / 3.5e-5 (fp64) away from it after these three
    # steps (total movement 3e-3) -- so only the bulk is held, loosely
    ref = g["clip_adam3/dec1.up.weight[:2]"]
    d = np.abs(sd["dec1.up.weight"][:2].cpu().numpy() - ref)
    assert np.median(d) <= 2e-4 and np.mean(d <= 1e-3) >= 0.95, (np.median(d), np.mean(d <= 1e-3))


def test_clip_autoencoder_fp32_vs_reference_golden(M, golden):
    """ClipAutoencoder (models/CLIP_models.py:136-188; same blocks, ConvBlockUpsample chain + one skip block whose
    256x256 up-sample is bilinearly resized down to the 32x32 skip): fp32 logits within 1e-4 of the reference."""
    g = golden("models_r2")
    feats = T("clip.feats", (2, 512), -1.0, 1.0).cuda()

    class Fake(torch.nn.Module):
        def forward(self, x):
            return feats

    m = M.cm.ClipAutoencoder(clip_feature_extractor=Fake())
    fill.fill_state_dict(m.state_dict())
    m = m.cuda()
    x = T("clipae.x", (2, 3, 32, 32)).cuda()
    t = torch.from_numpy(fill.randint("clipae.t", (2, 32, 32), 3)).cuda()
    with M.hipseg.precision_mode("fp32"):
        m.eval()
        with torch.no_grad():
            ev = m(x)
        m.train()
        logits = m(x)
        loss = M.ls.HybridLoss()(logits, t)
        loss.backward()
    assert np.abs(ev.cpu().numpy() - g["clipae/eval_logits"]).max() <= 1e-4
    assert np.abs(logits.detach().cpu().numpy() - g["clipae/train_logits"]).max() <= 1e-4
    assert abs(float(loss) - float(g["clipae/ce_loss"])) <= 1e-5
    params = dict(m.named_parameters())
    for k in ("coupler.weight", "dec1.up.weight", "dec4.conv.conv.3.weight", "out.weight", "input.weight"):
        gd = params[k].grad.double()
        np.testing.assert_allclose([float(gd.abs().sum()), float(gd.pow(2).sum())], g[f"clipae/gradstat/{k}"][1:], rtol=5e-3,
                                   atol=1e-6, err_msg=k)
    with torch.autocast("cuda"), torch.no_grad():
        tb = m(x).float().cpu().numpy()
    assert np.isfinite(tb).all()


def test_reference_loop_body_end_to_end_learns(M):
    """The reference's per-step loop body (models/model_wrappers.py:160-177) with every drop-in piece at once: on-device
    DataAugmentor -> autocast forward -> HybridLoss -> GradScaler -> hipseg.optim.Adam (the `optimizer_class` hook),
    on the learnable blob task.  The loss must fall and the validation IoU (IoU / PixelAccuracy / 2*IoU/(1+IoU), as
    model_wrappers.py:209-211) must rise well above chance."""
    from hipseg.optim import Adam
    from models.processing_blocks import DataAugmentor

    torch.manual_seed(0)
    model = M.un.UNet().cuda()
    opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    scaler = torch.amp.GradScaler("cuda")
    crit, iou, acc = M.ls.HybridLoss(), M.ls.IoU(), M.ls.PixelAccuracy()
    aug = DataAugmentor(4).cuda()
    # colour jitter off: the task's label IS the brightest channel, hue / saturation changes would relabel it
    aug.brightness = aug.contrast = aug.saturation = aug.hue = 0.0

    def batch(seed, B=10):
        g = torch.Generator().manual_seed(seed)
        low = torch.rand(B, 3, 8, 8, generator=g)
        x = torch.nn.functional.interpolate(low, size=(64, 64), mode="bilinear", align_corners=True)
        return x.cuda(), x.argmax(1).cuda()

    losses = []
    model.train()
    for step in range(60):
        x, t = batch(step)
        x, t = aug(x, t)  # (rotated-in corners: zero image, class 0 -- the augmentor's zero fill of both)
        opt.zero_grad()
        with torch.autocast("cuda"):
            loss = crit(model(x), t)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < 0.6 * np.mean(losses[:5]), (losses[:5], losses[-5:])
    assert opt.step_count() >= 55  # (GradScaler may skip a few early steps while it calibrates the scale)
    model.eval()
    xv, tv = batch(10_000)
    with torch.no_grad(), torch.autocast("cuda"):
        out = model(xv)
        i, a = float(iou(out, tv)), float(acc(out, tv))
    assert i > 0.5 and a > 0.6 and 2 * i / (1 + i) > 0.6, (i, a)

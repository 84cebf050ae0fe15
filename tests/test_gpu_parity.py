"""Block- and model-level parity of the HIP path (through the drop-in `models` package and the C ABI)
against the committed golden fixtures (generated from the reference) and the CPU oracle.

Bars (BASELINE.json north_star): fp32 logits within 1e-4 max-abs of the reference CPU forward;
bf16 masks within 1e-2 IoU.  Needs a real MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fill, torch_ref as R  # noqa: E402


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import hipseg
    from models import processing_blocks as pb, UNet as un, CLIP_models as cm, losses as ls

    class NS:
        pass

    ns = NS()
    ns.hipseg, ns.pb, ns.un, ns.cm, ns.ls = hipseg, pb, un, cm, ls
    return ns


def T(name, shape, lo=0.0, hi=1.0):
    return torch.from_numpy(fill.uniform(name, shape, lo, hi))


BLOCK_CASES = [
    ("cb_4_8", "ConvBlock", (4, 8), [(2, 4, 16, 16)]),
    ("cb_32_64", "ConvBlock", (32, 64), [(2, 32, 16, 16)]),
    ("cb_8_8_odd", "ConvBlock", (8, 8), [(1, 8, 8, 24)]),
    ("down_8_16", "ConvBlockDownsample", (8, 16), [(2, 8, 16, 16)]),
    ("upskip_16_8_identity", "ConvBlockUpsampleSkip", (16, 8), [(2, 16, 8, 8), (2, 8, 16, 16)]),
    ("upskip_16_8_dec1", "ConvBlockUpsampleSkip", (16, 8), [(2, 16, 8, 8), (2, 8, 8, 8)]),
    ("upskip_64_32_dec1", "ConvBlockUpsampleSkip", (64, 32), [(1, 64, 4, 12), (1, 32, 4, 12)]),
    ("up_16_8", "ConvBlockUpsample", (16, 8), [(2, 16, 8, 8)]),
]


def iou_masks(a, b, ncls=3):
    vals = []
    for c in range(ncls):
        pa, pb_ = a == c, b == c
        u = (pa | pb_).sum()
        if u:
            vals.append(float((pa & pb_).sum()) / float(u))
    return float(np.mean(vals))


def rel_l2(got, ref):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return float(np.sqrt(((got - ref) ** 2).sum()) / max(np.sqrt((ref ** 2).sum()), 1e-12))


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name,cls,args,shapes", BLOCK_CASES, ids=[c[0] for c in BLOCK_CASES])
def test_blocks_vs_reference_golden(M, golden, prec, name, cls, args, shapes):
    """fp32: element-wise 1e-4 (outputs) / 2e-3 of the tensor max (gradients).
    bf16: relative L2 error (ReLU / max-pool decisions flip under 8-bit mantissas, so single
    elements of a tiny-channel gradient may move a lot; the tensor as a whole may not)."""
    g = golden("blocks")
    m = getattr(M.pb, cls)(*args)
    fill.fill_state_dict(m.state_dict(), prefix=name + ".")
    m = m.cuda()
    ins = [T(f"{name}.in{i}", s, -1.0, 1.0).cuda().requires_grad_(True) for i, s in enumerate(shapes)]
    f32 = prec == "fp32"
    with M.hipseg.precision_mode(prec):
        m.eval()
        with torch.no_grad():
            ev = m(*ins)
        assert tuple(ev.shape) == g[f"{name}/eval_out"].shape
        m.train()
        y = m(*ins)
        gout = T(f"{name}.gout", tuple(y.shape), -1.0, 1.0).cuda()
        (y.float() * gout).sum().backward()
    torch.cuda.synchronize()
    evn, yn = ev.float().cpu().numpy(), y.detach().float().cpu().numpy()
    if f32:
        assert np.abs(evn - g[f"{name}/eval_out"]).max() <= 1e-4
        assert np.abs(yn - g[f"{name}/train_out"]).max() <= 1e-4
    else:
        assert rel_l2(evn, g[f"{name}/eval_out"]) <= 1.5e-2, rel_l2(evn, g[f"{name}/eval_out"])
        assert rel_l2(yn, g[f"{name}/train_out"]) <= 1.5e-2, rel_l2(yn, g[f"{name}/train_out"])
    for i, t in enumerate(ins):
        ref = g[f"{name}/grad_in{i}"]
        got = t.grad.float().cpu().numpy()
        if f32:
            assert np.abs(got - ref).max() <= 2e-3 * max(1.0, np.abs(ref).max()), f"grad_in{i}"
        else:
            assert rel_l2(got, ref) <= 0.12, f"grad_in{i}: rel L2 {rel_l2(got, ref)}"
    for k, p in m.named_parameters():
        ref = g[f"{name}/grad/{k}"]
        got = p.grad.cpu().numpy()
        if k.endswith(("conv.0.bias", "conv.3.bias")):
            # conv bias in front of train-mode BN: the true gradient is 0, both sides hold rounding noise
            assert np.abs(got).max() <= (1e-3 if f32 else 0.5), k
            continue
        if f32:
            assert np.abs(got - ref).max() <= 2e-3 * max(np.abs(ref).max(), 1e-3), f"grad {k}"
        else:
            bar = 0.25 if got.size <= 64 else 0.12  # tiny tensors: a single flipped ReLU moves them
            assert rel_l2(got, ref) <= bar, f"grad {k}: rel L2 {rel_l2(got, ref)}"
    for k, b in m.named_buffers():
        ref = g[f"{name}/buf/{k}"]
        np.testing.assert_allclose(b.cpu().numpy(), ref, rtol=1e-4 if f32 else 2e-2, atol=1e-5 if f32 else 2e-3)


MODEL_CASES = [
    ("unet_c1", "UNet", "c1", (2, 3, 128, 128)),
    ("large_64", "LargeUNet", "large", (1, 3, 64, 64)),
    ("unet_56x40", "UNet", "unet56", (1, 3, 56, 40)),
]


@pytest.mark.parametrize("tag,arch,key,shape", MODEL_CASES, ids=[c[0] for c in MODEL_CASES])
def test_models_fp32_vs_reference_golden(M, golden, tag, arch, key, shape):
    """fp32 path: logits within 1e-4 max-abs of the reference's PyTorch-CPU forward."""
    g = golden("models")
    m = getattr(M.un, arch)()
    fill.fill_state_dict(m.state_dict())
    m = m.cuda()
    x = T(f"{key}.x", shape).cuda()
    t = torch.from_numpy(fill.randint(f"{key}.t", (shape[0],) + shape[2:], 3)).cuda()
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
    assert ev.dtype == torch.float32 and tuple(ev.shape) == (shape[0], 3) + shape[2:]
    e1 = np.abs(ev.cpu().numpy() - g[f"{tag}/eval_logits"]).max()
    e2 = np.abs(logits.detach().cpu().numpy() - g[f"{tag}/train_logits"]).max()
    assert e1 <= 1e-4, f"eval logits max-abs {e1}"
    assert e2 <= 1e-4, f"train logits max-abs {e2}"
    assert abs(float(loss) - float(g[f"{tag}/ce_loss"])) <= 1e-5
    sd = m.state_dict()
    for k, p in m.named_parameters():
        s = g[f"{tag}/gradstat/{k}"]
        gd = p.grad.double()
        if k.endswith(("conv.0.bias", "conv.3.bias")):
            # conv bias in front of train-mode BN: exactly 0 here, ~1e-6 of rounding noise in the reference
            assert float(gd.abs().sum()) <= 1e-4 and s[1] <= 1e-4, k
            continue
        mine = np.array([float(gd.abs().sum()), float(gd.pow(2).sum())])
        np.testing.assert_allclose(mine, s[1:], rtol=5e-3, atol=1e-6, err_msg=k)
        gk = f"{tag}/grad/{k}"
        if gk in g:
            ref = g[gk]
            # deep gradients accumulate fp32 rounding through ~40 kernels and a few ReLU/pool ties
            assert np.abs(p.grad.cpu().numpy() - ref).max() <= 1e-2 * max(np.abs(ref).max(), 1e-4), k
    for k in ("enc1.block.0.conv.1.running_mean", "enc1.block.0.conv.1.running_var",
              "bottleneck.conv.4.running_mean", "bottleneck.conv.4.running_var",
              "bottleneck.conv.4.num_batches_tracked"):
        np.testing.assert_allclose(sd[k].cpu().numpy(), g[f"{tag}/buf/{k}"], rtol=1e-4, atol=1e-5, err_msg=k)


def test_bf16_vs_reference_bf16_yardstick(M, golden):
    """bf16 path (under torch.autocast, as the reference's TrainingWrapper runs) on the synthetic-fill
    UNet.  These untrained weights give near-tied class logits (median top-2 margin 0.18), so NO bf16
    implementation reaches mask IoU 0.99 here: the reference's own CPU bf16-autocast forward (fixture
    bf16ref_*) scores 0.983 / 0.936 (eval / train).  Bar: at least as accurate as that yardstick.
    The 1e-2 IoU bar itself is checked on trained weights in test_bf16_mask_iou_trained."""
    g = golden("models")
    m = M.un.UNet()
    fill.fill_state_dict(m.state_dict())
    m = m.cuda()
    x = T("c1.x", (2, 3, 128, 128)).cuda()
    with torch.autocast("cuda"):
        assert M.hipseg.precision() == "bf16"
        m.eval()
        with torch.no_grad():
            ev = m(x)
        m.train()
        with torch.no_grad():
            tr = m(x)
    for mode, got in (("eval", ev), ("train", tr)):
        ref = g[f"unet_c1/{mode}_logits"]
        yard = g[f"unet_c1/bf16ref_{mode}_logits"]
        got = got.float().cpu().numpy()
        rms = float(np.sqrt(((got - ref) ** 2).mean()))
        rms_yard = float(np.sqrt(((yard - ref) ** 2).mean()))
        iou = iou_masks(got.argmax(1), ref.argmax(1))
        iou_yard = iou_masks(yard.argmax(1), ref.argmax(1))
        assert rms <= 1.1 * rms_yard, (mode, rms, rms_yard)
        assert iou >= iou_yard - 5e-3, (mode, iou, iou_yard)


def test_bf16_mask_iou_trained(M):
    """The north-star bf16 bar -- masks within 1e-2 IoU of the fp32 CPU forward -- on TRAINED weights.
    A UNet is trained for a few dozen steps (bf16 autocast + GradScaler + Adam, the reference's loop
    body, model_wrappers.py:167-177) on a learnable synthetic task (class = dominant colour channel of
    the blurred image), which yields the confident logits a real checkpoint has; then the bf16 HIP
    eval forward is compared with the CPU oracle (fp32) on the same trained state_dict."""
    torch.manual_seed(0)
    m = M.un.UNet().cuda().train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)
    scaler = torch.amp.GradScaler("cuda")
    crit = M.ls.HybridLoss()

    def batch(seed, B=8):
        gen = torch.Generator().manual_seed(seed)
        img = torch.rand(B, 3, 64, 64, generator=gen)
        img = torch.nn.functional.avg_pool2d(img, 9, 1, 4)  # blur -> spatially coherent regions
        img = (img - img.amin((1, 2, 3), keepdim=True)) / (img.amax((1, 2, 3), keepdim=True) - img.amin((1, 2, 3), keepdim=True))
        return img, img.argmax(1)

    first = last = None
    for step in range(80):
        x, t = batch(step)
        opt.zero_grad()
        with torch.autocast("cuda"):
            loss = crit(m(x.cuda()), t.cuda())
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        last = float(loss)
        first = last if first is None else first
    assert last < 0.5 * first, (first, last)  # it learns
    x, t = batch(10_000, B=4)
    m.eval()
    with torch.autocast("cuda"), torch.no_grad():
        got = m(x.cuda()).float().cpu().numpy()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref = R.unet_forward(x, sd, "UNet", train=False).numpy()
    iou = iou_masks(got.argmax(1), ref.argmax(1))
    assert iou >= 1.0 - 1e-2, f"bf16 vs fp32-oracle mask IoU {iou}"
    # and the fp32 HIP path reproduces the oracle logits to 1e-4 on the trained weights
    with M.hipseg.precision_mode("fp32"), torch.no_grad():
        got32 = m(x.cuda()).cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(got32 - ref).max() <= 1e-4 * scale, np.abs(got32 - ref).max()


def test_adam_trajectory_fp32(M, golden):
    """5 optimiser steps of the reference's loop body (model_wrappers.py:167-177, fp32) reproduce its losses."""
    g = golden("models")
    m = M.un.UNet()
    fill.fill_state_dict(m.state_dict())
    m = m.cuda().train()
    x = T("c1.x", (2, 3, 128, 128)).cuda()
    t = torch.from_numpy(fill.randint("c1.t", (2, 128, 128), 3)).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)
    crit = M.ls.HybridLoss()
    traj = []
    with M.hipseg.precision_mode("fp32"):
        for _ in range(5):
            opt.zero_grad()
            loss = crit(m(x), t)
            loss.backward()
            opt.step()
            traj.append(float(loss))
    np.testing.assert_allclose(traj, g["unet_c1/adam_traj"], rtol=2e-3)


def test_clip_unet_fp32(M, golden):
    g = golden("clip")
    feats = T("clip.feats", (2, 512), -1.0, 1.0).cuda()

    class Fake(torch.nn.Module):
        def forward(self, x):
            return feats

    m = M.cm.ClipUnet(clip_feature_extractor=Fake())
    fill.fill_state_dict(m.state_dict())
    m = m.cuda()
    x = T("clip.x", (2, 3, 32, 32)).cuda()
    t = torch.from_numpy(fill.randint("clip.t", (2, 32, 32), 3)).cuda()
    with M.hipseg.precision_mode("fp32"):
        m.eval()
        with torch.no_grad():
            ev = m(x)
        m.train()
        logits = m(x)
        loss = M.ls.HybridLoss()(logits, t)
        loss.backward()
    assert np.abs(ev.cpu().numpy() - g["clip/eval_logits"]).max() <= 1e-4
    assert np.abs(logits.detach().cpu().numpy() - g["clip/train_logits"]).max() <= 1e-4
    assert abs(float(loss) - float(g["clip/ce_loss"])) <= 1e-5
    gb = m.cross_attention_fusion.cross_attn.out_proj.bias.grad
    np.testing.assert_allclose(gb.cpu().numpy(), g["clip/grad/out_proj.bias"], rtol=2e-3, atol=1e-6)
    for k, p in m.named_parameters():
        if k.startswith("bottleneck.") or p.grad is None:
            continue  # dead branch: exactly zero gradient here, ~1e-7 rounding noise in the reference
        s = g[f"clip/gradstat/{k}"]
        if "in_proj" in k:
            continue  # q/k rows: zero here, rounding noise in the reference; v rows checked via out_proj
        np.testing.assert_allclose(float(p.grad.double().abs().sum()), s[1], rtol=5e-3, atol=1e-6, err_msg=k)


def test_metrics_vs_reference_golden(M, golden):
    g = golden("losses")
    logits = T("loss.logits", (2, 3, 32, 32), -3.0, 3.0).cuda()
    tgt = torch.from_numpy(fill.randint("loss.t", (2, 32, 32), 3)).cuda()
    assert abs(float(M.ls.IoU()(logits, tgt)) - float(g["iou"])) < 1e-6
    assert abs(float(M.ls.PixelAccuracy()(logits, tgt)) - float(g["pixel_accuracy"])) < 1e-6
    tgt2 = torch.from_numpy(fill.randint("loss.t2", (2, 32, 32), 2)).cuda()
    assert abs(float(M.ls.IoU()(logits, tgt2)) - float(g["iou_2cls"])) < 1e-6
    assert abs(float(M.ls.PixelAccuracy()(logits, tgt2)) - float(g["pixel_accuracy_2cls"])) < 1e-6
    bl = T("loss.blogits", (2, 1, 32, 32), -3.0, 3.0).cuda()
    bt = torch.from_numpy(fill.randint("loss.bt", (2, 32, 32), 2)).float().cuda()
    assert abs(float(M.ls.IoUBinary()(bl, bt)) - float(g["iou_binary"])) < 1e-6
    assert abs(float(M.ls.PixelAccuracyBinary()(bl, bt)) - float(g["pixel_accuracy_binary"])) < 1e-6


def test_full_size_properties_c2(M):
    """BASELINE config 2 geometry (UNet, 16x3x256x256): size-independent properties (the CPU oracle
    is too slow at this size): determinism, linearity of the gradient in the upstream scale
    (GradScaler contract), bf16 vs the fp32 HIP path, BN bookkeeping."""
    m = M.un.UNet()
    fill.fill_state_dict(m.state_dict())
    m = m.cuda().train()
    torch.manual_seed(0)
    x = torch.rand(16, 3, 256, 256, device="cuda")
    t = torch.randint(0, 3, (16, 256, 256), device="cuda")
    crit = M.ls.HybridLoss()

    def grads(scale, prec):
        m.zero_grad(set_to_none=True)
        with M.hipseg.precision_mode(prec):
            out = m(x)
            loss = crit(out, t)
        (loss * scale).backward()
        return out.detach(), float(loss), {k: p.grad.clone() for k, p in m.named_parameters()}

    o1, l1, g1 = grads(1.0, "bf16")
    o2, l2, g2 = grads(1024.0, "bf16")
    assert np.isfinite(l1) and abs(l1 - l2) < 1e-6 * max(1.0, abs(l1))
    assert torch.equal(o1, o2), "forward is deterministic"
    for k in g1:
        if k.endswith(("conv.0.bias", "conv.3.bias")):
            continue  # conv bias before train-mode BN: mathematically zero, rounding noise
        a, b = g1[k].double() * 1024.0, g2[k].double()
        assert float((a - b).norm() / (b.norm() + 1e-30)) < 2e-2, k
    o32, l32, g32 = grads(1.0, "fp32")
    rel = float((o1.double() - o32.double()).norm() / o32.double().norm())
    assert rel < 0.1, rel
    assert abs(l1 - l32) < 2e-2
    # bf16 vs fp32 gradients.  With random targets the deep gradients nearly cancel, so 8-bit mantissas
    # perturb them strongly in ANY implementation: the reference's own CPU bf16-autocast backward differs
    # from its fp32 backward by 0.004 (out.weight), 0.05 (dec4), 0.5 (bottleneck and deeper) relative L2
    # on the C1 fixture (measured in the build container).  Bars sit just above that yardstick.
    for k, bar in (("out.weight", 0.02), ("dec4.conv.conv.3.weight", 0.12), ("bottleneck.conv.0.weight", 0.75),
                   ("enc1.block.0.conv.0.weight", 0.75)):
        r = float((g1[k].double() - g32[k].double()).norm() / g32[k].double().norm())
        assert r < bar, (k, r)
    assert int(m.bottleneck.conv[1].num_batches_tracked) == 3
    # run-to-run determinism down to the bit: the split-K weight gradients are summed in a fixed order
    # (no atomics), BN statistics in fixed row order
    o1b, l1b, g1b = grads(1.0, "bf16")
    assert l1b == l1 and torch.equal(o1, o1b)
    for k in g1:
        assert torch.equal(g1[k], g1b[k]), f"non-deterministic gradient {k}"
    # eval mode is per-sample independent: one batch of 16 == two batches of 8, bit for bit (the persistent
    # weights-stationary kernel assigns tiles to workgroups differently for the two grids)
    m.eval()
    with torch.no_grad(), M.hipseg.precision_mode("bf16"):
        full = m(x)
        halves = torch.cat([m(x[:8]), m(x[8:])], 0)
    assert torch.equal(full, halves)


def test_torch_compile_wrapper_is_transparent(M):
    """The reference's TrainingWrapper calls torch.compile(model) (models/model_wrappers.py:118).  Our forwards are
    torch.compiler.disable'd: the compiled wrapper must run the same HIP path (bit-identical results), expose the
    `_orig_mod.` state_dict keys the reference's checkpoints carry, and train."""
    m = M.un.UNet()
    fill.fill_state_dict(m.state_dict())
    m = m.cuda().train()
    x = T("c1.x", (2, 3, 128, 128)).cuda()
    t = torch.from_numpy(fill.randint("c1.t", (2, 128, 128), 3)).cuda()
    crit = M.ls.HybridLoss()

    def step(net):
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda"):
            out = net(x)
            loss = crit(out, t)
        loss.backward()
        return out.detach().clone(), float(loss.detach()), m.out.weight.grad.clone(), m.enc1.block[0].conv[0].weight.grad.clone()

    o0, l0, g0, h0 = step(m)
    cm = torch.compile(m)
    o1, l1, g1, h1 = step(cm)
    assert torch.equal(o0, o1) and l0 == l1 and torch.equal(g0, g1) and torch.equal(h0, h1)
    keys = list(cm.state_dict().keys())
    assert len(keys) == 124 and all(k.startswith("_orig_mod.") for k in keys)
    import hipseg.ckpt as ck
    fresh = M.un.UNet()
    ck.load_reference_checkpoint(fresh, {k: v.cpu() for k, v in cm.state_dict().items()})
    assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(fresh.state_dict().values(), m.state_dict().values()))

"""Pin the CPU oracle (oracle/torch_ref.py) against fixtures generated from the
reference's own modules (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import fill, torch_ref as R

torch.set_num_threads(1)

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


def T(name, shape, lo=0.0, hi=1.0):
    return torch.from_numpy(fill.uniform(name, shape, lo, hi))


def block_state(cls, args):
    """state mapping with the reference's key names for one block."""
    from collections import OrderedDict

    sd = OrderedDict()
    ci, co = args
    if cls == "ConvBlock":
        R._conv_block_entries(sd, "conv.", ci, co)
        fn = lambda ins, sd, tr: R.conv_block(ins[0], sd, "conv.", tr)
    elif cls == "ConvBlockDownsample":
        R._conv_block_entries(sd, "block.0.conv.", ci, co)
        fn = lambda ins, sd, tr: R.down(ins[0], sd, "", tr)
    else:
        sd["up.weight"] = torch.zeros(ci, co, 2, 2)
        sd["up.bias"] = torch.zeros(co)
        if cls == "ConvBlockUpsampleSkip":
            R._conv_block_entries(sd, "conv.conv.", 2 * co, co)
            fn = lambda ins, sd, tr: R.up_skip(ins[0], ins[1], sd, "", tr)
        else:
            R._conv_block_entries(sd, "conv.conv.", co, co)
            fn = lambda ins, sd, tr: R.up(ins[0], sd, "", tr)
    return sd, fn


@pytest.mark.parametrize("name,cls,args,shapes", BLOCK_CASES, ids=[c[0] for c in BLOCK_CASES])
def test_blocks(golden, name, cls, args, shapes):
    g = golden("blocks")
    sd, fn = block_state(cls, args)
    fill.fill_state_dict(sd, prefix=name + ".")
    for k, v in sd.items():
        if R.is_param(k):
            v.requires_grad_(True)
    ins = [T(f"{name}.in{i}", s, -1.0, 1.0).requires_grad_(True) for i, s in enumerate(shapes)]
    with torch.no_grad():
        ev = fn(ins, sd, False)
    np.testing.assert_allclose(ev.numpy(), g[f"{name}/eval_out"], rtol=0, atol=1e-5)
    y = fn(ins, sd, True)
    (y * T(f"{name}.gout", tuple(y.shape), -1.0, 1.0)).sum().backward()
    np.testing.assert_allclose(y.detach().numpy(), g[f"{name}/train_out"], rtol=0, atol=1e-5)
    for i, t in enumerate(ins):
        np.testing.assert_allclose(t.grad.numpy(), g[f"{name}/grad_in{i}"], rtol=1e-4, atol=1e-5)
    for k, v in sd.items():
        if R.is_param(k):
            ref = g[f"{name}/grad/{k}"]
            np.testing.assert_allclose(v.grad.numpy(), ref, rtol=1e-4, atol=2e-5 * max(1.0, np.abs(ref).max()))
        else:
            np.testing.assert_allclose(v.detach().numpy(), g[f"{name}/buf/{k}"], rtol=1e-5, atol=1e-6)


MODEL_CASES = [
    ("unet_c1", "UNet", "c1", (2, 3, 128, 128)),
    ("large_64", "LargeUNet", "large", (1, 3, 64, 64)),
    ("unet_56x40", "UNet", "unet56", (1, 3, 56, 40)),
]


@pytest.mark.parametrize("tag,arch,key,shape", MODEL_CASES, ids=[c[0] for c in MODEL_CASES])
def test_models(golden, tag, arch, key, shape):
    g = golden("models")
    x = T(f"{key}.x", shape)
    t = torch.from_numpy(fill.randint(f"{key}.t", (shape[0],) + shape[2:], 3))
    sd = fill.fill_state_dict(R.make_state(arch))
    with torch.no_grad():
        ev = R.unet_forward(x, sd, arch, train=False)
    assert np.abs(ev.numpy() - g[f"{tag}/eval_logits"]).max() <= 1e-4
    for k, v in sd.items():
        if R.is_param(k):
            v.requires_grad_(True)
    logits = R.unet_forward(x, sd, arch, train=True)
    loss = R.hybrid_loss(logits, t)
    loss.backward()
    assert np.abs(logits.detach().numpy() - g[f"{tag}/train_logits"]).max() <= 1e-4
    assert abs(float(loss) - float(g[f"{tag}/ce_loss"])) <= 1e-5
    for k, v in sd.items():
        if R.is_param(k):
            s = g[f"{tag}/gradstat/{k}"]
            mine = np.array([float(v.grad.double().sum()), float(v.grad.double().abs().sum()),
                             float(v.grad.double().pow(2).sum())])
            np.testing.assert_allclose(mine[1:], s[1:], rtol=2e-3, atol=1e-7)
            gk = f"{tag}/grad/{k}"
            if gk in g:
                np.testing.assert_allclose(v.grad.numpy(), g[gk], rtol=1e-3, atol=1e-4 * max(1e-3, np.abs(g[gk]).max()))
    for k in ("enc1.block.0.conv.1.running_mean", "bottleneck.conv.4.running_var",
              "bottleneck.conv.4.num_batches_tracked"):
        np.testing.assert_allclose(sd[k].detach().numpy(), g[f"{tag}/buf/{k}"], rtol=1e-5, atol=1e-6)


def test_adam_trajectory(golden):
    g = golden("models")
    tr = R.OracleTrainer("UNet")
    x = T("c1.x", (2, 3, 128, 128))
    t = torch.from_numpy(fill.randint("c1.t", (2, 128, 128), 3))
    traj = [tr.step(x, t) for _ in range(5)]
    np.testing.assert_allclose(traj, g["unet_c1/adam_traj"], rtol=1e-3)


def test_state_layout():
    sd = R.make_state("UNet")
    assert len(sd) == 124
    assert sum(v.numel() for k, v in sd.items() if R.is_param(k)) == 7_755_907
    sd = R.make_state("LargeUNet")
    assert sum(v.numel() for k, v in sd.items() if R.is_param(k)) == 31_096_451


def test_clip_unet(golden):
    g = golden("clip")
    feats = T("clip.feats", (2, 512), -1.0, 1.0)
    x = T("clip.x", (2, 3, 32, 32))
    t = torch.from_numpy(fill.randint("clip.t", (2, 32, 32), 3))
    sd = fill.fill_state_dict(R.make_state("ClipUnet"))
    with torch.no_grad():
        ev = R.unet_forward(x, sd, "ClipUnet", False, feats)
        ev_c = R.unet_forward(x, sd, "ClipUnet", False, feats, collapsed=True)
    assert np.abs(ev.numpy() - g["clip/eval_logits"]).max() <= 1e-4
    assert np.abs(ev_c.numpy() - g["clip/eval_logits"]).max() <= 1e-4  # degenerate attention == affine map
    for collapsed in (False, True):
        sd = fill.fill_state_dict(R.make_state("ClipUnet"))
        for k, v in sd.items():
            if R.is_param(k):
                v.requires_grad_(True)
        logits = R.unet_forward(x, sd, "ClipUnet", True, feats, collapsed=collapsed)
        loss = R.hybrid_loss(logits, t)
        loss.backward()
        assert np.abs(logits.detach().numpy() - g["clip/train_logits"]).max() <= 1e-4
        assert abs(float(loss) - float(g["clip/ce_loss"])) <= 1e-5
        gb = sd["cross_attention_fusion.cross_attn.out_proj.bias"].grad
        np.testing.assert_allclose(gb.numpy(), g["clip/grad/out_proj.bias"], rtol=1e-3, atol=1e-6)
    # fusion alone, full attention vs collapsed vs reference
    sd = fill.fill_state_dict(R.make_state("ClipUnet"))
    b = T("caf.bott", (2, 512, 4, 4), -1.0, 1.0)
    with torch.no_grad():
        full = R.cross_attention_fusion(b, feats, sd)
        coll = R.cross_attention_fusion(b, feats, sd, collapsed=True)
    assert np.abs(full.numpy() - g["caf/out"]).max() <= 1e-5
    assert np.abs(coll.numpy() - g["caf/out"]).max() <= 1e-5


def test_losses_metrics(golden):
    g = golden("losses")
    logits = T("loss.logits", (2, 3, 32, 32), -3.0, 3.0)
    tgt = torch.from_numpy(fill.randint("loss.t", (2, 32, 32), 3))
    lg = logits.clone().requires_grad_(True)
    ce = R.hybrid_loss(lg, tgt)
    ce.backward()
    assert abs(float(ce) - float(g["ce"])) < 1e-6
    np.testing.assert_allclose(lg.grad.numpy(), g["ce_grad"], rtol=1e-5, atol=1e-9)
    assert abs(float(R.iou(logits, tgt)) - float(g["iou"])) < 1e-6
    assert abs(float(R.pixel_accuracy(logits, tgt)) - float(g["pixel_accuracy"])) < 1e-6
    tgt2 = torch.from_numpy(fill.randint("loss.t2", (2, 32, 32), 2))
    assert abs(float(R.iou(logits, tgt2)) - float(g["iou_2cls"])) < 1e-6
    assert abs(float(R.pixel_accuracy(logits, tgt2)) - float(g["pixel_accuracy_2cls"])) < 1e-6
    bl = T("loss.blogits", (2, 1, 32, 32), -3.0, 3.0)
    bt = torch.from_numpy(fill.randint("loss.bt", (2, 32, 32), 2)).float()
    bce = torch.nn.functional.binary_cross_entropy_with_logits(bl, bt.unsqueeze(1))
    assert abs(float(bce) - float(g["bce"])) < 1e-6
    assert abs(float(R.iou_binary(bl, bt)) - float(g["iou_binary"])) < 1e-6
    assert abs(float(R.pixel_accuracy_binary(bl, bt)) - float(g["pixel_accuracy_binary"])) < 1e-6


def test_dice_binary_known_answers():
    """smp DiceLoss(binary) is third-party and absent: PARITY UNPINNED, hand-derived KATs only."""
    # all-ones target, constant probability p (from_logits=False): dice = 2p/(p+1)
    p = torch.full((2, 1, 4, 4), 0.25)
    t = torch.ones(2, 1, 4, 4)
    assert abs(float(R.dice_loss_binary_smp(p, t, from_logits=False)) - (1 - 2 * 0.25 / 1.25)) < 1e-6
    # empty target -> loss masked to zero
    assert float(R.dice_loss_binary_smp(p, torch.zeros(2, 1, 4, 4), from_logits=False)) == 0.0
    # from_logits=True applies a sigmoid: logits 0 -> p = 0.5 -> dice = 2*0.5/1.5
    z = torch.zeros(2, 1, 4, 4)
    assert abs(float(R.dice_loss_binary_smp(z, t)) - (1 - 1.0 / 1.5)) < 1e-6
    # HybridLossBinary on zero logits, all-ones target: BCE = ln2; dice arg = sigmoid(sigmoid(0)) = sigmoid(.5)
    s = 1 / (1 + np.exp(-0.5))
    want = np.log(2.0) + (1 - 2 * s / (s + 1))
    assert abs(float(R.hybrid_loss_binary(z, t[:, 0])) - want) < 1e-6

"""Round-3 parity cases (need a real MI355X):
  * the BINARY model step of the reference's scripts/prompt_train.py:58 -- `UNet(out_channels=1)` (1-channel head) +
    `models.losses.HybridLossBinary` with a (B,H,W) float target (the unsqueeze path of models/losses.py:30-31) +
    `DataAugmentorPrompt` in front -- as ONE train step on the HIP path.  fp32 logits and the BCE half are pinned by
    tests/golden/models_r3.npz (reference UNet(out_channels=1) + nn.BCEWithLogitsLoss); the Dice half restates
    segmentation_models_pytorch 0.4.0 (absent): PARITY UNPINNED, checked against oracle.torch_ref only.
  * `ClipUnetPrompt` (models/prompt_segmentation.py:32-95), the model that script trains: fp32 logits / BCE / gradients
    against tests/golden/prompt_r3.npz (the reference's own ClipUnetPrompt with an injected CLIP vector), then bf16
    training steps with HybridLossBinary."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fill, torch_ref as R  # noqa: E402

from test_gpu_parity import M, T  # noqa: E402,F401  (fixture + helper)


def _binary_case():
    x = T("bin.x", (2, 3, 64, 64))
    t = torch.from_numpy((fill.uniform("bin.t", (2, 64, 64), 0.0, 1.0) > 0.5).astype(np.float32))
    return x, t


def test_unet_1ch_fp32_logits_and_bce_gradients_vs_reference_golden(M, golden):
    g = golden("models_r3")
    x, t = _binary_case()
    m = M.un.UNet(out_channels=1)
    fill.fill_state_dict(m.state_dict())
    m = m.cuda()
    xd, td = x.cuda(), t.cuda()
    with M.hipseg.precision_mode("fp32"):
        m.eval()
        with torch.no_grad():
            ev = m(xd)
        m.train()
        logits = m(xd)
        assert logits.shape == (2, 1, 64, 64) and logits.dtype == torch.float32
        # the pinnable half: the reference's nn.BCEWithLogitsLoss value and its gradients through the 1-channel head
        bce = torch.nn.functional.binary_cross_entropy_with_logits(logits, td.unsqueeze(1))
        bce.backward()
    torch.cuda.synchronize()
    assert np.abs(ev.cpu().numpy() - g["unet_bin/eval_logits"]).max() <= 1e-4
    assert np.abs(logits.detach().cpu().numpy() - g["unet_bin/train_logits"]).max() <= 1e-4
    assert abs(float(bce.detach()) - float(g["unet_bin/bce_loss"])) <= 1e-5
    n = 0
    for k, p in m.named_parameters():
        if k.endswith(("conv.0.bias", "conv.3.bias")):  # conv bias before train-mode BN: true gradient 0 (+ noise)
            continue
        s = g[f"unet_bin/gradstat/{k}"]
        gd = p.grad.double()
        np.testing.assert_allclose([float(gd.abs().sum()), float(gd.pow(2).sum())], s[1:], rtol=5e-3, atol=1e-7, err_msg=k)
        gk = f"unet_bin/grad/{k}"
        if gk in g:
            assert np.abs(p.grad.cpu().numpy() - g[gk]).max() <= 1e-2 * max(np.abs(g[gk]).max(), 1e-5), k
        n += 1
    assert n >= 30


def test_hybrid_loss_binary_module_train_step(M):
    """models.losses.HybridLossBinary (module, not the Function) on the 1-channel model's logits with a (B,H,W) float
    target: value and d(loss)/d(logits) against oracle.torch_ref.hybrid_loss_binary on the SAME logits (Dice: parity
    unpinned), the whole fp32 step's parameter gradients against backpropagating the oracle's logit gradient, then a
    bf16 (autocast) step."""
    x, t = _binary_case()
    m = M.un.UNet(out_channels=1)
    fill.fill_state_dict(m.state_dict())
    m = m.cuda().train()
    crit = M.ls.HybridLossBinary()
    xd, td = x.cuda(), t.cuda()
    with M.hipseg.precision_mode("fp32"):
        logits = m(xd)
        logits.retain_grad()
        loss = crit(logits, td)  # (B,H,W) target: unsqueezed inside, as models/losses.py:30-31
        loss.backward()
    torch.cuda.synchronize()
    lc = logits.detach().cpu().clone().requires_grad_(True)
    ref = R.hybrid_loss_binary(lc, t)
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 1e-5
    np.testing.assert_allclose(logits.grad.cpu().numpy(), lc.grad.numpy(), rtol=1e-4, atol=1e-9)
    # a (B,1,H,W) target gives the same value
    with M.hipseg.precision_mode("fp32"), torch.no_grad():
        assert float(crit(logits.detach(), td.unsqueeze(1))) == float(loss.detach())
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    # the same parameter gradients come out of pushing the ORACLE's logit gradient through the model
    m.zero_grad(set_to_none=True)
    fill.fill_state_dict(m.state_dict())  # (running statistics back: identical forward)
    with M.hipseg.precision_mode("fp32"):
        m(xd).backward(lc.grad.cuda())
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        assert torch.allclose(p.grad, grads[k], rtol=1e-3, atol=1e-7 + 1e-3 * float(grads[k].abs().max())), k
    # bf16 production path: finite, close to the fp32 value, every parameter gets a finite gradient
    m.zero_grad(set_to_none=True)
    with torch.autocast("cuda"):
        lb = m(xd)
        loss_b = crit(lb, td)
    loss_b.backward()
    torch.cuda.synchronize()
    assert lb.shape == (2, 1, 64, 64)
    assert np.isfinite(float(loss_b.detach())) and abs(float(loss_b.detach()) - float(loss.detach())) <= 5e-2
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())


def test_prompt_pipeline_augmentor_model_loss_optimizer(M):
    """DataAugmentorPrompt -> UNet(out_channels=1) -> HybridLossBinary -> GradScaler/Adam: the loop body of
    scripts/prompt_train.py (augment at :95-ish, criterion :58) with the trunk standing in for ClipUnetPrompt."""
    from hipseg.optim import Adam

    torch.manual_seed(0)
    B, S = 5, 64
    imgs = torch.rand(B, 3, S, S, device="cuda")
    masks = (torch.rand(B, S, S, device="cuda") > 0.5).long()
    prompts = torch.rand(B, 1, S, S, device="cuda")
    aug = M.pb.DataAugmentorPrompt(4)
    xi, mi, pi = aug(imgs, masks, prompts)
    assert xi.shape == imgs.shape and mi.shape == masks.shape and pi.shape == prompts.shape
    assert set(mi.unique().tolist()) <= {0, 1} and mi.dtype == torch.long
    m = M.un.UNet(out_channels=1).cuda().train()
    opt = Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)
    scaler = torch.amp.GradScaler("cuda")
    crit = M.ls.HybridLossBinary()
    losses = []
    for _ in range(4):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda"):
            out = m(xi)
            loss = crit(out, mi.float())
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def _prompt_model(M):
    from models import prompt_segmentation as ps

    feats = T("prompt.feats", (2, 512), -1.0, 1.0).cuda()

    class Fake(torch.nn.Module):
        def forward(self, x):
            return feats

    m = ps.ClipUnetPrompt(clip_feature_extractor=Fake())
    fill.fill_state_dict(m.state_dict())
    x = T("prompt.x", (2, 3, 64, 64)).cuda()
    heat = T("prompt.heat", (2, 1, 64, 64)).cuda()
    t = torch.from_numpy((fill.uniform("prompt.t", (2, 64, 64), 0.0, 1.0) > 0.5).astype(np.float32)).cuda()
    return m.cuda(), x, heat, t


def test_clip_unet_prompt_fp32_vs_reference_golden(M, golden):
    g = golden("prompt_r3")
    m, x, heat, t = _prompt_model(M)
    with M.hipseg.precision_mode("fp32"):
        m.eval()
        with torch.no_grad():
            ev = m(x, heat)
            pe = m.prompt_encoder(heat)
        m.train()
        logits = m(x, heat)
        assert logits.shape == (2, 1, 64, 64) and logits.dtype == torch.float32
        bce = torch.nn.functional.binary_cross_entropy_with_logits(logits, t.unsqueeze(1))
        bce.backward()
    torch.cuda.synchronize()
    assert np.abs(ev.cpu().numpy() - g["prompt/eval_logits"]).max() <= 1e-4
    np.testing.assert_allclose([float(pe.double().sum()), float(pe.double().abs().sum())],
                               g["prompt/eval_prompt_embedding_stat"], rtol=1e-4)
    assert np.abs(logits.detach().cpu().numpy() - g["prompt/train_logits"]).max() <= 1e-4
    assert abs(float(bce.detach()) - float(g["prompt/bce_loss"])) <= 1e-5
    n = 0
    for k, p in m.named_parameters():
        if k.startswith("bottleneck.") or "in_proj" in k or p.grad is None:
            continue  # dead branch / q,k rows: exactly zero here, rounding noise in the reference (as for ClipUnet)
        if k.endswith(("conv.0.bias", "conv.3.bias")):  # conv bias before train-mode BN: true gradient 0 (+ noise)
            continue
        s = g[f"prompt/gradstat/{k}"]
        gd = p.grad.double()
        np.testing.assert_allclose([float(gd.abs().sum()), float(gd.pow(2).sum())], s[1:], rtol=5e-3, atol=1e-7, err_msg=k)
        n += 1
    assert n >= 50
    params = dict(m.named_parameters())
    for k in ("prompt_fusion.bias", "prompt_encoder.enc1.block.0.conv.0.weight",
              "prompt_encoder.enc2.block.0.conv.3.weight", "out.weight"):
        ref = g[f"prompt/grad/{k}"]
        assert np.abs(params[k].grad.cpu().numpy() - ref).max() <= 1e-2 * max(np.abs(ref).max(), 1e-5), k
    ref = g["prompt/grad/prompt_fusion.weight[:16]"]
    assert np.abs(params["prompt_fusion.weight"].grad[:16].cpu().numpy() - ref).max() <= 1e-2 * np.abs(ref).max()
    # train-mode BatchNorm bookkeeping of the prompt branch and of the dead bottleneck
    bufs = dict(m.named_buffers())
    for k in g:
        if k.startswith("prompt/buf/"):
            np.testing.assert_allclose(bufs[k[len("prompt/buf/"):]].cpu().numpy(), g[k], rtol=1e-4, atol=1e-6, err_msg=k)


def test_clip_unet_prompt_bf16_training_steps(M):
    """the loop body of scripts/prompt_train.py:86-103 on the HIP path: autocast forward of (image, heat map),
    HybridLossBinary, GradScaler + Adam; the sigmoid activation variant applies `activation` (prompt_segmentation.py:95)."""
    from hipseg.optim import Adam
    from models import prompt_segmentation as ps

    m, x, heat, t = _prompt_model(M)
    m.train()
    opt = Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)
    scaler = torch.amp.GradScaler("cuda")
    crit = M.ls.HybridLossBinary()
    losses = []
    for _ in range(5):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda"):
            out = m(x, heat)
            loss = crit(out, t)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for k, p in m.named_parameters()
               if "in_proj" not in k or True)
    m.activation = torch.nn.Sigmoid()
    m.eval()
    with torch.no_grad(), torch.autocast("cuda"):
        y = m(x, heat)
    assert float(y.min()) >= 0.0 and float(y.max()) <= 1.0
    with pytest.raises(ValueError):
        m(x, heat[:, :, :32])

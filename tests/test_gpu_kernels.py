"""Kernel-level parity: each C-ABI entry point vs the CPU oracle (plain PyTorch fp32 on the same
inputs).  Needs a real MI355X.  Tolerances: fp32 kernels 1e-4-class (stated per test); bf16 kernels
are compared against the oracle evaluated on the bf16-rounded operands."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import fill  # noqa: E402


@pytest.fixture(scope="module")
def hs():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import hipseg  # noqa: F401
    from hipseg import _lib as L, ops

    class H:
        pass

    h = H()
    h.L, h.ops = L, ops
    return h


DTYPES = [("fp32", torch.float32, 0), ("bf16", torch.bfloat16, 1)]


def T(name, shape, lo=-1.0, hi=1.0):
    return torch.from_numpy(fill.uniform(name, shape, lo, hi))


def rnd(t, td):
    """round a CPU fp32 tensor through the storage dtype"""
    return t.detach().to(td).float().clone()


def to_dev_nhwc(t, td):
    return t.cuda().to(td).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def tol(td, scale=1.0):
    return (2e-5 * scale + 1e-5, 1e-4) if td == torch.float32 else (1.5e-2 * scale, 1.5e-2)


def check(got, want, td, scale=None, what=""):
    got = got.float().cpu()
    scale = float(want.abs().max()) if scale is None else scale
    atol, rtol = tol(td, max(scale, 1e-6))
    err = (got - want).abs()
    bound = atol + rtol * want.abs()
    bad = err > bound
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.3e} (scale {scale:.3e})"


CONV_CASES = [
    # B, C0, C1, Cout, H, W
    (2, 4, 0, 8, 16, 16),
    (1, 32, 0, 64, 16, 16),
    (1, 8, 0, 8, 8, 24),
    (2, 16, 0, 32, 20, 12),
    (1, 64, 0, 128, 32, 32),
    (1, 8, 8, 8, 16, 16),
    (1, 32, 32, 32, 24, 40),
    (1, 128, 0, 256, 8, 8),
    (1, 24, 0, 40, 16, 16),
    (1, 5, 0, 7, 10, 18),
    # 128-wide tiles with K % 32 == 0: the super-chunk ring kernel (dual source / dual destination, ragged borders,
    # a tall-tile geometry with >= 256 workgroups)
    (1, 64, 64, 128, 24, 40),
    (1, 96, 32, 160, 18, 30),
    (4, 128, 0, 128, 128, 128),
    # N % 128 == 0 and K % 32 == 0: the 16x16x32-MFMA kernel (conv3_m16.hip) -- 16- and 8-row tiles, deep K, several
    # channel tiles, dual source with dual destination in the data gradient, ragged borders, H smaller than a tile
    (2, 256, 0, 256, 20, 36),
    (1, 128, 128, 128, 33, 17),
    (1, 512, 0, 512, 16, 16),
    (2, 128, 0, 384, 16, 32),
    (16, 32, 0, 128, 64, 64),
    # N % 64 == 0 but not % 128, K % 32 == 0: the same kernel with 64-channel workgroup tiles (round 4: two channel blocks
    # x two pixel-row halves per workgroup) -- 16- and 8-row tiles, dual source, the mirror data gradient with a dual
    # destination, three channel tiles, ragged borders
    (4, 128, 0, 64, 64, 64),
    (1, 64, 64, 64, 20, 36),
    (2, 96, 0, 192, 17, 33),
    (16, 128, 0, 64, 32, 32),
    # >= 1024 tiles of 8x16 pixels with <= 64 channels: the weights-stationary persistent kernel (all four
    # K/N shapes between fwd and dgrad, dual source / dual destination, ragged image borders)
    (2, 64, 0, 64, 256, 256),
    (2, 32, 0, 64, 256, 256),
    (4, 32, 32, 32, 130, 250),
    (3, 32, 0, 32, 200, 232),
]


@pytest.mark.parametrize("prec,td,dt", DTYPES, ids=[d[0] for d in DTYPES])
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv3_fwd_stats(hs, prec, td, dt, case):
    B, C0, C1, Cout, H, W = case
    L, ops = hs.L, hs.ops
    x0 = rnd(T("k.x0", (B, C0, H, W)), td)
    x1 = rnd(T("k.x1", (B, C1, H, W)), td) if C1 else None
    w = T("k.w", (Cout, C0 + C1, 3, 3), -0.3, 0.3)
    b = T("k.b", (Cout,), -0.5, 0.5)
    xin = torch.cat([x0, x1], 1) if C1 else x0
    want = F.conv2d(xin, rnd(w, td), b, padding=1)
    dx0 = to_dev_nhwc(x0, td)
    dx1 = to_dev_nhwc(x1, td) if C1 else None
    wp = ops._pack_conv(w.cuda(), dt, False)
    out = ops.nhwc_empty(B, Cout, H, W, td, "cuda")
    mt = L.conv_mtiles(B, H, W)
    stats = torch.full((mt, 2, Cout), float("nan"), device="cuda")
    s = ops._stream()
    bd = b.cuda()  # keep alive: raw pointers are passed
    L.conv_igemm(dt, L.CONV3, ops.ptr(dx0), C0, ops.ptr(dx1), C1, ops.ptr(wp), ops.ptr(bd), ops.ptr(out), Cout, 0, 0,
                 ops.ptr(stats), B, H, W, s)
    torch.cuda.synchronize()
    check(out, want, td, what="conv3 fwd")
    rows = L.conv_stats_rows(dt, L.CONV3, C0, C1, Cout, 0, B, H, W)
    assert 0 < rows <= mt
    st = stats[:rows].cpu().double().sum(0)
    ssum = want.double().sum((0, 2, 3))
    ssq = (want.double() ** 2).sum((0, 2, 3))
    n = B * H * W
    rt = 1e-4 if td == torch.float32 else 2e-2
    np.testing.assert_allclose(st[0].numpy(), ssum.numpy(), rtol=rt, atol=rt * n * 0.05)
    np.testing.assert_allclose(st[1].numpy(), ssq.numpy(), rtol=rt, atol=rt * n * 0.05)


@pytest.mark.parametrize("prec,td,dt", DTYPES, ids=[d[0] for d in DTYPES])
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv3_affine_relu_inference_epilogue(hs, prec, td, dt, case):
    """hipseg_conv_affine_relu == relu(conv3x3(x) * scale + shift) (conv -> eval BatchNorm -> ReLU folded into the conv
    epilogue; processing_blocks.py:42-48 under model.eval()), on every dispatch shape of the conv tests"""
    B, C0, C1, Cout, H, W = case
    L, ops = hs.L, hs.ops
    x0 = rnd(T("k.x0", (B, C0, H, W)), td)
    x1 = rnd(T("k.x1", (B, C1, H, W)), td) if C1 else None
    w = T("k.w", (Cout, C0 + C1, 3, 3), -0.3, 0.3)
    scale = T("k.sc", (Cout,), 0.2, 1.5)
    scale[::3] *= -1.0  # negative BatchNorm weights exist too
    shift = T("k.sh", (Cout,), -1.0, 1.0)
    xin = torch.cat([x0, x1], 1) if C1 else x0
    want = torch.relu(F.conv2d(xin, rnd(w, td), None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    dx0 = to_dev_nhwc(x0, td)
    dx1 = to_dev_nhwc(x1, td) if C1 else None
    wp = ops._pack_conv(w.cuda(), dt, False)
    out = ops.nhwc_empty(B, Cout, H, W, td, "cuda")
    sc, sh = scale.cuda(), shift.cuda()
    L.conv_affine_relu(dt, ops.ptr(dx0), C0, ops.ptr(dx1), C1, ops.ptr(wp), ops.ptr(sc), ops.ptr(sh), ops.ptr(out), Cout,
                       B, H, W, ops._stream())
    torch.cuda.synchronize()
    check(out, want, td, what="conv3 + affine + relu")
    assert float(out.float().min()) >= 0.0 and float((out == 0).float().mean()) > 0.05  # the ReLU really clipped


@pytest.mark.parametrize("prec,td,dt", DTYPES, ids=[d[0] for d in DTYPES])
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv3_dgrad_wgrad(hs, prec, td, dt, case):
    B, C0, C1, Cout, H, W = case
    L, ops = hs.L, hs.ops
    Cin = C0 + C1
    x = rnd(T("g.x", (B, Cin, H, W)), td).requires_grad_(True)
    w = T("g.w", (Cout, Cin, 3, 3), -0.3, 0.3)
    wr = rnd(w, td).requires_grad_(True)
    dy = rnd(T("g.dy", (B, Cout, H, W)), td)
    F.conv2d(x, wr, None, padding=1).backward(dy)
    s = ops._stream()
    ddy = to_dev_nhwc(dy, td)
    # dgrad with split destination
    wpt = ops._pack_conv(w.cuda(), dt, True)
    dx0 = ops.nhwc_empty(B, C0, H, W, td, "cuda")
    dx1 = ops.nhwc_empty(B, C1, H, W, td, "cuda") if C1 else None
    L.conv_igemm(dt, L.CONV3, ops.ptr(ddy), Cout, 0, 0, ops.ptr(wpt), 0, ops.ptr(dx0), C0, ops.ptr(dx1), C1, 0, B, H, W, s)
    torch.cuda.synchronize()
    check(dx0, x.grad[:, :C0], td, scale=float(x.grad.abs().max()), what="dgrad out0")
    if C1:
        check(dx1, x.grad[:, C0:], td, scale=float(x.grad.abs().max()), what="dgrad out1")
    # wgrad with dual source
    xd = x.detach()
    p0 = to_dev_nhwc(xd[:, :C0], td)
    p1 = to_dev_nhwc(xd[:, C0:], td) if C1 else None
    dw = torch.full((Cout, Cin, 3, 3), float("nan"), device="cuda")
    ops._wgrad(dt, L.CONV3, p0, p1, ddy, dw, B, H, W)
    torch.cuda.synchronize()
    # operands are exactly representable in the storage dtype, products accumulate in fp32
    g = wr.grad
    err = (dw.cpu() - g).abs().max()
    assert err <= 2e-4 * max(1.0, float(g.abs().max())), f"wgrad max err {float(err):.3e} vs scale {float(g.abs().max()):.3e}"


CONVT_CASES = [(2, 16, 8, 8, 8), (1, 64, 32, 4, 12), (1, 128, 64, 16, 16), (1, 12, 6, 5, 7),
               # the one-tap GEMM kernel (C0 % 64 == 0, N % 128 == 0): ragged tiles, several images, K chunks across taps
               (2, 256, 128, 7, 9), (1, 512, 256, 3, 5), (3, 64, 64, 18, 17),
               # the LDS-free streaming kernel (convt_stream.hip: W % 16 == 0, small channel counts): forward K = 64 / 128,
               # data gradient K = 128 / 256, a pixel count that leaves a ragged tail for the 2-block unroll
               (2, 64, 32, 32, 48), (1, 128, 64, 16, 32), (3, 64, 32, 5, 16), (1, 64, 64, 8, 16)]


@pytest.mark.parametrize("prec,td,dt", DTYPES, ids=[d[0] for d in DTYPES])
@pytest.mark.parametrize("case", CONVT_CASES, ids=[str(c) for c in CONVT_CASES])
def test_convT(hs, prec, td, dt, case):
    B, Cin, Cout, H, W = case
    ops = hs.ops
    x = rnd(T("t.x", (B, Cin, H, W)), td).requires_grad_(True)
    w = T("t.w", (Cin, Cout, 2, 2), -0.3, 0.3)
    wr = rnd(w, td).requires_grad_(True)
    b = T("t.b", (Cout,), -0.5, 0.5).requires_grad_(True)
    dy = rnd(T("t.dy", (B, Cout, 2 * H, 2 * W)), td)
    y = F.conv_transpose2d(x, wr, b, stride=2)
    y.backward(dy)
    dx_ = to_dev_nhwc(x.detach(), td).requires_grad_(True)
    dw_ = w.cuda().requires_grad_(True)
    db_ = b.detach().cuda().requires_grad_(True)
    yy = ops.ConvT2x2Fn.apply(dx_, dw_, db_)
    yy.backward(to_dev_nhwc(dy, td))
    torch.cuda.synchronize()
    check(yy.detach(), y.detach(), td, what="convT fwd")
    check(dx_.grad, x.grad, td, what="convT dgrad")
    assert (dw_.grad.cpu() - wr.grad).abs().max() <= 2e-4 * max(1.0, float(wr.grad.abs().max()))
    rt = 1e-4 if td == torch.float32 else 1e-2
    np.testing.assert_allclose(db_.grad.cpu().numpy(), b.grad.numpy(), rtol=rt, atol=rt * float(b.grad.abs().max()))


# B, C (= K = N), H, W: the second conv of a ConvBlock at the U-Net's >= 128-channel levels (16- and 8-row tiles, ragged
# borders, several channel tiles)
DGRAD_BNSTATS_CASES = [(2, 128, 32, 32), (1, 256, 20, 36), (4, 128, 64, 64), (1, 512, 16, 16), (2, 128, 9, 17),
                       (2, 192, 24, 40), (8, 192, 32, 32)]  # (64-channel tiles: two statistics rows per tile)


@pytest.mark.parametrize("case", DGRAD_BNSTATS_CASES, ids=[str(c) for c in DGRAD_BNSTATS_CASES])
def test_conv3_dgrad_with_bn_backward_sums(hs, case):
    """hipseg_conv3_dgrad_bnstats == hipseg_conv_igemm (bit for bit on the data gradient) + hipseg_bn_bwd_reduce on its
    output (the finalised [sum g | sum g * xhat] vectors; fp32 sums in a different order)."""
    B, C, H, W = case
    L, ops = hs.L, hs.ops
    td, dt = torch.bfloat16, L.BF16
    rows = L.conv3_dgrad_bnstats_rows(dt, C, C, B, H, W)
    assert rows > 0
    dy = to_dev_nhwc(rnd(T("db.dy", (B, C, H, W), -1, 1), td), td)
    raw = to_dev_nhwc(rnd(T("db.raw", (B, C, H, W), -2, 2), td), td)
    w = T("db.w", (C, C, 3, 3), -0.05, 0.05).cuda()
    wpt = ops._pack_conv(w, dt, True)
    bn = torch.cat([T("db.mean", (C,), -0.5, 0.5), T("db.is", (C,), 0.5, 2.0), T("db.sc", (C,), -1.5, 1.5),
                    T("db.sh", (C,), -0.5, 0.5)]).cuda()
    s = ops._stream()
    # separate kernels
    ref = ops.nhwc_empty(B, C, H, W, td, "cuda")
    L.conv_igemm(dt, L.CONV3, ops.ptr(dy), C, 0, 0, ops.ptr(wpt), 0, ops.ptr(ref), C, 0, 0, 0, B, H, W, s)
    nblk = L.bn_bwd_blocks(B, H, W, C, dt, 0)
    part = torch.empty(nblk * 2 * C, device="cuda")
    bp = bn.data_ptr()
    L.bn_bwd_reduce(dt, ops.ptr(ref), ops.ptr(raw), bp, bp + 4 * C, bp + 8 * C, bp + 12 * C, ops.ptr(part), B, H, W, C, 0, s)
    want = torch.empty(2 * C, device="cuda")
    L.colsum_finalize(ops.ptr(part), nblk, 2, C, ops.ptr(want), 0, s)
    # fused
    out = ops.nhwc_empty(B, C, H, W, td, "cuda")
    part2 = torch.full((rows * 2 * C,), float("nan"), device="cuda")
    L.conv3_dgrad_bnstats(dt, ops.ptr(dy), C, ops.ptr(wpt), ops.ptr(out), C, ops.ptr(raw), bp, ops.ptr(part2), B, H, W, s)
    got = torch.empty(2 * C, device="cuda")
    L.colsum_finalize(ops.ptr(part2), rows, 2, C, ops.ptr(got), 0, s)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert bool(torch.isfinite(part2).all())
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-5 * max(scale, 1.0) + 1e-6 * (B * H * W) ** 0.5, (got - want).abs().max()
    # an exact statement on the masks: the double-precision sums from the stored tensors
    o64, x64 = out.double(), raw.double()
    mean, istd, sc, sh = (bn[i * C:(i + 1) * C].double().view(1, C, 1, 1) for i in range(4))
    g = torch.where((raw.float() * sc.float() + sh.float()) > 0, o64, torch.zeros_like(o64))
    ex = torch.cat([g.sum((0, 2, 3)), (g * ((x64 - mean) * istd)).sum((0, 2, 3))])
    assert float((got.double() - ex).abs().max()) <= 5e-5 * max(float(ex.abs().max()), 1.0)


# B, CUa0, CUa1, CUb (= CV), H, W: encoder block (single source), decoder block (dual source), more channel tiles, a
# pixel grid with fewer tiles than CUs / tiles (the split count is capped)
# (on 256 CUs: 42 splits x 6 tiles = 252 workgroups + 4 of grid padding; 21 x 12; 10 x 24 = 240; 85 x 3 with the narrow
# reduction; 2 tiles)
WGRAD_PAIR_CASES = [(16, 64, 0, 128, 64, 64), (8, 128, 128, 128, 64, 64), (2, 128, 0, 256, 32, 48), (8, 64, 64, 64, 64, 64),
                    (1, 64, 0, 64, 16, 32)]


@pytest.mark.parametrize("case", WGRAD_PAIR_CASES, ids=[str(c) for c in WGRAD_PAIR_CASES])
def test_wgrad_pair_matches_two_launches(hs, case):
    """hipseg_conv_wgrad_pair (both 3x3 weight gradients of a ConvBlock in one launch) against two hipseg_conv_wgrad
    launches (fp32 sums split differently over the pixels) and against autograd's fp32 weight gradient."""
    B, Ca0, Ca1, Cb, H, W = case
    L, ops = hs.L, hs.ops
    td, dt = torch.bfloat16, L.BF16
    CV = Cb
    assert L.conv_wgrad_pair_applies(dt, Ca0, Ca1, Cb, CV, B, H, W) == 1
    xa0 = rnd(T("wp.xa0", (B, Ca0, H, W), -1, 1), td)
    xa1 = rnd(T("wp.xa1", (B, Ca1, H, W), -1, 1), td) if Ca1 else None
    xb = rnd(T("wp.xb", (B, Cb, H, W), -1, 1), td)
    qa = rnd(T("wp.qa", (B, CV, H, W), -1, 1), td)
    qb = rnd(T("wp.qb", (B, CV, H, W), -1, 1), td)
    d = {k: (to_dev_nhwc(v, td) if v is not None else None) for k, v in dict(xa0=xa0, xa1=xa1, xb=xb, qa=qa, qb=qb).items()}
    Ca = Ca0 + Ca1
    slabs = torch.empty(max(L.wgrad_workspace_elems(L.CONV3, Ca, CV, B, H, W), L.wgrad_workspace_elems(L.CONV3, Cb, CV, B, H, W)),
                        device="cuda")
    s = ops._stream()
    ra, rb = torch.empty(CV, Ca, 3, 3, device="cuda"), torch.empty(CV, Cb, 3, 3, device="cuda")
    L.conv_wgrad(dt, L.CONV3, ops.ptr(d["xa0"]), Ca0, ops.ptr(d["xa1"]), Ca1, ops.ptr(d["qa"]), CV, ops.ptr(ra), ops.ptr(slabs),
                 B, H, W, s)
    L.conv_wgrad(dt, L.CONV3, ops.ptr(d["xb"]), Cb, 0, 0, ops.ptr(d["qb"]), CV, ops.ptr(rb), ops.ptr(slabs), B, H, W, s)
    ga = torch.full((CV, Ca, 3, 3), float("nan"), device="cuda")
    gb = torch.full((CV, Cb, 3, 3), float("nan"), device="cuda")
    slabs.fill_(float("nan"))
    L.conv_wgrad_pair(dt, ops.ptr(d["xa0"]), Ca0, ops.ptr(d["xa1"]), Ca1, ops.ptr(d["qa"]), ops.ptr(ga), ops.ptr(d["xb"]), Cb,
                      ops.ptr(d["qb"]), ops.ptr(gb), CV, ops.ptr(slabs), B, H, W, s)
    torch.cuda.synchronize()
    for got, ref, x, q in ((ga, ra, torch.cat([xa0, xa1], 1) if Ca1 else xa0, qa), (gb, rb, xb, qb)):
        assert bool(torch.isfinite(got).all())
        scale = max(1.0, float(ref.abs().max()))
        assert float((got - ref).abs().max()) <= 2e-5 * scale * (B * H * W) ** 0.5 / 16
        w = torch.zeros(CV, x.shape[1], 3, 3, requires_grad=True)
        F.conv2d(x, w, None, padding=1).backward(q)
        assert float((got.cpu() - w.grad).abs().max()) <= 2e-4 * max(1.0, float(w.grad.abs().max()))


# B, C0, C1, Cout, H, W: the 1x1 fusion conv of ClipUnetPrompt at training size (dual source 512 + 512 -> 512 at H/8),
# single source, ragged channel counts / pixel counts
CONV1_CASES = [(2, 512, 512, 512, 8, 8), (1, 64, 0, 128, 16, 16), (2, 24, 8, 40, 5, 7), (1, 128, 128, 64, 12, 20)]


@pytest.mark.parametrize("prec,td,dt", DTYPES, ids=[d[0] for d in DTYPES])
@pytest.mark.parametrize("case", CONV1_CASES, ids=[str(c) for c in CONV1_CASES])
def test_conv1x1_dual_source_fwd_bwd(hs, prec, td, dt, case):
    """ops.Conv1x1Fn (nn.Conv2d(k=1) on cat([x0, x1])) against F.conv2d: forward, both data gradients, weight and
    bias gradients."""
    B, C0, C1, Cout, H, W = case
    ops = hs.ops
    x0 = rnd(T("c1.x0", (B, C0, H, W), -1, 1), td).requires_grad_(True)
    x1 = rnd(T("c1.x1", (B, C1, H, W), -1, 1), td).requires_grad_(True) if C1 else None
    w = T("c1.w", (Cout, C0 + C1, 1, 1), -0.2, 0.2)
    wr = rnd(w, td).requires_grad_(True)
    b = T("c1.b", (Cout,), -0.5, 0.5).requires_grad_(True)
    dy = rnd(T("c1.dy", (B, Cout, H, W), -1, 1), td)
    y = F.conv2d(torch.cat([x0, x1], 1) if C1 else x0, wr, b)
    y.backward(dy)
    d0 = to_dev_nhwc(x0.detach(), td).requires_grad_(True)
    d1 = to_dev_nhwc(x1.detach(), td).requires_grad_(True) if C1 else None
    dw_ = w.cuda().requires_grad_(True)
    db_ = b.detach().cuda().requires_grad_(True)
    yy = ops.Conv1x1Fn.apply(d0, d1, dw_, db_)
    yy.backward(to_dev_nhwc(dy, td))
    torch.cuda.synchronize()
    check(yy.detach(), y.detach(), td, what="conv1x1 fwd")
    check(d0.grad, x0.grad, td, what="conv1x1 dgrad (first source)")
    if C1:
        check(d1.grad, x1.grad, td, what="conv1x1 dgrad (second source)")
    assert (dw_.grad.cpu() - wr.grad).abs().max() <= 2e-4 * max(1.0, float(wr.grad.abs().max()))
    rt = 1e-4 if td == torch.float32 else 1e-2
    np.testing.assert_allclose(db_.grad.cpu().numpy(), b.grad.numpy(), rtol=rt, atol=rt * float(b.grad.abs().max()))


@pytest.mark.parametrize("prec,td,dt", DTYPES, ids=[d[0] for d in DTYPES])
@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("shape", [(2, 8, 16, 16), (1, 64, 8, 24), (2, 6, 4, 6), (1, 256, 8, 8)])
def test_bn_relu_pool_fwd_bwd(hs, prec, td, dt, pool, shape):
    B, C, H, W = shape
    L, ops = hs.L, hs.ops
    x = rnd(T("bn.x", shape, -2, 2), td).requires_grad_(True)
    gamma = T("bn.g", (C,), 0.5, 1.5).requires_grad_(True)
    beta = T("bn.b", (C,), -0.3, 0.3).requires_grad_(True)
    y = F.relu(F.batch_norm(x, None, None, gamma, beta, True, 0.1, 1e-5))
    if pool:
        y = F.max_pool2d(y, 2, 2)
    dy = rnd(T("bn.dy", tuple(y.shape)), td)
    y.backward(dy)
    # device: statistics from exact double sums (the conv epilogue is tested separately)
    xd = x.detach().double()
    stats = torch.stack([xd.sum((0, 2, 3)), (xd ** 2).sum((0, 2, 3))]).float().reshape(1, 2, C).cuda()
    bn = ops._BN(C, "cuda")
    s = ops._stream()
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    gd, bd = gamma.detach().cuda(), beta.detach().cuda()  # keep alive: raw pointers are passed
    L.bn_finalize(ops.ptr(stats), 1, C, float(B * H * W), ops.ptr(gd), ops.ptr(bd),
                  1e-5, 0.1, ops.ptr(rm), ops.ptr(rv), ops.ptr(nbt), ops.ptr(bn.mean), ops.ptr(bn.invstd),
                  ops.ptr(bn.scale), ops.ptr(bn.shift), s)
    raw = to_dev_nhwc(x.detach(), td)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    act = ops.nhwc_empty(B, C, Ho, Wo, td, "cuda")
    L.bn_relu_apply(dt, ops.ptr(raw), ops.ptr(bn.scale), ops.ptr(bn.shift), ops.ptr(act), B, H, W, C, int(pool), s)
    torch.cuda.synchronize()
    check(act, y.detach(), td, what="bn+relu(+pool) fwd")
    assert int(nbt) == 1
    np.testing.assert_allclose(rm.cpu().numpy(), 0.1 * xd.mean((0, 2, 3)).float().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), (0.9 + 0.1 * xd.var((0, 2, 3), unbiased=True)).float().numpy(), rtol=1e-4)
    draw, dg, dbe, dbias = ops._bn_relu_bwd(dt, to_dev_nhwc(dy, td), raw, bn, True, pool, torch.zeros(C, device="cuda"), gd, bd)
    torch.cuda.synchronize()
    check(draw, x.grad, td, scale=float(x.grad.abs().max()), what="bn bwd dx")
    rt = 2e-4 if td == torch.float32 else 2e-2
    np.testing.assert_allclose(dg.cpu().numpy(), gamma.grad.numpy(), rtol=rt, atol=rt * float(gamma.grad.abs().max()))
    np.testing.assert_allclose(dbe.cpu().numpy(), beta.grad.numpy(), rtol=rt, atol=rt * float(beta.grad.abs().max()))
    # conv-bias gradient = sum of the (un-rounded fp32) dx; in train mode the exact value is 0
    want_db = x.grad.double().sum((0, 2, 3)).numpy()
    mag = x.grad.double().abs().sum((0, 2, 3)).numpy()
    assert np.all(np.abs(dbias.cpu().numpy() - want_db) <= 1e-5 * mag + 1e-6)


@pytest.mark.parametrize("prec,td,dt", DTYPES, ids=[d[0] for d in DTYPES])
def test_stem_head(hs, prec, td, dt):
    ops = hs.ops
    B, H, W = 2, 16, 24
    x = T("s.x", (B, 3, H, W), 0, 1)
    w = T("s.w", (32, 3, 1, 1)).requires_grad_(True)
    b = T("s.b", (32,), -0.1, 0.1).requires_grad_(True)
    y = F.conv2d(x, w, b)
    dy = rnd(T("s.dy", (B, 32, H, W)), td)
    y.backward(dy)
    wd, bd = w.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    yy = ops.StemFn.apply(x.cuda(), wd, bd, prec)
    yy.backward(to_dev_nhwc(dy, td))
    torch.cuda.synchronize()
    check(yy.detach(), y.detach(), td, what="stem fwd")
    assert (wd.grad.cpu() - w.grad).abs().max() <= 2e-4 * float(w.grad.abs().max())
    assert (bd.grad.cpu() - b.grad).abs().max() <= 2e-4 * float(b.grad.abs().max())
    # two aliases of the output (the U-Nets: first encoder block + last decoder block's skip): each consumer's gradient
    # arrives on its own, hipseg_stem_bwd2 reads both -> gradients of y used twice; one alias unused -> the plain result
    dy2 = rnd(T("s.dy2", (B, 32, H, W)), td)
    w2 = w.detach().clone().requires_grad_(True)
    b2 = b.detach().clone().requires_grad_(True)
    y2 = F.conv2d(x, w2, b2)
    ((y2 * dy).sum() + (y2 * dy2).sum()).backward()
    wd, bd = w.detach().cuda().requires_grad_(True), b.detach().cuda().requires_grad_(True)
    ya, yb = ops.StemFn.apply(x.cuda(), wd, bd, prec, True)
    assert ya.data_ptr() == yb.data_ptr()
    ((ya.float() * to_dev_nhwc(dy, td).float()).sum() + (yb.float() * to_dev_nhwc(dy2, td).float()).sum()).backward()
    torch.cuda.synchronize()
    assert (wd.grad.cpu() - w2.grad).abs().max() <= 2e-4 * float(w2.grad.abs().max())
    assert (bd.grad.cpu() - b2.grad).abs().max() <= 2e-4 * float(b2.grad.abs().max())
    wd.grad = bd.grad = None
    ya, yb = ops.StemFn.apply(x.cuda(), wd, bd, prec, True)
    yb.backward(to_dev_nhwc(dy, td))  # (only the second alias is used)
    torch.cuda.synchronize()
    assert (wd.grad.cpu() - w.grad).abs().max() <= 2e-4 * float(w.grad.abs().max())
    # head
    for cout in (3, 1):
        hx = rnd(T("h.x", (B, 32, H, W)), td).requires_grad_(True)
        hw = T("h.w", (cout, 32, 1, 1)).requires_grad_(True)
        hb = T("h.b", (cout,), -0.1, 0.1).requires_grad_(True)
        lg = F.conv2d(hx, hw, hb)
        dl = T("h.dl", (B, cout, H, W))
        lg.backward(dl)
        dxh = to_dev_nhwc(hx.detach(), td).requires_grad_(True)
        dwh, dbh = hw.detach().cuda().requires_grad_(True), hb.detach().cuda().requires_grad_(True)
        lgd = ops.HeadFn.apply(dxh, dwh, dbh)
        lgd.backward(dl.cuda())
        torch.cuda.synchronize()
        assert lgd.dtype == torch.float32 and lgd.is_contiguous()
        assert (lgd.detach().cpu() - lg.detach()).abs().max() <= 1e-4
        check(dxh.grad, hx.grad, td, what="head dx")
        assert (dwh.grad.cpu() - hw.grad).abs().max() <= 2e-4 * float(hw.grad.abs().max())
        assert (dbh.grad.cpu() - hb.grad).abs().max() <= 2e-4 * float(hb.grad.abs().max())


@pytest.mark.parametrize("prec,td,dt", DTYPES, ids=[d[0] for d in DTYPES])
@pytest.mark.parametrize("geo", [(16, 16, 8, 8), (8, 24, 4, 12), (8, 8, 16, 16), (6, 10, 6, 10), (4, 4, 1, 1),
                                 # the production case (dec1: 64 -> 32) and non-integer ratios both ways: the backward
                                 # gather's candidate window must contain every output pixel that sampled an input pixel
                                 (64, 64, 32, 32), (33, 47, 12, 20), (12, 20, 33, 47), (31, 17, 30, 16), (5, 7, 64, 2)])
def test_bilinear(hs, prec, td, dt, geo):
    ops = hs.ops
    Hi, Wi, Ho, Wo = geo
    x = rnd(T("bl.x", (2, 8, Hi, Wi)), td).requires_grad_(True)
    y = F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=True)
    dy = rnd(T("bl.dy", (2, 8, Ho, Wo)), td)
    y.backward(dy)
    xd = to_dev_nhwc(x.detach(), td).requires_grad_(True)
    yd = ops.BilinearFn.apply(xd, Ho, Wo)
    yd.backward(to_dev_nhwc(dy, td))
    torch.cuda.synchronize()
    check(yd.detach(), y.detach(), td, what="bilinear fwd")
    check(xd.grad, x.grad, td, what="bilinear bwd")


def test_losses_and_confusion(hs, golden):
    ops = hs.ops
    g = golden("losses")
    logits = T("loss.logits", (2, 3, 32, 32), -3.0, 3.0)
    tgt = torch.from_numpy(fill.randint("loss.t", (2, 32, 32), 3))
    lg = logits.cuda().requires_grad_(True)
    ce = ops.CrossEntropyFn.apply(lg, tgt.cuda())
    (ce * 3.0).backward()
    assert abs(float(ce) - float(g["ce"])) < 2e-6
    np.testing.assert_allclose(lg.grad.cpu().numpy(), 3.0 * g["ce_grad"], rtol=1e-4, atol=1e-9)
    # ignore_index = -100
    t2 = tgt.clone()
    t2[0, :5] = -100
    l2 = logits.clone().requires_grad_(True)
    want = F.cross_entropy(l2, t2)
    want.backward()
    l2d = logits.cuda().requires_grad_(True)
    got = ops.CrossEntropyFn.apply(l2d, t2.cuda())
    got.backward()
    assert abs(float(got) - float(want)) < 2e-6
    np.testing.assert_allclose(l2d.grad.cpu().numpy(), l2.grad.numpy(), rtol=1e-4, atol=1e-9)
    conf = ops.confusion_matrix(logits.cuda(), tgt.cuda()).cpu()
    pred = logits.argmax(1)
    for t in range(3):
        for p in range(3):
            assert int(conf[t, p]) == int(((tgt == t) & (pred == p)).sum())
    # BCE + Dice
    from oracle import torch_ref as R
    bl = T("loss.blogits", (2, 1, 32, 32), -3.0, 3.0)
    bt = torch.from_numpy(fill.randint("loss.bt", (2, 32, 32), 2)).float().unsqueeze(1)
    blr = bl.clone().requires_grad_(True)
    want = R.hybrid_loss_binary(blr, bt)
    want.backward()
    bld = bl.cuda().requires_grad_(True)
    got = ops.BceDiceFn.apply(bld, bt.cuda())
    got.backward()
    assert abs(float(got) - float(want)) < 5e-6
    np.testing.assert_allclose(bld.grad.cpu().numpy(), blr.grad.numpy(), rtol=2e-4, atol=1e-8)
    # empty target: dice masked, BCE only
    z = torch.zeros_like(bt)
    want0 = R.hybrid_loss_binary(bl, z)
    got0 = ops.BceDiceFn.apply(bl.cuda(), z.cuda())
    assert abs(float(got0) - float(want0)) < 5e-6


@pytest.mark.parametrize("prec,td,dt", DTYPES, ids=[d[0] for d in DTYPES])
def test_pack_batch_matches_per_layer_pack(hs, prec, td, dt):
    """hipseg_pack_batch (all weights of a module, one launch) == the per-layer pack entry points, bit for bit."""
    L, ops = hs.L, hs.ops
    net = torch.nn.Sequential(torch.nn.Conv2d(24, 40, 3, padding=1), torch.nn.ConvTranspose2d(64, 32, 2, stride=2),
                              torch.nn.Conv2d(64, 64, 3, padding=1), torch.nn.Conv2d(3, 32, 1),
                              torch.nn.ConvTranspose2d(12, 6, 2, stride=2),
                              # unpadded 3x3 shapes: the LDS-staged tile path of the batch kernel (32- and 64-wide k tiles)
                              torch.nn.Conv2d(32, 64, 3, padding=1), torch.nn.Conv2d(128, 256, 3, padding=1),
                              torch.nn.Conv2d(256, 128, 3, padding=1)).cuda()
    want = []
    for m in net:
        if isinstance(m, torch.nn.ConvTranspose2d):
            want.append((ops._pack_convT(m.weight, dt, False), ops._pack_convT(m.weight, dt, True)))
        elif m.kernel_size == (3, 3):
            want.append(ops._pack_conv_both(m.weight, dt))
    ops.prepack(net, prec)
    torch.cuda.synchronize()
    got = [ops._cached_pack(m.weight, dt) for m in net if not (isinstance(m, torch.nn.Conv2d) and m.kernel_size == (1, 1))]
    assert len(got) == len(want) == 7 and all(g is not None for g in got)
    for (gp, gt), (wp, wt) in zip(got, want):
        assert torch.equal(gp.float(), wp.float()) and torch.equal(gt.float(), wt.float())
    # a weight modified in place after prepack is no longer served from the cache
    with torch.no_grad():
        net[0].weight.add_(1.0)
    assert ops._cached_pack(net[0].weight, dt) is None


def test_hip_adam_matches_torch_adam():
    """hipseg.optim.Adam == torch.optim.Adam (the reference's optimizer_class, models/model_wrappers.py:40,124) over
    several steps: weight decay, odd sizes (vector tails, unaligned views), a parameter without gradient, the
    GradScaler contract (grad_scale division, found_inf skips the step and does not count it) and hipGraph replay."""
    from hipseg.optim import Adam

    torch.manual_seed(0)
    shapes = [(3,), (64, 32, 3, 3), (513,), (4099,), (1, 1), (128, 64, 2, 2), (9001,)]
    ref_p = [torch.randn(s, dtype=torch.float32).requires_grad_(True) for s in shapes]
    flat = torch.zeros(sum(p.numel() for p in ref_p) + 1, device="cuda")
    dev_p = [p.detach().clone().cuda().requires_grad_(True) for p in ref_p]
    ref_p.append(torch.randn(7).requires_grad_(True))          # never receives a gradient
    dev_p.append(ref_p[-1].detach().clone().cuda().requires_grad_(True))
    kw = dict(lr=1e-2, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)
    ref = torch.optim.Adam(ref_p, **kw)
    opt = Adam(dev_p, **kw)
    v0 = [p._version for p in dev_p]
    scale = torch.tensor([1024.0], device="cuda")
    applied = 0
    for it in range(7):
        grads = [torch.randn(s) * (10.0 ** (it % 3 - 1)) for s in shapes]
        overflow = it == 3
        o = 1  # gradients live at odd offsets of one flat buffer on some steps (unaligned -> scalar path)
        for p, rp, g in zip(dev_p, ref_p, grads):
            if it % 2:
                view = flat[o:o + g.numel()].view(g.shape)
                view.copy_(g * 1024.0)
                p.grad = view
                o += g.numel()
            else:
                p.grad = (g * 1024.0).cuda()
            rp.grad = g.clone()
        opt.grad_scale = scale
        opt.found_inf = torch.tensor([1.0 if overflow else 0.0], device="cuda")
        opt.step()
        if not overflow:
            ref.step()
            applied += 1
        for p, rp in zip(dev_p, ref_p):
            np.testing.assert_allclose(p.detach().cpu().numpy(), rp.detach().numpy(), rtol=2e-5, atol=2e-6)
    assert opt.step_count() == applied == 6
    assert torch.equal(dev_p[-1].cpu(), ref_p[-1])  # no gradient -> untouched (no weight decay either), as torch
    np.testing.assert_allclose(opt.state[dev_p[1]]["exp_avg_sq"].cpu().numpy(), ref.state[ref_p[1]]["exp_avg_sq"].numpy(),
                               rtol=1e-4, atol=1e-9)
    del opt.grad_scale, opt.found_inf
    # hipGraph: static gradient buffers, the device-side step counter advances on every replay
    for p in dev_p[:-1]:
        p.grad = torch.zeros_like(p)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        opt.step()  # warm-up: builds the table for these gradient addresses
        ref_grads = [torch.zeros(sh) for sh in shapes]
        for rp, g in zip(ref_p, ref_grads):
            rp.grad = g
        ref.step()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            opt.step()
        for it in range(3):
            for p, rp in zip(dev_p[:-1], ref_p):
                g = torch.randn(p.shape)
                p.grad.copy_(g)
                rp.grad = g
            graph.replay()
            ref.step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    assert opt.step_count() == applied + 1 + 3
    # the kernel writes through raw pointers; autograd must still see in-place updates (as after torch.optim.Adam.step)
    assert all(p._version > v for p, v in zip(dev_p[:-1], v0)) and dev_p[-1]._version == v0[-1]
    for p, rp in zip(dev_p, ref_p):
        np.testing.assert_allclose(p.detach().cpu().numpy(), rp.detach().numpy(), rtol=5e-5, atol=5e-6)


def test_hip_adam_state_dict_roundtrip_with_torch_adam():
    """hipseg.optim.Adam's state_dict has torch.optim.Adam's layout: a run continued from a checkpoint -- loaded into a
    fresh hipseg Adam or into torch.optim.Adam -- follows the uninterrupted run."""
    from hipseg.optim import Adam

    torch.manual_seed(1)
    shapes = [(64, 32, 3, 3), (17,), (3, 5)]
    init = [torch.randn(s) for s in shapes]
    grads = [[torch.randn(s) for s in shapes] for _ in range(6)]
    kw = dict(lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3)

    def run(opt_cls, params, steps, opt=None):
        opt = opt or opt_cls(params, **kw)
        for gs in steps:
            for p, g in zip(params, gs):
                p.grad = g.to(p.device).clone()
            opt.step()
        return opt

    full = [p.clone().cuda().requires_grad_(True) for p in init]
    run(Adam, full, grads)
    half = [p.clone().cuda().requires_grad_(True) for p in init]
    o1 = run(Adam, half, grads[:3])
    sd = o1.state_dict()
    assert sorted(sd["state"][0]) == ["exp_avg", "exp_avg_sq", "step"] and float(sd["state"][0]["step"]) == 3.0
    # (a) continue in a fresh hipseg Adam
    cont = [p.detach().clone().requires_grad_(True) for p in half]
    o2 = Adam(cont, **kw)
    o2.load_state_dict(sd)
    # a checkpoint saved right after resuming (before any step) must keep the step count: with step = 0 the bias
    # correction would restart and the first update after the next resume would be ~10x too large
    assert o2.step_count() == 3
    sd_again = o2.state_dict()
    assert float(sd_again["state"][0]["step"]) == 3.0
    o2b = Adam([p.detach().clone().requires_grad_(True) for p in half], **kw)
    o2b.load_state_dict(sd_again)
    assert o2b.step_count() == 3
    run(Adam, cont, grads[3:], opt=o2)
    assert o2.step_count() == 6
    for a, b in zip(cont, full):
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-6, atol=1e-7)
    # (b) continue in torch.optim.Adam on the CPU from the same checkpoint
    cpu = [p.detach().cpu().clone().requires_grad_(True) for p in half]
    o3 = torch.optim.Adam(cpu, **kw)
    o3.load_state_dict({"state": {k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in sd["state"].items()},
                        "param_groups": [{**g, **{k: torch.optim.Adam(cpu, **kw).param_groups[0][k]
                                                  for k in torch.optim.Adam(cpu, **kw).param_groups[0] if k not in g}}
                                         for g in sd["param_groups"]]})
    run(torch.optim.Adam, cpu, grads[3:], opt=o3)
    for a, b in zip(cpu, full):
        np.testing.assert_allclose(a.detach().numpy(), b.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)

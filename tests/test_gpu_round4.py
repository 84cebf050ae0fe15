"""Round-4 cases (need a real MI355X):
  * the two-gradient BatchNorm-backward entry points (hipseg_bn_bwd_reduce2 / _apply2) against the one-gradient ones fed
    with the bf16 sum autograd's accumulation pass would have written, and the output-alias form of ConvBlockFn /
    the U-Net (one gradient per consumer) against the single-tensor form, bit for bit;
  * the per-op Python path of ConvBlockFn (what bench.py's roofline leg times: HIPSEG_NO_BLOCK_CALLS / ops.PROFILE) against
    the one-C-call block path that every other test runs, bit for bit;
  * hipseg_conv_igemm(mode = CONV2S2) WITH a bias on a shape the LDS-free streaming kernel takes: the bias must not be
    dropped;
  * the roofline leg's per-launch records carry algorithmic bytes for the HBM-bound groups."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from test_gpu_kernels import T, hs, rnd, to_dev_nhwc  # noqa: E402,F401


@pytest.mark.parametrize("pool", [0, 1])
@pytest.mark.parametrize("prec,td,dt", [("fp32", torch.float32, 0), ("bf16", torch.bfloat16, 1)])
def test_bn_backward_with_two_gradients(hs, prec, td, dt, pool):
    L, ops = hs.L, hs.ops
    B, C, H, W = 3, 64, 12, 20
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    raw = to_dev_nhwc(rnd(T("r4.raw", (B, C, H, W), -2, 2), td), td)
    dy1 = to_dev_nhwc(rnd(T("r4.dy1", (B, C, Ho, Wo)), td), td)
    dy2 = to_dev_nhwc(rnd(T("r4.dy2", (B, C, Ho, Wo)), td), td)
    dsum = (dy1 + dy2)  # what autograd's accumulation writes (rounded to the activation dtype)
    bn = torch.cat([T("r4.mean", (C,), -0.5, 0.5), T("r4.is", (C,), 0.5, 2.0), T("r4.sc", (C,), -1.5, 1.5),
                    T("r4.sh", (C,), -0.5, 0.5)]).cuda()
    bp, s = bn.data_ptr(), ops._stream()
    nblk = L.bn_bwd_blocks(B, H, W, C, dt, pool)

    def run(two):
        part = torch.empty(nblk * 2 * C, device="cuda")
        if two:
            L.bn_bwd_reduce2(dt, ops.ptr(dy1), ops.ptr(dy2), ops.ptr(raw), bp, bp + 4 * C, bp + 8 * C, bp + 12 * C, ops.ptr(part),
                             B, H, W, C, pool, s)
        else:
            L.bn_bwd_reduce(dt, ops.ptr(dsum), ops.ptr(raw), bp, bp + 4 * C, bp + 8 * C, bp + 12 * C, ops.ptr(part), B, H, W, C,
                            pool, s)
        sums = torch.empty(2 * C, device="cuda")
        L.colsum_finalize(ops.ptr(part), nblk, 2, C, ops.ptr(sums), 0, s)
        dx = ops.nhwc_empty(B, C, H, W, td, "cuda")
        if two:
            L.bn_bwd_apply2(dt, ops.ptr(dy1), ops.ptr(dy2), ops.ptr(raw), bp, bp + 4 * C, bp + 8 * C, bp + 12 * C, ops.ptr(sums),
                            float(B * H * W), 0, ops.ptr(dx), 0, B, H, W, C, pool, s)
        else:
            L.bn_bwd_apply(dt, ops.ptr(dsum), ops.ptr(raw), bp, bp + 4 * C, bp + 8 * C, bp + 12 * C, ops.ptr(sums),
                           float(B * H * W), 0, ops.ptr(dx), 0, B, H, W, C, pool, s)
        torch.cuda.synchronize()
        return sums, dx

    s1, d1 = run(False)
    s2, d2 = run(True)
    assert torch.equal(s1, s2) and torch.equal(d1, d2)
    assert float(d1.float().abs().max()) > 0


def _unet_grads(hs, alias):
    import models.UNet as un
    from models.losses import HybridLoss

    torch.manual_seed(5)
    m = un.UNet().cuda().train()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 32, 48, generator=g).cuda()
    t = torch.randint(0, 3, (2, 32, 48), generator=g).cuda()
    old = un._NO_ENC_ALIAS
    un._NO_ENC_ALIAS = not alias
    try:
        with torch.autocast("cuda"):
            out = m(x)
            loss = HybridLoss()(out, t)
        loss.backward()
    finally:
        un._NO_ENC_ALIAS = old
    torch.cuda.synchronize()
    return out.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}


def test_unet_output_alias_per_consumer_is_bit_identical(hs):
    """models/UNet.py:64-72: every encoder output feeds the next block and a decoder's skip input.  One alias per consumer
    (no elementwise gradient sum by autograd; the BatchNorm-backward kernels read both gradients) must give exactly the
    gradients of the single-tensor form."""
    o1, g1 = _unet_grads(hs, alias=False)
    o2, g2 = _unet_grads(hs, alias=True)
    assert torch.equal(o1, o2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k


def test_convblock_per_op_path_equals_block_call(hs):
    """the per-op Python launch sequence (bench.py's profiling path) == hipseg_convblock_forward/backward"""
    from models.processing_blocks import ConvBlock, ConvBlockDownsample, ConvBlockUpsampleSkip

    ops = hs.ops
    cases = [(lambda: ConvBlock(64, 128), [(2, 64, 32, 32)]), (lambda: ConvBlockDownsample(32, 64), [(2, 32, 32, 32)]),
             (lambda: ConvBlockUpsampleSkip(256, 128), [(2, 256, 16, 16), (2, 128, 32, 32)])]
    for make, shapes in cases:
        res = []
        for per_op in (False, True):
            torch.manual_seed(7)
            m = make().cuda().train()
            xs = [to_dev_nhwc(T(f"r4.blk{i}", sh), torch.bfloat16).requires_grad_(True) for i, sh in enumerate(shapes)]
            old = ops._NO_BLOCK_CALLS
            ops._NO_BLOCK_CALLS = per_op
            try:
                with torch.autocast("cuda"):
                    y = m(*xs)
                y.float().square().mean().backward()
            finally:
                ops._NO_BLOCK_CALLS = old
            torch.cuda.synchronize()
            res.append((y.detach(), [x.grad for x in xs], [p.grad.clone() for p in m.parameters()],
                        [b.clone() for b in m.buffers()]))
        (y0, gx0, gp0, b0), (y1, gx1, gp1, b1) = res
        assert torch.equal(y0, y1)
        for a, b in zip(gx0 + gp0 + b0, gx1 + gp1 + b1):
            assert torch.equal(a, b)


def test_conv2s2_with_bias_on_a_stream_shape_keeps_the_bias(hs):
    """mode CONV2S2 (the ConvTranspose2d data gradient as a stride-2 2x2 convolution) through the public entry point with a
    bias, on the shape the LDS-free streaming kernel takes for the U-Net (bf16, 32 input channels, 64 output channels, W
    % 16 == 0): that kernel applies no bias in this mode, so the dispatch must route the call elsewhere."""
    L, ops = hs.L, hs.ops
    td, dt = torch.bfloat16, L.BF16
    B, cout, cin, H, W = 2, 32, 64, 16, 32     # ConvT(cin -> cout): dy has cout channels at (2H, 2W), dx has cin at (H, W)
    w = T("r4.tw", (cin, cout, 2, 2), -0.3, 0.3).cuda()
    bias = T("r4.tb", (cin,), -0.5, 0.5).cuda()
    dy = to_dev_nhwc(rnd(T("r4.tdy", (B, cout, 2 * H, 2 * W)), td), td)
    wpt = ops._pack_convT(w, dt, True)
    outs = []
    for b in (None, bias):
        dx = ops.nhwc_empty(B, cin, H, W, td, "cuda")
        L.conv_igemm(dt, L.CONV2S2, ops.ptr(dy), cout, 0, 0, ops.ptr(wpt), ops.ptr(b), ops.ptr(dx), cin, 0, 0, 0, B, H, W,
                     ops._stream())
        torch.cuda.synchronize()
        outs.append(dx.float())
    ref = F.conv2d(dy.float().cpu(), rnd(w.cpu(), td).permute(0, 1, 2, 3), None, stride=2)  # (B, cin, H, W): w is (cin, cout, 2, 2)
    assert (outs[0].cpu() - ref).abs().max() <= 2e-2 * max(1.0, float(ref.abs().max()))
    diff = (outs[1] - outs[0]).cpu()
    want = bias.cpu().view(1, -1, 1, 1).expand_as(diff)
    assert (diff - want).abs().max() <= 2e-2 * max(1.0, float(ref.abs().max())), "the bias was dropped"


def test_profile_records_carry_bytes_for_hbm_groups(hs):
    import models.UNet as un
    from models.losses import HybridLoss

    ops = hs.ops
    m = un.UNet().cuda().train()
    x = torch.rand(2, 3, 32, 32, device="cuda")
    t = torch.randint(0, 3, (2, 32, 32), device="cuda")
    ops.PROFILE = []
    try:
        with torch.autocast("cuda"):
            loss = HybridLoss()(m(x), t)
        loss.backward()
        torch.cuda.synchronize()
        rec = list(ops.PROFILE)
    finally:
        ops.PROFILE = None
    keys = {r[0] for r in rec}
    for k in ("hbm:bn_relu_apply", "hbm:bn_bwd_apply", "hbm:stem_fwd", "hbm:stem_bwd", "hbm:head_fwd+bn_relu",
              "hbm:head_bwd+bn_bwd_sums",
              "hbm:ce_fwd", "hbm:ce_bwd", "hbm:bilinear_fwd"):
        assert k in keys, (k, sorted(keys))
    for key, flops, nbytes, e0, e1 in rec:
        assert e0.elapsed_time(e1) >= 0.0
        if key.startswith("hbm:"):
            assert flops == 0.0 and nbytes > 0
        else:
            assert flops > 0
    assert np.isfinite(float(loss.detach()))


CONVT_WGRAD_CASES = [  # B, Cin, Cout, H, W   (x is H x W, dy is 2H x 2W)
    (2, 512, 256, 16, 16),   # dec1.up's channel counts: 4 x 8 channel tiles, several pixel splits, XCD-ordered grid
    (4, 64, 32, 32, 32),     # dec4.up's: one channel tile, every workgroup a pixel split
    (1, 136, 40, 8, 16),     # ragged channel tiles on both sides (136 = 128 + 8 input channels, 160 = 128 + 32 columns)
    (3, 128, 64, 24, 48),    # dec3.up's channel counts, non-square
    (2, 64, 32, 6, 16),      # H % 8 != 0: ragged bottom tile rows
    (2, 256, 128, 28, 28),   # ClipUnet-224's dec2.up geometry: ragged on both axes (28 = 3.5 x 8 = 1.75 x 16)
    (1, 128, 64, 5, 7),      # an image smaller than one tile
    (2, 12, 6, 8, 16),       # Cin % 8 != 0: the generic one-tap kernel + column sum inside the same entry point
]


@pytest.mark.parametrize("case", CONVT_WGRAD_CASES, ids=[str(c) for c in CONVT_WGRAD_CASES])
def test_convT_weight_and_bias_gradient_in_one_kernel(hs, case):
    """hipseg_convT_wgrad_bias (bf16) against autograd's ConvTranspose2d backward on the bf16-rounded operands (fp64 on
    the CPU): dW (Cin, Cout, 2, 2) and db (Cout); bit-level determinism (fixed-order slab reduction)."""
    L, ops = hs.L, hs.ops
    B, Cin, Cout, H, W = case
    td, dt = torch.bfloat16, L.BF16
    x = rnd(T("r4.ctx", (B, Cin, H, W)), td)
    dy = rnd(T("r4.ctdy", (B, Cout, 2 * H, 2 * W)), td)
    w = torch.zeros(Cin, Cout, 2, 2, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(x.double(), w, b, stride=2).backward(dy.double())
    xd, dyd = to_dev_nhwc(x, td), to_dev_nhwc(dy, td)
    work = torch.full((L.convT_wgrad_workspace_elems(Cin, Cout, B, H, W),), float("nan"), device="cuda")
    outs = []
    for _ in range(2):
        dw = torch.full((Cin, Cout, 2, 2), float("nan"), device="cuda")
        db = torch.full((Cout,), float("nan"), device="cuda")
        L.convT_wgrad_bias(dt, ops.ptr(dyd), ops.ptr(xd), ops.ptr(dw), ops.ptr(db), ops.ptr(work), B, H, W, Cin, Cout,
                           ops._stream())
        torch.cuda.synchronize()
        outs.append((dw.cpu(), db.cpu()))
    (dw, db), (dw2, db2) = outs
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    sw, sb = float(w.grad.abs().max()), float(b.grad.abs().max())
    assert (dw.double() - w.grad).abs().max() <= 2e-5 * max(1.0, sw) * (B * H * W) ** 0.5
    assert (db.double() - b.grad).abs().max() <= 2e-5 * max(1.0, sb) * (B * H * W) ** 0.5


@pytest.mark.parametrize("case", [(2, 64, 64, 256, 256), (4, 32, 32, 128, 256), (2, 64, 32, 256, 256), (3, 32, 64, 200, 232)],
                         ids=str)
def test_conv_with_batchnorm_relu_on_load_is_bit_identical(hs, case):
    """hipseg_conv3_bnrelu_in == hipseg_bn_relu_apply followed by hipseg_conv_igemm, bit for bit (output and statistics
    rows), incl. ragged image borders: the zero padding is of the ACTIVATED tensor (relu(shift) must not leak in)."""
    L, ops = hs.L, hs.ops
    B, C, N, H, W = case
    td, dt = torch.bfloat16, L.BF16
    assert L.conv3_bnrelu_in_applies(dt, C, N, B, H, W)
    raw = to_dev_nhwc(rnd(T("r4.ol.raw", (B, C, H, W), -2, 2), td), td)
    scale = T("r4.ol.sc", (C,), -1.5, 1.5).cuda()
    shift = T("r4.ol.sh", (C,), 0.2, 0.9).cuda()  # positive shifts: relu(shift) != 0 would show at the borders
    w = T("r4.ol.w", (N, C, 3, 3), -0.3, 0.3).cuda()
    bias = T("r4.ol.b", (N,), -0.5, 0.5).cuda()
    wp = ops._pack_conv(w, dt, False)
    s = ops._stream()
    act = ops.nhwc_empty(B, C, H, W, td, "cuda")
    L.bn_relu_apply(dt, ops.ptr(raw), ops.ptr(scale), ops.ptr(shift), ops.ptr(act), B, H, W, C, 0, s)
    mt = L.conv_mtiles(B, H, W)
    rows = L.conv_stats_rows(dt, L.CONV3, C, 0, N, 0, B, H, W)
    ref, st_ref = ops.nhwc_empty(B, N, H, W, td, "cuda"), torch.zeros(mt, 2, N, device="cuda")
    L.conv_igemm(dt, L.CONV3, ops.ptr(act), C, 0, 0, ops.ptr(wp), ops.ptr(bias), ops.ptr(ref), N, 0, 0, ops.ptr(st_ref), B, H, W, s)
    out, st = ops.nhwc_empty(B, N, H, W, td, "cuda"), torch.zeros(mt, 2, N, device="cuda")
    L.conv3_bnrelu_in(dt, ops.ptr(raw), C, ops.ptr(scale), ops.ptr(shift), ops.ptr(wp), ops.ptr(bias), ops.ptr(out), N, ops.ptr(st),
                      B, H, W, s)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert torch.equal(st[:rows], st_ref[:rows])
    assert float(out.float().abs().max()) > 0


@pytest.mark.parametrize("case", [(2, 64, 64, 64, 64), (1, 128, 64, 32, 48), (16, 64, 64, 128, 128)], ids=str)
def test_wgrad_with_batchnorm_relu_on_load_is_bit_identical(hs, case):
    """hipseg_conv_wgrad_bnrelu_p == hipseg_bn_relu_apply followed by hipseg_conv_wgrad, bit for bit."""
    L, ops = hs.L, hs.ops
    B, CU, CV, H, W = case
    td, dt = torch.bfloat16, L.BF16
    assert L.conv_wgrad_bnrelu_p_applies(dt, CU, CV, B, H, W)
    raw = to_dev_nhwc(rnd(T("r4.ow.raw", (B, CU, H, W), -2, 2), td), td)
    dy = to_dev_nhwc(rnd(T("r4.ow.dy", (B, CV, H, W)), td), td)
    scale = T("r4.ow.sc", (CU,), -1.5, 1.5).cuda()
    shift = T("r4.ow.sh", (CU,), 0.2, 0.9).cuda()
    s = ops._stream()
    act = ops.nhwc_empty(B, CU, H, W, td, "cuda")
    L.bn_relu_apply(dt, ops.ptr(raw), ops.ptr(scale), ops.ptr(shift), ops.ptr(act), B, H, W, CU, 0, s)
    slabs = torch.empty(L.wgrad_workspace_elems(L.CONV3, CU, CV, B, H, W), device="cuda")
    ref = torch.full((CV, CU, 3, 3), float("nan"), device="cuda")
    L.conv_wgrad(dt, L.CONV3, ops.ptr(act), CU, 0, 0, ops.ptr(dy), CV, ops.ptr(ref), ops.ptr(slabs), B, H, W, s)
    got = torch.full((CV, CU, 3, 3), float("nan"), device="cuda")
    L.conv_wgrad_bnrelu_p(dt, ops.ptr(raw), CU, ops.ptr(scale), ops.ptr(shift), ops.ptr(dy), CV, ops.ptr(got), ops.ptr(slabs), B, H,
                          W, s)
    torch.cuda.synchronize()
    assert torch.equal(got, ref) and bool(torch.isfinite(got).all())


def test_convblock_with_batchnorm_on_load_equals_per_op_path(hs):
    """a full-resolution ConvBlockDownsample (32 -> 64 at 2 x 256 x 256: the block call applies the first BatchNorm + ReLU in
    the second convolution's load path and never writes the intermediate) against the per-op launch sequence, once with the
    intermediate materialised and once through the same on-load kernels (what bench.py's profiling leg runs): outputs,
    input gradient, every parameter gradient and the BatchNorm buffers bit for bit."""
    from models.processing_blocks import ConvBlockDownsample

    L, ops = hs.L, hs.ops
    assert L.conv3_bnrelu_in_applies(L.BF16, 64, 64, 2, 256, 256) and L.conv_wgrad_bnrelu_p_applies(L.BF16, 64, 64, 2, 256, 256)
    assert not L.conv_wgrad_pair_applies(L.BF16, 32, 0, 64, 64, 2, 256, 256)
    res = []
    for per_op in (0, 1, 2):
        torch.manual_seed(7)
        m = ConvBlockDownsample(32, 64).cuda().train()
        x = to_dev_nhwc(T("r4.blkol", (2, 32, 256, 256)), torch.bfloat16).requires_grad_(True)
        old = ops._NO_BLOCK_CALLS, ops._PEROP_ON_LOAD
        ops._NO_BLOCK_CALLS, ops._PEROP_ON_LOAD = bool(per_op), per_op != 1  # 1: materialising per-op path, 2: on-load per-op path
        try:
            with torch.autocast("cuda"):
                y = m(x)
            y.float().square().mean().backward()
        finally:
            ops._NO_BLOCK_CALLS, ops._PEROP_ON_LOAD = old
        torch.cuda.synchronize()
        res.append([y.detach(), x.grad] + [p.grad.clone() for p in m.parameters()] + [b.clone() for b in m.buffers()])
    for other in res[1:]:
        for a, b in zip(res[0], other):
            assert torch.equal(a, b)


def test_hip_adam_through_gradscaler_with_its_own_inf_check(hs):
    """`scaler.step(opt)` / `scaler.update()` exactly as models/model_wrappers.py:175-177 drives them, hipseg.optim.Adam (which
    takes GradScaler's `grad_scaler=` hand-over and runs the inf check itself: hipseg_grads_nonfinite) against
    torch.optim.Adam under its own GradScaler: an inf and a NaN gradient on two of the steps must skip those steps, back
    the scale off identically, and leave both parameter sets in step; also after scaler.unscale_(opt), and replayed as a
    hipGraph."""
    from hipseg.optim import Adam

    torch.manual_seed(3)
    shapes = [(64, 32, 3, 3), (515,), (7,), (128, 64, 2, 2)]
    p_ref = [torch.randn(s, device="cuda").requires_grad_(True) for s in shapes]
    p_hip = [p.detach().clone().requires_grad_(True) for p in p_ref]
    kw = dict(lr=1e-2, weight_decay=1e-3)
    o_ref, o_hip = torch.optim.Adam(p_ref, **kw), Adam(p_hip, **kw)
    s_ref = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10, growth_interval=3)
    s_hip = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10, growth_interval=3)
    for sc in (s_ref, s_hip):
        sc.scale(torch.zeros(1, device="cuda"))  # (creates the scale tensors, as scaler.scale(loss) does in a real step)
    for it in range(9):
        for ps, sc in ((p_ref, s_ref), (p_hip, s_hip)):
            g = torch.Generator(device="cuda").manual_seed(100 + it)
            for k, p in enumerate(ps):
                grad = torch.randn(p.shape, device="cuda", generator=g) * float(sc.get_scale())  # "scaled" gradients
                if it == 2 and k == 1:
                    grad[17] = float("inf")
                if it == 5 and k == 3:
                    grad[3, 2, 1, 0] = float("nan")
                p.grad = grad
        if it == 7:  # the explicit-unscale order of use (gradient clipping etc.)
            s_ref.unscale_(o_ref)
            s_hip.unscale_(o_hip)
        s_ref.step(o_ref)
        s_hip.step(o_hip)
        s_ref.update()
        s_hip.update()
        assert float(s_ref.get_scale()) == float(s_hip.get_scale()), it
        for a, b in zip(p_ref, p_hip):
            assert torch.isfinite(b).all()
            np.testing.assert_allclose(b.detach().cpu().numpy(), a.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
    assert o_hip.step_count() == 7  # two of the nine steps were skipped
    # replayed as a hipGraph (static gradient buffers): the inf check, the step and the scale update are all captured
    for p in p_hip:
        p.grad = torch.zeros_like(p)
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        s_hip.step(o_hip)
        s_hip.update()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            s_hip.step(o_hip)
            s_hip.update()
        before = o_hip.step_count()
        scale0 = float(s_hip.get_scale())
        p_hip[0].grad.fill_(float("inf"))
        graph.replay()
        torch.cuda.synchronize()
        assert o_hip.step_count() == before and float(s_hip.get_scale()) == scale0 * 0.5
        p_hip[0].grad.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert o_hip.step_count() == before + 1
    torch.cuda.current_stream().wait_stream(st)


def test_bucket_allreduce_through_a_raw_rccl_communicator(hs):
    """hipseg_bucket_allreduce (SURVEY 8b's bucket_allreduce(ptr, count, dtype, comm, stream)) with a communicator the HOST
    owns: a one-rank RCCL communicator created through RCCL's C API (ctypes), as a non-Python host would; the average
    over one rank is the identity, the call must run on the given stream and leave the bucket unchanged.  The two enum
    values the entry point hard-codes (ncclFloat32 = 7, ncclAvg = 4) are checked against rccl.h where the header exists."""
    import ctypes
    import os
    import re

    hdr = "/opt/rocm/include/rccl/rccl.h"
    if os.path.exists(hdr):
        txt = open(hdr).read()
        assert re.search(r"ncclFloat32\s*=\s*7\b", txt) and re.search(r"ncclAvg\s*=\s*4\b", txt)

    L, ops = hs.L, hs.ops
    rccl = None
    for name in ("librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"):
        try:
            rccl = ctypes.CDLL(name)
            break
        except OSError:
            continue
    if rccl is None:
        import glob, os
        cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))
        assert cands, "no RCCL library found"
        rccl = ctypes.CDLL(cands[0])

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        b = torch.randn(100003, device="cuda")
        want = b.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            L.bucket_allreduce(ops.ptr(b), b.numel(), L.F32, comm.value, s.cuda_stream)
        s.synchronize()
        assert torch.equal(b, want)
        with pytest.raises(Exception, match="bucket_allreduce"):
            L.bucket_allreduce(ops.ptr(b), b.numel(), L.BF16, comm.value, s.cuda_stream)
        with pytest.raises(Exception, match="bucket_allreduce"):
            L.bucket_allreduce(ops.ptr(b), b.numel(), L.F32, 0, s.cuda_stream)
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.parametrize("prec,td,dt", [("fp32", torch.float32, 0), ("bf16", torch.bfloat16, 1)])
@pytest.mark.parametrize("case", [(2, 32, 3, 64, 64), (3, 32, 3, 40, 56), (1, 64, 5, 24, 24), (16, 32, 3, 128, 128)], ids=str)
def test_head_with_batchnorm_relu_on_load(hs, prec, td, dt, case):
    """hipseg_head_fwd_bnrelu / hipseg_head_bwd_bnrelu == hipseg_bn_relu_apply followed by hipseg_head_fwd / hipseg_head_bwd,
    bit for bit (logits, dX, dW, db), and the BatchNorm-backward rows the backward leaves behind finalize to the sums
    hipseg_bn_bwd_reduce computes from the stored dX (another partition of the same fp32 sum: compared to rounding)."""
    L, ops = hs.L, hs.ops
    B, C, K, H, W = case
    raw = to_dev_nhwc(rnd(T("r4.hd.raw", (B, C, H, W), -2, 2), td), td)
    mean = T("r4.hd.mean", (C,), -0.3, 0.3).cuda()
    invstd = T("r4.hd.is", (C,), 0.5, 1.5).cuda()
    gamma = T("r4.hd.g", (C,), -1.2, 1.2).cuda()
    beta = T("r4.hd.b", (C,), -0.4, 0.4).cuda()
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    w = T("r4.hd.w", (K, C, 1, 1), -0.5, 0.5).cuda()
    b = T("r4.hd.bias", (K,), -0.5, 0.5).cuda()
    dl = T("r4.hd.dl", (B, K, H, W), -1, 1).cuda()
    s = ops._stream()
    p = ops.ptr
    # reference: two kernels forward, three backward
    act = ops.nhwc_empty(B, C, H, W, td, "cuda")
    L.bn_relu_apply(dt, p(raw), p(scale), p(shift), p(act), B, H, W, C, 0, s)
    lg_ref = torch.empty(B, K, H, W, device="cuda")
    L.head_fwd(dt, p(act), p(w), p(b), p(lg_ref), B, H, W, C, K, s)
    nb = L.head_bwd_blocks(B, H, W)
    part = torch.zeros(nb * K * (C + 1), device="cuda")
    dx_ref, dw_ref, db_ref = ops.nhwc_empty(B, C, H, W, td, "cuda"), torch.zeros(K, C, device="cuda"), torch.zeros(K, device="cuda")
    L.head_bwd(dt, p(act), p(dl), p(w), p(dx_ref), p(part), p(dw_ref), p(db_ref), B, H, W, C, K, s)
    nred = L.bn_bwd_blocks(B, H, W, C, dt, 0)
    rows_ref = torch.zeros(nred, 2, C, device="cuda")
    L.bn_bwd_reduce2(dt, p(dx_ref), 0, p(raw), p(mean), p(invstd), p(scale), p(shift), p(rows_ref), B, H, W, C, 0, s)
    sums_ref = torch.zeros(2 * C, device="cuda")
    L.colsum_finalize(p(rows_ref), nred, 2, C, p(sums_ref), 0, s)
    # on load
    lg = torch.empty(B, K, H, W, device="cuda")
    L.head_fwd_bnrelu(dt, p(raw), p(scale), p(shift), p(w), p(b), p(lg), B, H, W, C, K, s)
    part2 = torch.zeros(nb * K * (C + 1), device="cuda")
    dx, dw, db = ops.nhwc_empty(B, C, H, W, td, "cuda"), torch.zeros(K, C, device="cuda"), torch.zeros(K, device="cuda")
    rows = torch.full((nb, 2, C), float("nan"), device="cuda")
    L.head_bwd_bnrelu(dt, p(raw), p(mean), p(invstd), p(scale), p(shift), p(dl), p(w), p(dx), p(part2), p(dw), p(db), p(rows),
                      B, H, W, C, K, s)
    sums = torch.zeros(2 * C, device="cuda")
    L.colsum_finalize(p(rows), nb, 2, C, p(sums), 0, s)
    torch.cuda.synchronize()
    assert torch.equal(lg, lg_ref) and torch.equal(dx, dx_ref) and torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
    assert float(lg.abs().max()) > 0 and float(dx.float().abs().max()) > 0
    assert torch.isfinite(rows).all()
    err = (sums.double() - sums_ref.double()).abs().max()
    assert err <= 2e-5 * max(1.0, float(sums_ref.abs().max())) * (B * H * W) ** 0.5 * 0.05 + 1e-4, float(err)
    assert float(sums_ref.abs().max()) > 0


def _unet_step_with_head(hs, fuse, prec, per_op=False, train=True):
    import models.UNet as un
    from models.losses import HybridLoss

    ops = hs.ops
    torch.manual_seed(11)
    m = un.UNet().cuda().train(train)
    g = torch.Generator().manual_seed(13)
    x = torch.rand(2, 3, 32, 48, generator=g).cuda()
    t = torch.randint(0, 3, (2, 32, 48), generator=g).cuda()
    old = ops._NO_HEAD_FUSE, ops._NO_BLOCK_CALLS
    ops._NO_HEAD_FUSE, ops._NO_BLOCK_CALLS = not fuse, per_op
    try:
        with ops.precision_mode(prec):
            out = m(x)
            loss = HybridLoss()(out, t)
        loss.backward()
    finally:
        ops._NO_HEAD_FUSE, ops._NO_BLOCK_CALLS = old
    torch.cuda.synchronize()
    return out.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}, {k: b.clone() for k, b in m.named_buffers()}


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_unet_last_block_and_head_as_one_node(hs, prec):
    """models/UNet.py:72-73 (dec4 -> out): with the last block's BatchNorm + ReLU applied in the head's load path and that
    layer's BatchNorm-backward sums reduced by the head's backward, the logits, the head's gradients and the BatchNorm
    buffers are bit-identical to the separate launches; every other gradient sits downstream of the re-partitioned fp32
    sums and is compared to rounding (fp32) / to the storage precision (bf16)."""
    o0, g0, b0 = _unet_step_with_head(hs, False, prec)
    o1, g1, b1 = _unet_step_with_head(hs, True, prec)
    assert torch.equal(o0, o1)
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k
    for k in ("out.weight", "out.bias"):
        assert torch.equal(g0[k], g1[k]), k
    rel = 2e-5 if prec == "fp32" else 2e-2
    for k in g0:
        d = float((g0[k].double() - g1[k].double()).norm())
        n = float(g0[k].double().norm())
        assert d <= rel * max(n, 1e-6), (k, d, n)
    # the fused form through the per-op launch sequence (bench.py's profiling path) == through the block call
    o2, g2, _ = _unet_step_with_head(hs, True, prec, per_op=True)
    assert torch.equal(o1, o2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k


def test_unet_head_node_in_eval_and_no_grad_modes(hs):
    """outside training the node runs block and head one after the other (no pass to save): the eval-mode step matches
    the two-node form bit for bit, with and without a graph"""
    o0, g0, _ = _unet_step_with_head(hs, False, "bf16", train=False)
    o1, g1, _ = _unet_step_with_head(hs, True, "bf16", train=False)
    assert torch.equal(o0, o1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    import models.UNet as un

    torch.manual_seed(11)
    m = un.UNet().cuda().eval()
    x = torch.rand(2, 3, 32, 48).cuda()
    ops = hs.ops
    res = []
    for off in (True, False):
        old = ops._NO_HEAD_FUSE
        ops._NO_HEAD_FUSE = off
        try:
            with torch.no_grad(), torch.autocast("cuda"):
                res.append(m(x))
        finally:
            ops._NO_HEAD_FUSE = old
    with torch.autocast("cuda"):
        b = m(x)
    a = res[1]
    assert torch.equal(res[0], a) and a.dtype == torch.float32 and a.shape == (2, 3, 32, 48)
    # (without a graph the eval-mode blocks run as fused conv + folded-BatchNorm + ReLU kernels: same values to bf16 rounding)
    assert (a - b.detach()).abs().max() <= 2e-2 * max(1.0, float(a.abs().max()))


CONVT_DGRAD_BWS_CASES = [  # B, Cout (channels of dy), Cin (channels of dx), H, W of dx
    (4, 32, 64, 32, 32),     # streaming kernel, one channel set per wave (dec4.up of the U-Nets)
    (2, 32, 128, 16, 48),    # streaming kernel, two channel sets
    (2, 32, 256, 16, 16),    # streaming kernel, four channel sets
    (2, 64, 128, 32, 32),    # one-tap GEMM kernel (dec3.up)
    (2, 128, 256, 16, 16),   # dec2.up
    (1, 256, 512, 16, 32),   # dec1.up, four column tiles
    (2, 64, 128, 24, 40),    # ragged pixel tiles
]


@pytest.mark.parametrize("case", CONVT_DGRAD_BWS_CASES, ids=str)
def test_convT_data_gradient_with_batchnorm_backward_sums(hs, case):
    """hipseg_convT_dgrad_bnstats == hipseg_conv_igemm(CONV2S2) bit for bit, and its rows finalize to the sums
    hipseg_bn_bwd_reduce computes from the stored dx (another partition of the same fp32 sum)."""
    L, ops = hs.L, hs.ops
    B, CO, CI, H, W = case
    td, dt = torch.bfloat16, L.BF16
    rows = L.convT_dgrad_bnstats_rows(dt, CO, CI, B, H, W)
    assert rows > 0
    dy = to_dev_nhwc(rnd(T("r4.td.dy", (B, CO, 2 * H, 2 * W), -1, 1), td), td)
    w = T("r4.td.w", (CI, CO, 2, 2), -0.4, 0.4).cuda()
    raw = to_dev_nhwc(rnd(T("r4.td.raw", (B, CI, H, W), -2, 2), td), td)
    mean = T("r4.td.mean", (CI,), -0.3, 0.3)
    invstd = T("r4.td.is", (CI,), 0.5, 1.5)
    gamma = T("r4.td.g", (CI,), -1.2, 1.2)
    beta = T("r4.td.b", (CI,), -0.4, 0.4)
    scale = gamma * invstd
    bn = torch.cat([mean, invstd, scale, beta - mean * scale]).cuda().contiguous()
    wpt = ops._pack_convT(w, dt, True)
    s, p = ops._stream(), ops.ptr
    ref = ops.nhwc_empty(B, CI, H, W, td, "cuda")
    L.conv_igemm(dt, L.CONV2S2, p(dy), CO, 0, 0, p(wpt), 0, p(ref), CI, 0, 0, 0, B, H, W, s)
    nred = L.bn_bwd_blocks(B, H, W, CI, dt, 0)
    rows_ref = torch.zeros(nred, 2, CI, device="cuda")
    L.bn_bwd_reduce2(dt, p(ref), 0, p(raw), p(bn), p(bn[CI:]), p(bn[2 * CI:]), p(bn[3 * CI:]), p(rows_ref), B, H, W, CI, 0, s)
    sums_ref = torch.zeros(2 * CI, device="cuda")
    L.colsum_finalize(p(rows_ref), nred, 2, CI, p(sums_ref), 0, s)
    dx = ops.nhwc_empty(B, CI, H, W, td, "cuda")
    part = torch.full((rows, 2, CI), float("nan"), device="cuda")
    L.convT_dgrad_bnstats(dt, p(dy), CO, p(wpt), p(dx), CI, p(raw), p(bn), p(part), B, H, W, s)
    sums = torch.zeros(2 * CI, device="cuda")
    L.colsum_finalize(p(part), rows, 2, CI, p(sums), 0, s)
    torch.cuda.synchronize()
    assert torch.equal(dx, ref) and float(dx.float().abs().max()) > 0
    assert torch.isfinite(part).all()
    err = (sums.double() - sums_ref.double()).abs().max()
    assert err <= 1e-6 * max(1.0, float(sums_ref.abs().max())) * (B * H * W) ** 0.5 + 1e-4, float(err)
    assert float(sums_ref.abs().max()) > 0


def test_convT_data_gradient_with_sums_refuses_other_shapes(hs):
    L = hs.L
    assert L.convT_dgrad_bnstats_rows(L.BF16, 32, 96, 2, 16, 16) == 0     # three channel sets: not a divisor of 4 waves
    assert L.convT_dgrad_bnstats_rows(L.BF16, 64, 64, 2, 16, 16) == 0     # GEMM kernel needs 128-channel column tiles
    assert L.convT_dgrad_bnstats_rows(L.F32, 64, 128, 2, 16, 16) == 0     # bf16 kernels only
    x = torch.zeros(8, device="cuda")
    with pytest.raises(L.HipsegError):
        L.convT_dgrad_bnstats(L.BF16, x.data_ptr(), 64, x.data_ptr(), x.data_ptr(), 64, x.data_ptr(), x.data_ptr(), x.data_ptr(),
                              2, 16, 16, hs.ops._stream())


def _unet_step_with_up(hs, fuse, prec, per_op=False, model="UNet"):
    import models.UNet as un
    from models.losses import HybridLoss

    ops = hs.ops
    torch.manual_seed(17)
    m = getattr(un, model)().cuda().train()
    g = torch.Generator().manual_seed(19)
    hw = (64, 96) if model == "UNet" else (64, 64)
    x = torch.rand(2, 3, *hw, generator=g).cuda()
    t = torch.randint(0, 3, (2, *hw), generator=g).cuda()
    old = ops._NO_UP_FUSE, ops._NO_BLOCK_CALLS
    ops._NO_UP_FUSE, ops._NO_BLOCK_CALLS = not fuse, per_op
    try:
        with ops.precision_mode(prec):
            out = m(x)
            loss = HybridLoss()(out, t)
        loss.backward()
    finally:
        ops._NO_UP_FUSE, ops._NO_BLOCK_CALLS = old
    torch.cuda.synchronize()
    return out.detach(), {k: p.grad.clone() for k, p in m.named_parameters()}, {k: b.clone() for k, b in m.named_buffers()}


@pytest.mark.parametrize("prec,model", [("bf16", "UNet"), ("fp32", "UNet"), ("bf16", "LargeUNet")])
def test_unet_block_and_following_convtranspose_as_one_node(hs, prec, model):
    """models/UNet.py:66-71: every decoder ConvBlock (and the bottleneck) feeds exactly one ConvTranspose2d.  As one autograd
    node the forward launches are the same (logits and BatchNorm buffers bit-identical); in backward the ConvTranspose2d's
    data gradient also reduces the block's BatchNorm-backward sums, so the ConvTranspose2d's own gradients stay bit-identical
    and everything behind the re-partitioned fp32 sums is compared to rounding / the storage precision."""
    o0, g0, b0 = _unet_step_with_up(hs, False, prec, model=model)
    o1, g1, b1 = _unet_step_with_up(hs, True, prec, model=model)
    assert torch.equal(o0, o1)
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k
    last = max(int(k[3]) for k in g0 if k.startswith("dec"))
    for k in (f"dec{last}.up.weight", f"dec{last}.up.bias", "out.weight", "out.bias"):
        assert torch.equal(g0[k], g1[k]), k
    rel = 2e-5 if prec == "fp32" else 2e-2
    for k in g0:
        d = float((g0[k].double() - g1[k].double()).norm())
        n = float(g0[k].double().norm())
        if k.endswith(".bias") and k[:-4] + "weight" in g0:
            # (a bias in front of a BatchNorm has a gradient that is a sum of cancelling terms -- exactly zero in exact
            # arithmetic for the conv biases, border effects only for the stem's: scale by the module's weight gradient)
            n = max(n, float(g0[k[:-4] + "weight"].double().norm()))
        assert d <= rel * max(n, 1e-6), (k, d, n)
    o2, g2, _ = _unet_step_with_up(hs, True, prec, per_op=True, model=model)
    assert torch.equal(o1, o2)
    for k in g1:
        assert torch.equal(g1[k], g2[k]), k


def test_unet_with_a_forward_hook_takes_the_module_by_module_path(hs):
    """the fused block + consumer nodes bypass Module.__call__ of the decoder blocks; a model with a hook registered there
    must still see it fire, with the same logits"""
    import models.UNet as un

    torch.manual_seed(23)
    m = un.UNet().cuda().train()
    x = torch.rand(2, 3, 32, 32, device="cuda")
    with torch.autocast("cuda"):
        ref = m(x).detach()
    seen = []
    h = m.dec2.register_forward_hook(lambda mod, inp, out: seen.append(tuple(out.shape)))
    try:
        torch.manual_seed(23)
        m2 = un.UNet().cuda().train()
        h2 = m2.dec2.register_forward_hook(lambda mod, inp, out: seen.append(tuple(out.shape)))
        with torch.autocast("cuda"):
            got = m2(x).detach()
        h2.remove()
    finally:
        h.remove()
    assert seen == [(2, 128, 8, 8)]
    assert torch.equal(ref, got)


def test_frozen_clip_tower_shadow_and_single_norm_cast_keep_autocast_values(hs):
    """the frozen CLIP image tower under autocast (reference: processing_blocks.py:173-233 runs it inside the step's
    autocast region): pre-cast weights (`_lowp_shadow`) and one cast per branch LayerNorm give bit-identical features to
    torch's own per-call casts, and leave the module as they found it (fp32 parameters, class forward)."""
    import os

    from transformers import CLIPConfig, CLIPModel

    from models.processing_blocks import ClipFeatureExtractor

    torch.manual_seed(31)
    cfg = CLIPConfig(vision_config=dict(hidden_size=64, intermediate_size=128, num_hidden_layers=3, num_attention_heads=4,
                                        image_size=224, patch_size=32),
                     text_config=dict(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2),
                     projection_dim=32)
    ext = ClipFeatureExtractor(clip_model=CLIPModel(cfg)).cuda().eval()
    x = torch.rand(3, 3, 224, 224, device="cuda")
    keys = list(ext.state_dict())
    res = {}
    for name, env in (("torch", {"HIPSEG_NO_CLIP_SHADOW": "1"}), ("shadow", {"HIPSEG_NO_CLIP_NORM_CAST": "1"}), ("both", {})):
        old = {k: os.environ.pop(k, None) for k in ("HIPSEG_NO_CLIP_SHADOW", "HIPSEG_NO_CLIP_NORM_CAST")}
        os.environ.update(env)
        try:
            with torch.autocast("cuda"):
                res[name] = ext(x).float().clone()
        finally:
            for k in env:
                os.environ.pop(k, None)
            os.environ.update({k: v for k, v in old.items() if v is not None})
    assert torch.equal(res["torch"], res["shadow"]) and torch.equal(res["torch"], res["both"])
    assert float(res["both"].abs().max()) > 0 and res["both"].shape == (3, 32)
    assert len(ext._branch_norms()) == 6
    assert list(ext.state_dict()) == keys and all(p.dtype == torch.float32 for p in ext.clip_model.parameters())
    assert all("forward" not in m.__dict__ for m in ext.clip_model.modules())
    # outside autocast nothing is swapped: fp32 features
    assert ext(x).dtype == torch.float32

#!/usr/bin/env python3
"""Does the weight-gradient kernel overlap with the data-gradient + BatchNorm-backward chain when it runs on a second
stream?  Per layer shape: serial (one stream) vs forked (wgrad on a side stream, joined at the end), both as hipGraphs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import torch
import hipseg
from hipseg import _lib as L, ops

dt, td = L.BF16, torch.bfloat16
B = 16
side = torch.cuda.Stream()
for C, H in [(64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)]:
    dev = "cuda"
    a1 = ops.nhwc_empty(B, C, H, H, td, dev).normal_()
    draw2 = ops.nhwc_empty(B, C, H, H, td, dev).normal_()
    raw1 = ops.nhwc_empty(B, C, H, H, td, dev).normal_()
    da1 = ops.nhwc_empty(B, C, H, H, td, dev)
    draw1 = ops.nhwc_empty(B, C, H, H, td, dev)
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    wpt = ops._pack_conv(w, dt, True)
    dw = torch.empty(C, C, 3, 3, device=dev)
    slabs = ops._f32(L.wgrad_workspace_elems(L.CONV3, C, C, B, H, H), dev)
    bn = torch.rand(4 * C, device=dev) + 0.5
    nblk = L.bn_bwd_blocks(B, H, H, C, dt, 0)
    partial = ops._f32(nblk * 2 * C, dev)
    sums = ops._f32(2 * C, dev)
    P = ops.ptr

    def wgrad(s):
        L.conv_wgrad(dt, L.CONV3, P(a1), C, 0, 0, P(draw2), C, P(dw), P(slabs), B, H, H, s)

    def chain(s):
        L.conv_igemm(dt, L.CONV3, P(draw2), C, 0, 0, P(wpt), 0, P(da1), C, 0, 0, 0, B, H, H, s)
        bp = bn.data_ptr()
        L.bn_bwd_reduce(dt, P(da1), P(raw1), bp, bp + 4 * C, bp + 8 * C, bp + 12 * C, P(partial), B, H, H, C, 0, s)
        L.colsum_finalize(P(partial), nblk, 2, C, P(sums), 0, s)
        L.bn_bwd_apply(dt, P(da1), P(raw1), bp, bp + 4 * C, bp + 8 * C, bp + 12 * C, P(sums), float(B * H * H), 0, P(draw1), 0,
                       B, H, H, C, 0, s)

    def serial():
        s = torch.cuda.current_stream().cuda_stream
        wgrad(s)
        chain(s)

    def forked():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        wgrad(side.cuda_stream)
        chain(cur.cuda_stream)
        cur.wait_stream(side)

    res = {}
    use_graph = bool(os.environ.get("OVERLAP_GRAPH"))
    for name, fn in (("serial", serial), ("forked", forked), ("wgrad", lambda: wgrad(torch.cuda.current_stream().cuda_stream)),
                     ("chain", lambda: chain(torch.cuda.current_stream().cuda_stream))):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            if use_graph:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    for _ in range(4):
                        fn()
                run, per = g.replay, 4
            else:
                run, per = fn, 1
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200 // per):
                run()
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / 200 * 1e3
    print(f"overlap C={C:4d} H={H:3d}: wgrad {res['wgrad']:6.1f}  chain {res['chain']:6.1f}  serial {res['serial']:6.1f}  "
          f"forked {res['forked']:6.1f} us  (saved {res['serial'] - res['forked']:5.1f})", flush=True)

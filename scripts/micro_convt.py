#!/usr/bin/env python3
"""Micro-benchmark of the ConvTranspose2d(k2,s2) stages through the C ABI (forward and data gradient).
env HIPSEG_NO_CONVT_STREAM=1 for the GEMM kernels"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import torch
import hipseg
from hipseg import _lib as L, ops

LAYERS = [("dec1.up", 16, 512, 256, 32), ("dec2.up", 16, 256, 128, 32), ("dec3.up", 16, 128, 64, 64), ("dec4.up", 16, 64, 32, 128)]
dt, td = L.BF16, torch.bfloat16
for name, B, ci, co, H in LAYERS:
    x = ops.nhwc_empty(B, ci, H, H, td, "cuda").normal_()
    dy = ops.nhwc_empty(B, co, 2 * H, 2 * H, td, "cuda").normal_()
    w = torch.randn(ci, co, 2, 2, device="cuda") * 0.05
    b = torch.zeros(co, device="cuda")
    wp, wpt = ops._pack_convT(w, dt, False), ops._pack_convT(w, dt, True)
    y = ops.nhwc_empty(B, co, 2 * H, 2 * H, td, "cuda")
    dx = ops.nhwc_empty(B, ci, H, H, td, "cuda")
    dw = torch.empty(ci, co, 2, 2, device="cuda")
    slabs = ops._f32(L.wgrad_workspace_elems(L.CONVT, co, ci, B, H, H), "cuda")
    s = torch.cuda.current_stream().cuda_stream
    db = torch.empty(co, device="cuda")
    work = ops._f32(L.convT_wgrad_workspace_elems(ci, co, B, H, H), "cuda")
    part = ops._f32(L.colsum_blocks(B * 4 * H * H, co, dt) * co, "cuda")
    fns = {"fwd": lambda: ops.igemm(dt, L.CONVT, x, ci, None, 0, wp, b, y, co, None, 0, None, B, H, H),
           "dgrad": lambda: ops.igemm(dt, L.CONV2S2, dy, co, None, 0, wpt, None, dx, ci, None, 0, None, B, H, H),
           "wgrad+bias (fused, round 4)": lambda: L.convT_wgrad_bias(dt, ops.ptr(dy), ops.ptr(x), ops.ptr(dw), ops.ptr(db), ops.ptr(work), B, H, H, ci, co, s),
           "colsum(bias)": lambda: L.colsum(dt, ops.ptr(dy), B * 4 * H * H, co, ops.ptr(part), ops.ptr(db), s),
           "wgrad": lambda: L.conv_wgrad(dt, L.CONVT, ops.ptr(dy), co, 0, 0, ops.ptr(x), ci, ops.ptr(dw), ops.ptr(slabs), B, H, H, s)}
    for which, fn in fns.items():
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        mb = (x.numel() + y.numel()) * 2 / 1e6
        print(f"convT {which:28s} {name:8s} {ms*1e3:8.1f} us  {mb/ms/1e3:6.2f} TB/s of activations  {2.0*B*H*H*ci*co*4/ms/1e9:8.1f} TF/s", flush=True)

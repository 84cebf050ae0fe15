#!/usr/bin/env python3
"""Micro-benchmark of single conv layers through the C ABI (HIP-event timing).
usage: micro_conv.py [igemm|wgrad] ; env HIPSEG_IGEMM_DEBUG for ablations"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import torch
import hipseg
from hipseg import _lib as L, ops

LAYERS = [  # name, B, Cin, Cout, H
    ("enc1.c1", 16, 64, 64, 256), ("enc2.c0", 16, 64, 128, 128), ("enc2.c1", 16, 128, 128, 128),
    ("enc3.c0", 16, 128, 256, 64), ("enc3.c1", 16, 256, 256, 64), ("bott.c0", 16, 256, 512, 32),
    ("bott.c1", 16, 512, 512, 32), ("dec4.c1", 16, 32, 32, 256),
    ("dec1.c0", 16, 512, 256, 32), ("dec1.c1", 16, 256, 256, 32), ("dec2.c0", 16, 256, 128, 64),
    ("dec2.c1", 16, 128, 128, 64),
    # data gradients whose shape differs from a forward layer's (K = Cout, N = Cin)
    ("enc3.c0^T", 16, 256, 128, 64), ("bott.c0^T", 16, 512, 256, 32), ("dec1.c0^T", 16, 256, 512, 32),
    ("dec2.c0^T", 16, 128, 256, 64),
]
which = sys.argv[1] if len(sys.argv) > 1 else "igemm"
only = os.environ.get("MICRO_LAYERS")
if only:
    LAYERS = [l for l in LAYERS if l[0] in only.split(",")]
REPS = int(os.environ.get("MICRO_REPS", "20"))
dt, td = L.BF16, torch.bfloat16
for name, B, ci, co, H in LAYERS:
    x = ops.nhwc_empty(B, ci, H, H, td, "cuda").normal_()
    dy = ops.nhwc_empty(B, co, H, H, td, "cuda").normal_()
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wp = ops._pack_conv(w, dt, False)
    out = ops.nhwc_empty(B, co, H, H, td, "cuda")
    stats = torch.empty(L.conv_mtiles(B, H, H) * 2 * co, device="cuda")
    dw = torch.empty_like(w)
    if which == "igemm":
        fn = lambda: ops.igemm(dt, L.CONV3, x, ci, None, 0, wp, None, out, co, None, 0, stats, B, H, H)
    else:
        fn = lambda: ops._wgrad(dt, L.CONV3, x, None, dy, dw, B, H, H)
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = REPS
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * H * H * ci * co * 9
    dbg = os.environ.get("HIPSEG_IGEMM_DEBUG", "0") + "/" + os.environ.get("HIPSEG_WGRAD_DEBUG", "0")
    print(f"{which} {name:8s} {ms*1e3:8.1f} us  {fl/ms/1e9:8.1f} TF/s  dbg={dbg}", flush=True)

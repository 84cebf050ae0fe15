#!/usr/bin/env python3
"""micro-benchmark: the two 64-output-channel 3x3 layers of the U-Net step (dec3.c0 forward, enc2.c0 data gradient);
HIPSEG_NO_M16_BN64=1 sends them to the ring kernel (A/B)"""

import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "image-segmentation_amd")]
import torch, hipseg
from hipseg import _lib as L, ops
dt, td = L.BF16, torch.bfloat16
for name, B, c0, c1, co, H in [("dec3.c0", 16, 64, 64, 64, 128), ("enc2.c0^T", 16, 128, 0, 64, 128)]:
    x0 = ops.nhwc_empty(B, c0, H, H, td, "cuda").normal_(); x1 = ops.nhwc_empty(B, c1, H, H, td, "cuda").normal_() if c1 else None
    w = torch.randn(co, c0 + c1, 3, 3, device="cuda") * 0.05
    wp = ops._pack_conv(w, dt, False)
    out = ops.nhwc_empty(B, co, H, H, td, "cuda")
    stats = torch.empty(L.conv_mtiles(B, H, H) * 2 * co, device="cuda")
    fn = lambda: ops.igemm(dt, L.CONV3, x0, c0, x1, c1, wp, None, out, co, None, 0, stats, B, H, H)
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{os.environ.get('HIPSEG_NO_M16_BN64','m16-64'):8s} {name:10s} {ms*1e3:7.1f} us {2.0*B*H*H*(c0+c1)*co*9/ms/1e9:7.1f} TF/s", flush=True)

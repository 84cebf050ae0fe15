#!/usr/bin/env python3
"""Phase cycle sums of the tr16 weight-gradient kernel from a WG_STAMP build (HIPSEG_LIB=libhipseg_wgstamp.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import numpy as np
import torch
import hipseg
from hipseg import _lib as L, ops
lib = ctypes.CDLL(L.LIB_PATH)
LAYERS = [("enc1.c1", 16, 64, 64, 256), ("enc2.c1", 16, 128, 128, 128), ("enc3.c1", 16, 256, 256, 64), ("bott.c1", 16, 512, 512, 32),
          ("dec1.c1", 16, 256, 256, 32)]
dt, td = L.BF16, torch.bfloat16
for name, B, ci, co, H in LAYERS:
    x = ops.nhwc_empty(B, ci, H, H, td, "cuda").normal_()
    dy = ops.nhwc_empty(B, co, H, H, td, "cuda").normal_()
    dw = torch.empty(co, ci, 3, 3, device="cuda")
    fn = lambda: ops._wgrad(dt, L.CONV3, x, None, dy, dw, B, H, H)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    n = 8000
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    nwg = 256
    buf = np.zeros((nwg, 8), dtype=np.uint64)
    assert lib.hipseg_debug_wg_stamps(buf.ctypes.data_as(ctypes.c_void_p), nwg) == 0
    b = buf.astype(np.float64)
    dma, bar, cmp_, nt, tot, rt, r0 = (b[:, i] for i in range(7))
    clk = tot / np.maximum(rt, 1) * 100e6
    print(f"{name}: {ms*1e3:.1f} us/call (incl. reduce) | tiles/WG {np.median(nt):.0f}, clock {np.median(clk)/1e9:.3f} GHz, kernel span {(r0.max()-r0.min()+rt.max())/100:.1f} us")
    print(f"   per tile (cycles, median over WGs): dma-wait {np.median(dma/nt):.0f}  barrier {np.median(bar/nt):.0f}  compute {np.median(cmp_/nt):.0f}  (MFMA 2 x 2304 = 4608)")
    print(f"   whole WG cycles median {np.median(tot):.0f}; tile loop {np.median(dma+bar+cmp_):.0f}; rest (prologue + k-split reduce + slab store) {np.median(tot-dma-bar-cmp_):.0f}")

#!/usr/bin/env python3
"""Phase cycle sums of the weights-stationary 3x3 kernel from a WS_STAMP build (HIPSEG_LIB=libhipseg_wsstamp.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import numpy as np
import torch
import hipseg
from hipseg import _lib as L, ops
lib = ctypes.CDLL(L.LIB_PATH)
dt, td = L.BF16, torch.bfloat16
for name, B, ci, co, H in [("enc1.c1 64->64", 16, 64, 64, 256), ("enc1.c0 32->64", 16, 32, 64, 256), ("dec4.c1 32->32", 16, 32, 32, 256)]:
    x = ops.nhwc_empty(B, ci, H, H, td, "cuda").normal_()
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wp = ops._pack_conv(w, dt, False)
    out = ops.nhwc_empty(B, co, H, H, td, "cuda")
    stats = torch.empty(L.conv_mtiles(B, H, H) * 2 * co, device="cuda")
    fn = lambda: ops.igemm(dt, L.CONV3, x, ci, None, 0, wp, None, out, co, None, 0, stats, B, H, H)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    n = 4000
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    nwg = 512
    buf = np.zeros((nwg, 8), dtype=np.uint64)
    assert lib.hipseg_debug_ws_stamps(buf.ctypes.data_as(ctypes.c_void_p), nwg) == 0
    b = buf.astype(np.float64)
    wait, bar, main, epi, nt, tot, rt, r0 = (b[:, i] for i in range(8))
    clk = tot / np.maximum(rt, 1) * 100e6
    print(f"{name}: {ms*1e3:.1f} us/call | tiles/WG {np.median(nt):.0f}, clock {np.median(clk)/1e9:.3f} GHz, kernel span {(r0.max()-r0.min()+rt.max())/100:.1f} us")
    print(f"   per tile (cycles, median over WGs): piece-wait {np.median(wait/nt):.0f}  barrier {np.median(bar/nt):.0f}  main loop {np.median(main/nt):.0f}  "
          f"epilogue {np.median(epi/nt):.0f}   (MFMA issue per tile and wave: {(ci//16)*9*(2 if co==64 else 1)*32} cycles)")
    print(f"   whole WG cycles median {np.median(tot):.0f}; tile loop {np.median(wait+bar+main+epi):.0f}; rest {np.median(tot-wait-bar-main-epi):.0f}")

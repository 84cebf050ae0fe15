#!/usr/bin/env python3
"""In-kernel clock and phase lengths of the 16x16x32 conv kernel from a STAMP build (HIPSEG_LIB=libhipseg_stamp.so).
usage: micro_stamp.py  (env MICRO_LAYERS, HIPSEG_M16_ROWS)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import numpy as np
import torch
import hipseg
from hipseg import _lib as L, ops
lib = ctypes.CDLL(L.LIB_PATH)
LAYERS = [("enc2.c1", 16, 128, 128, 128), ("enc3.c1", 16, 256, 256, 64), ("bott.c1", 16, 512, 512, 32),
          ("dec1.c0", 16, 512, 256, 32), ("dec2.c1", 16, 128, 128, 64)]
only = os.environ.get("MICRO_LAYERS")
if only:
    LAYERS = [l for l in LAYERS if l[0] in only.split(",")]
dt, td = L.BF16, torch.bfloat16
for name, B, ci, co, H in LAYERS:
    x = ops.nhwc_empty(B, ci, H, H, td, "cuda").normal_()
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wp = ops._pack_conv(w, dt, False)
    out = ops.nhwc_empty(B, co, H, H, td, "cuda")
    stats = torch.empty(L.conv_mtiles(B, H, H) * 2 * co, device="cuda")
    fn = lambda: ops.igemm(dt, L.CONV3, x, ci, None, 0, wp, None, out, co, None, 0, stats, B, H, H)
    fl = 2.0 * B * H * H * ci * co * 9
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    n = int(2.5 / 70e-6)  # >= 2 s of back-to-back launches before the stamped one
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    rows = L.conv_stats_rows(dt, L.CONV3, ci, 0, co, 0, B, H, H)
    nwg = rows * (co // 128)
    buf = np.zeros((nwg, 8), dtype=np.uint64)
    assert lib.hipseg_debug_m16_stamps(buf.ctypes.data_as(ctypes.c_void_p), nwg) == 0
    b = buf.astype(np.int64)
    t0, r0, t2, t3, t4, t5, r6 = (b[:, i] for i in range(7))
    clk = (t5 - t0) / np.maximum(1, (r6 - r0)) * 100e6
    span_us = (r6.max() - r0.min()) / 100.0
    print(f"{name}: {ms*1e3:.1f} us/launch {fl/ms/1e9:.0f} TF/s | {nwg} WGs, kernel span {span_us:.1f} us, "
          f"in-kernel clock median {np.median(clk)/1e9:.3f} GHz (p10 {np.percentile(clk,10)/1e9:.3f}, p90 {np.percentile(clk,90)/1e9:.3f})")
    for nm, d in (("prologue", t2 - t0), ("main", t3 - t2), ("epilogue(issue)", t4 - t3), ("store drain", t5 - t4), ("total", t5 - t0)):
        print(f"   {nm:16s} cycles median {np.median(d):9.0f}  p10 {np.percentile(d,10):9.0f}  p90 {np.percentile(d,90):9.0f}  max {d.max():9.0f}")
    st = (r0 - r0.min()) / 100.0
    en = (r6 - r0.min()) / 100.0
    print(f"   WG start us: median {np.median(st):.1f} p90 {np.percentile(st,90):.1f} max {st.max():.1f};  WG end us: p10 {np.percentile(en,10):.1f} median {np.median(en):.1f} max {en.max():.1f}")
    K = ci
    ideal = (K // 32) * 9 * 2 * 16 * (16 if nwg * 1 >= 0 else 8)
    print(f"   MFMA cycles per wave in main: THT16 {(K//32)*9*2*16*16}  THT8 {(K//32)*9*2*8*16}")

#!/usr/bin/env python3
"""Micro-benchmark of the 1x1 stem / head kernels through the C ABI (bf16, 16 x 3 x 256 x 256)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import torch
import hipseg
from hipseg import _lib as L, ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B, H = 16, 256
x = torch.rand(B, 3, H, H, device="cuda"); w = torch.randn(32, 3, 1, 1, device="cuda"); b = torch.randn(32, device="cuda")
y = ops.nhwc_empty(B, 32, H, H, torch.bfloat16, "cuda"); s = ops._stream()
t = timeit(lambda: L.stem_fwd(L.BF16, ops.ptr(x), ops.ptr(w), ops.ptr(b), ops.ptr(y), B, 3, H, H, 32, s))
print(f"stem_fwd {t:6.1f} us  {(x.numel()*4 + y.numel()*2)/t/1e6:5.2f} TB/s")
wh = torch.randn(3, 32, 1, 1, device="cuda"); bh = torch.randn(3, device="cuda"); lg = torch.empty(B, 3, H, H, device="cuda")
t = timeit(lambda: L.head_fwd(L.BF16, ops.ptr(y), ops.ptr(wh), ops.ptr(bh), ops.ptr(lg), B, H, H, 32, 3, s))
print(f"head_fwd {t:6.1f} us  {(lg.numel()*4 + y.numel()*2)/t/1e6:5.2f} TB/s")
# backward kernels (round 4): stem_bwd2 with both gradients, head_bwd
dy = torch.randn(B, H, H, 32, device="cuda").to(torch.bfloat16); dy2 = torch.randn_like(dy)
part = torch.empty(L.stem_bwd_blocks(B, H, H) * 4 * 32, device="cuda"); dw = torch.empty(32, 3, device="cuda"); db = torch.empty(32, device="cuda")
t = timeit(lambda: L.stem_bwd2(L.BF16, ops.ptr(x), ops.ptr(dy), ops.ptr(dy2), ops.ptr(part), ops.ptr(dw), ops.ptr(db), B, 3, H, H, 32, s))
print(f"stem_bwd2(+finalize) {t:6.1f} us  {(x.numel()*4 + 2*dy.numel()*2)/t/1e6:5.2f} TB/s")
t = timeit(lambda: L.stem_bwd2(L.BF16, ops.ptr(x), ops.ptr(dy), 0, ops.ptr(part), ops.ptr(dw), ops.ptr(db), B, 3, H, H, 32, s))
print(f"stem_bwd (+finalize) {t:6.1f} us  {(x.numel()*4 + dy.numel()*2)/t/1e6:5.2f} TB/s")
dl = torch.randn(B, 3, H, H, device="cuda"); dx = torch.empty_like(dy)
hp = torch.empty(L.head_bwd_blocks(B, H, H) * 3 * 33, device="cuda"); hdw = torch.empty(3, 32, device="cuda"); hdb = torch.empty(3, device="cuda")
t = timeit(lambda: L.head_bwd(L.BF16, ops.ptr(dy), ops.ptr(dl), ops.ptr(wh), ops.ptr(dx), ops.ptr(hp), ops.ptr(hdw), ops.ptr(hdb), B, H, H, 32, 3, s))
print(f"head_bwd (+finalize) {t:6.1f} us  {(dl.numel()*4 + 2*dy.numel()*2)/t/1e6:5.2f} TB/s")

#!/usr/bin/env python3
"""2-read + 1-write streaming reference points: torch.mul(out=) vs bn_bwd_apply, with and without skewed buffer bases."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import torch
import hipseg
from hipseg import _lib as L, ops
dt, td = L.BF16, torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, B, C, H in (("enc1", 16, 64, 256), ("enc3", 16, 256, 64), ("bott", 16, 512, 32)):
    n = B * C * H * H
    for skew in (0, 4096 + 256, 1 << 20):
        pool = torch.empty(3 * n + 3 * skew + 64, dtype=td, device="cuda").normal_()
        a = pool[:n]; b = pool[n + skew:2 * n + skew]; c = pool[2 * n + 2 * skew:3 * n + 2 * skew]
        t0 = timeit(lambda: torch.mul(a, b, out=c))
        A = a.view(B, H, H, C).permute(0, 3, 1, 2); Bt = b.view(B, H, H, C).permute(0, 3, 1, 2); Ct = c.view(B, H, H, C).permute(0, 3, 1, 2)
        bn = ops._BN(C, "cuda")
        for v in (bn.mean, bn.shift): v.zero_()
        for v in (bn.invstd, bn.scale): v.fill_(1.0)
        sums = torch.zeros(2 * C, device="cuda")
        s = ops._stream()
        t1 = timeit(lambda: L.bn_bwd_apply(dt, ops.ptr(A), ops.ptr(Bt), ops.ptr(bn.mean), ops.ptr(bn.invstd), ops.ptr(bn.scale),
                                           ops.ptr(bn.shift), ops.ptr(sums), float(B * H * H), 0, ops.ptr(Ct), 0, B, H, H, C, 0, s))
        print(f"{name} skew {skew:8d} elems: torch.mul {t0:6.1f} us {3*n*2/t0/1e6:5.2f} TB/s | bn_bwd_apply {t1:6.1f} us {3*n*2/t1/1e6:5.2f} TB/s", flush=True)

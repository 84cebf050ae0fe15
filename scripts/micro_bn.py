#!/usr/bin/env python3
"""Micro-benchmark of the BatchNorm streaming kernels through the C ABI: TB/s of algorithmic bytes."""
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
tot = {"relu_apply": 0.0, "bwd_reduce": 0.0, "bwd_apply": 0.0}
for name, B, C, H, pool in (("enc1", 16, 64, 256, 0), ("enc1p", 16, 64, 256, 1), ("enc2", 16, 128, 128, 0), ("enc3", 16, 256, 64, 0),
                            ("bott", 16, 512, 32, 0), ("dec4", 16, 32, 256, 0)):
    raw = ops.nhwc_empty(B, C, H, H, td, "cuda").normal_()
    Ho = H // 2 if pool else H
    dy = ops.nhwc_empty(B, C, Ho, Ho, td, "cuda").normal_()
    act = ops.nhwc_empty(B, C, Ho, Ho, td, "cuda")
    draw = ops.nhwc_empty(B, C, H, H, td, "cuda")
    bn = ops._BN(C, "cuda")
    for v in (bn.mean, bn.shift): v.zero_()
    for v in (bn.invstd, bn.scale): v.fill_(1.0)
    nblk = L.bn_bwd_blocks(B, H, H, C, dt, pool)
    partial = torch.empty(nblk * 2 * C, device="cuda"); sums = torch.zeros(2 * C, device="cuda")
    s = ops._stream()
    nb = raw.numel() * 2; nd = dy.numel() * 2
    t1 = timeit(lambda: L.bn_relu_apply(dt, ops.ptr(raw), ops.ptr(bn.scale), ops.ptr(bn.shift), ops.ptr(act), B, H, H, C, pool, s))
    t2 = timeit(lambda: L.bn_bwd_reduce(dt, ops.ptr(dy), ops.ptr(raw), ops.ptr(bn.mean), ops.ptr(bn.invstd), ops.ptr(bn.scale),
                                        ops.ptr(bn.shift), ops.ptr(partial), B, H, H, C, pool, s))
    t3 = timeit(lambda: L.bn_bwd_apply(dt, ops.ptr(dy), ops.ptr(raw), ops.ptr(bn.mean), ops.ptr(bn.invstd), ops.ptr(bn.scale),
                                       ops.ptr(bn.shift), ops.ptr(sums), float(B * H * H), 0, ops.ptr(draw), 0, B, H, H, C, pool, s))
    print(f"{name:6s} C={C:3d} H={H:3d} pool={pool}: relu_apply {t1:6.1f} us {(nb+nd)/t1/1e6:5.2f} TB/s | bwd_reduce {t2:6.1f} us "
          f"{(nb+nd)/t2/1e6:5.2f} TB/s | bwd_apply {t3:6.1f} us {(2*nb+nd)/t3/1e6:5.2f} TB/s", flush=True)
    tot["relu_apply"] += t1; tot["bwd_reduce"] += t2; tot["bwd_apply"] += t3
print("sum", {k: round(v, 1) for k, v in tot.items()})

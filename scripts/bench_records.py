#!/usr/bin/env python3
"""Measurement of the dataset-record decode (hipseg_decode_records): records/s with inputs resident in HBM, the
HBM roofline fraction (algorithmic bytes = 4 B in + 20 B out per pixel), and the CPU oracle timed beside it.
usage: python scripts/bench_records.py [n_records]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import numpy as np
import torch
import hipseg.data as D
from oracle import records

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
H = W = 256
g = torch.Generator(device="cuda").manual_seed(0)
images = torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
masks = torch.tensor([0, 38, 75, 255], dtype=torch.uint8, device="cuda")[torch.randint(0, 4, (n, H, W), device="cuda", generator=g)]
for _ in range(3):
    D.decode_records(images, masks)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    oi, om = D.decode_records(images, masks)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
nbytes = n * H * W * (4 + 20)
# parity of the timed configuration: a sample of records against the oracle, bit-exact
k = min(n, 8)
wi, wm = records.decode_records(images[:k].cpu().numpy(), masks[:k].cpu().numpy())
assert np.array_equal(oi[:k].cpu().numpy(), wi) and np.array_equal(om[:k].cpu().numpy(), wm)
# CPU baseline: the oracle on a bounded sample
ci, cm = images[:64].cpu().numpy(), masks[:64].cpu().numpy()
t0 = time.perf_counter()
records.decode_records(ci, cm)
cpu_s = time.perf_counter() - t0
print(json.dumps({"metric": "dataset records/s (256x256 record decode)", "value": round(n / (ms * 1e-3), 1), "unit": "records/s",
                  "n_records": n, "ms": round(ms, 4), "dtype": "u8 -> f32/i64",
                  "roofline": {"bound": "hbm", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": 8000.0,
                               "unit": "GB/s", "frac": round(nbytes / (ms * 1e-3) / 8e12, 4), "traffic": None},
                  "cpu_baseline": {"value": round(64 / cpu_s, 1), "unit": "records/s", "cores": 1, "kind": "port",
                                   "sample": "64 records through oracle/records.py (numpy, one thread)"}}))

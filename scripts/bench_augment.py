#!/usr/bin/env python3
"""On-device DataAugmentor (hipseg_augment): time and HBM rate on the bench batch, next to the CPU oracle restatement.
usage: bench_augment.py [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import torch
import hipseg
from models.processing_blocks import DataAugmentor
from oracle import augment as A

B, H = 16, 256
torch.manual_seed(0)
img = torch.rand(B, 3, H, H, device="cuda")
msk = torch.randint(0, 3, (B, H, H), device="cuda")
aug = DataAugmentor(4).cuda()
for _ in range(3):
    aug(img, msk)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n):
    aug(img, msk)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / n * 1e3
naug = B - (B + 4) // 5
# algorithmic bytes: grey-mean pass reads the image of augmented samples; apply pass reads image + mask, writes both
byt = naug * H * H * 12 + B * H * H * (12 + 8) * 2
p, order = aug.last_params
t0 = time.perf_counter()
A.augment(img.cpu(), msk.cpu(), None, p.cpu(), order.cpu().tolist())
cpu_s = time.perf_counter() - t0
out = {"workload": f"DataAugmentor(4) on {B} x 3 x {H} x {H} fp32 images + int64 masks (parameter sampling included)",
       "us_per_call": round(us, 1), "algorithmic_MB": round(byt / 1e6, 1), "GBps": round(byt / us / 1e3, 1),
       "frac_of_8TBps": round(byt / us / 1e3 / 8000, 3), "images_per_s": round(B / us * 1e6),
       "cpu_oracle_s_per_call": round(cpu_s, 3), "cpu_oracle_images_per_s": round(B / cpu_s, 1)}
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)

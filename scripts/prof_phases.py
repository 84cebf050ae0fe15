#!/usr/bin/env python3
"""HOST issue time of the phases of one eager train step (no synchronisation inside the timed loop)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import torch
import hipseg  # noqa: F401
import models.UNet as un
from models.losses import HybridLoss
from hipseg.optim import Adam

m = un.UNet().cuda().train()
opt = Adam(m.parameters(), lr=1e-3, weight_decay=1e-4)
scaler = torch.amp.GradScaler("cuda")
crit = HybridLoss()
x = torch.rand(16, 3, 256, 256, device="cuda")
t = torch.randint(0, 3, (16, 256, 256), device="cuda")
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
acc = [0.0] * 5


def step(rec):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda"):
        out = m(x)
        t1 = time.perf_counter()
        loss = crit(out, t)
    t2 = time.perf_counter()
    scaler.scale(loss).backward()
    t3 = time.perf_counter()
    scaler.step(opt)
    t4 = time.perf_counter()
    scaler.update()
    t5 = time.perf_counter()
    if rec:
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[i] += d


for _ in range(5):
    step(False)
torch.cuda.synchronize()
N = 40
for _ in range(N):
    step(True)
    torch.cuda.synchronize()  # isolate the phases from back-pressure of a full launch queue
print("host ms/step: forward %.3f  loss %.3f  backward %.3f  scaler.step(opt) %.3f  scaler.update %.3f  total %.3f" % (
    *(1e3 * a / N for a in acc), 1e3 * sum(acc) / N))

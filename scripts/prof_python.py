#!/usr/bin/env python3
"""where the HOST time of one eager train step goes (cProfile over 30 steps, GPU kept busy): prof_python.py [ddp]"""
import cProfile, io, os, pstats, sys, time
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


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda"):
        loss = crit(m(x), t)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()


for _ in range(5):
    step()
torch.cuda.synchronize()
# pure host time: how long Python needs to ISSUE a step (no sync inside)
t0 = time.perf_counter()
for _ in range(30):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue {1e3 * (t1 - t0) / 30:.3f} ms/step, incl. drain {1e3 * (t2 - t0) / 30:.3f} ms/step", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    step()
pr.disable()
torch.cuda.synchronize()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
print(out.getvalue())

#!/usr/bin/env python3
"""Where does the host time of one EAGER train step go?  cProfile over N steps of the bench loop body (UNet 16x3x256^2,
bf16 autocast, HybridLoss, GradScaler + hipseg Adam), no sync inside the profiled region.  Prints the top functions by
own time and by cumulative time."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import torch
import hipseg  # noqa: F401
from hipseg.optim import Adam
from models import UNet as un
from models.losses import HybridLoss

N = int(os.environ.get("HP_STEPS", "30"))
torch.manual_seed(0)
model = un.UNet().cuda().train()
opt = Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
scaler = torch.amp.GradScaler("cuda")
crit = HybridLoss()
x = torch.rand(16, 3, 256, 256, device="cuda")
t = torch.randint(0, 3, (16, 256, 256), device="cuda")


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda"):
        loss = crit(model(x), t)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    return loss


for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue {1e3 * (t1 - t0) / N:.3f} ms/step, wall {1e3 * (t2 - t0) / N:.3f} ms/step (no per-step sync)")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])

#!/usr/bin/env python3
"""inference (model.eval() + no_grad + autocast) throughput of the U-Net forward, one hipGraph replay per batch:
   bench_infer.py [--model UNet|LargeUNet] [--batch 16] [--size 256]      (HIPSEG_NO_FUSED_INFERENCE=1: unfused A/B)"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import torch
import hipseg  # noqa: F401
import models.UNet as un

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="UNet")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
torch.manual_seed(0)
m = getattr(un, a.model)().cuda().eval()
x = torch.rand(a.batch, 3, a.size, a.size, device="cuda")
s = torch.cuda.Stream()
torch.cuda.set_stream(s)


def fwd():
    with torch.no_grad(), torch.autocast("cuda"):
        return m(x)


for _ in range(3):
    y = fwd()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    y = fwd()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    g.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"metric": f"images/s {a.model} 3x{a.size}x{a.size} inference forward (eval, no_grad, bf16 autocast)",
                  "value": round(a.batch / dt, 1), "ms_per_batch": round(dt * 1e3, 4), "batch": a.batch,
                  "fused_conv_bn_relu": not bool(os.environ.get("HIPSEG_NO_FUSED_INFERENCE")),
                  "logits_checksum": float(y.float().abs().mean())}))

"""Diagnostic: bf16 vs fp32 logit error / mask agreement on the HIP path (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-segmentation_amd")]
import numpy as np, torch
import hipseg
from models.UNet import UNet
from oracle import fill

def iou_masks(a, b, ncls=3):
    v = []
    for c in range(ncls):
        pa, pb = a == c, b == c
        u = (pa | pb).sum()
        if u: v.append(float((pa & pb).sum()) / float(u))
    return float(np.mean(v))

g = dict(np.load(os.path.join(ROOT, "tests/golden/models.npz")))
m = UNet(); fill.fill_state_dict(m.state_dict()); m = m.cuda()
x = torch.from_numpy(fill.uniform("c1.x", (2,3,128,128), 0, 1)).cuda()
for mode in ("eval", "train"):
    ref = g[f"unet_c1/{mode}_logits"]
    for prec in ("fp32", "bf16"):
        m.train(mode == "train")
        with hipseg.precision_mode(prec), torch.no_grad():
            out = m(x).float().cpu().numpy()
        err = out - ref
        srt = np.sort(ref, 1); margin = srt[:, -1] - srt[:, -2]
        flips = (out.argmax(1) != ref.argmax(1)).mean()
        print(mode, prec, "rms err %.4g max %.4g | logit std %.3f | margin median %.3f p(<0.02) %.4f | flips %.4f IoU %.4f" % (
            np.sqrt((err**2).mean()), np.abs(err).max(), ref.std(), np.median(margin), (margin < 0.02).mean(), flips,
            iou_masks(out.argmax(1), ref.argmax(1))))

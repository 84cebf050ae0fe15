#!/usr/bin/env python3
"""HBM-bound kernels of the C2 step against the chip's 8 TB/s, from rocprofv3 durations:
    hbm_rates.py <bench.json (with the roofline leg's `kernels`)> <kernel_stats.csv> <steps executed under the profiler> <out.json>
Algorithmic bytes per launch (every operand read once, every result written once) come from the
bench line's `kernels` records (ops._hbm / ops._timed pass them per launch); durations are rocprofv3's per-kernel totals of
a graph-replayed run (scripts/prof_quick.sh), i.e. the step as the benchmark times it.  The weights-stationary 3x3
kernels (<= 64 channels, full resolution) are listed per instantiation with the bytes of the C2 layers they run."""
import csv
import json
import re
import sys

bench = json.load(open(sys.argv[1]))
steps = float(sys.argv[3])
rows = [(r["Name"], int(r["Calls"]), int(r["TotalDurationNs"])) for r in csv.DictReader(open(sys.argv[2]))]
PEAK = 8000.0


def dur(*pats, no=()):
    calls = ns = 0
    for n, c, t in rows:
        if all(re.search(p, n) for p in pats) and not any(re.search(p, n) for p in no):
            calls += c
            ns += t
    return calls / steps, ns / steps / 1e3  # launches per step, us per step


# bench key -> regular expressions on the (mangled or demangled) kernel name
MAP = {
    "hbm:bn_relu_apply": (r"bn_relu_apply_kernel", r"Lb0E"), "hbm:bn_relu_apply+pool": (r"bn_relu_apply_kernel", r"Lb1E"),
    "hbm:bn_bwd_reduce": (r"bn_bwd_reduce_kernel", r"Lb0ELb"), "hbm:bn_bwd_reduce+pool": (r"bn_bwd_reduce_kernel", r"Lb1ELb"),
    "hbm:bn_bwd_apply": (r"bn_bwd_apply_kernel", r"Lb0ELb"), "hbm:bn_bwd_apply+pool": (r"bn_bwd_apply_kernel", r"Lb1ELb"),
    "hbm:stem_fwd": (r"stem_fwd_kernel",), "hbm:stem_bwd": (r"stem_bwd_kernel",), "hbm:head_fwd": (r"head_fwd_kernel", r"Lb0E"),
    "hbm:head_bwd": (r"head_bwd_kernel", r"Lb0E"), "hbm:head_fwd+bn_relu": (r"head_fwd_kernel", r"Lb1E"),
    "hbm:head_bwd+bn_bwd_sums": (r"head_bwd_kernel", r"Lb1E"), "hbm:ce_fwd": (r"ce_fwd_kernel",), "hbm:ce_bwd": (r"ce_bwd_kernel",),
    "hbm:bilinear_fwd": (r"bilinear_fwd_kernel",), "hbm:bilinear_bwd": (r"bilinear_bwd_kernel",),
    "hbm:adam_step": (r"adam_kernel",),
}
out = {"note": "algorithmic GB per step / rocprofv3 us per step; peak 8000 GB/s (MI355X_MICROARCH.md)", "peak_gbps": PEAK,
       "bench_ms_per_step": bench["ms_per_step"], "kernels": {}}
tb = tt = 0.0
for key, pats in MAP.items():
    k = bench["kernels"].get(key)
    if not k:
        continue
    n, us = dur(*pats)
    if not us:
        continue
    mb = k["mb_per_launch"] * k["launches_per_step"]
    out["kernels"][key[4:]] = {"launches_per_step": round(n, 2), "us_per_step": round(us, 1), "mb_per_step": round(mb, 1),
                               "gbps": round(mb / us * 1e3, 1), "frac_hbm": round(mb / us * 1e3 / PEAK, 4)}
    tb += mb
    tt += us
# weights-stationary 3x3 conv (forward + data gradient), C2 layers: bf16 activations in + out, B = 16
px256, px128 = 16 * 256 * 256, 16 * 128 * 128
WSTAT = {"conv3_wstat<4,2> 64->64": (r"conv3_wstat_kernel<4, 2", 2 * px256 * 128 * 2 + 2 * px128 * 128 * 2),
         "conv3_wstat<2,2> 32->64": (r"conv3_wstat_kernel<2, 2", 2 * px256 * 96 * 2),
         "conv3_wstat<4,1> 64->32": (r"conv3_wstat_kernel<4, 1", 2 * px256 * 96 * 2),
         "conv3_wstat<2,1> 32->32": (r"conv3_wstat_kernel<2, 1", 2 * px256 * 64 * 2)}
if bench["config"]["workload"].startswith("UNet 3x256x256") and bench["config"]["per_gpu_batch"] == 16:
    for key, (pat, nbytes) in WSTAT.items():
        n, us = dur(pat)
        if us:
            mb = nbytes / 1e6
            out["kernels"][key] = {"launches_per_step": round(n, 2), "us_per_step": round(us, 1), "mb_per_step": round(mb, 1),
                                   "gbps": round(mb / us * 1e3, 1), "frac_hbm": round(mb / us * 1e3 / PEAK, 4)}
            tb += mb
            tt += us
out["total"] = {"us_per_step": round(tt, 1), "gb_per_step": round(tb / 1e3, 3), "gbps": round(tb / tt * 1e3, 1),
                "frac_hbm": round(tb / tt * 1e3 / PEAK, 4)}
json.dump(out, open(sys.argv[4], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k:28s} {v['launches_per_step']:5.1f}/step {v['us_per_step']:7.1f} us {v['mb_per_step']:8.1f} MB {v['gbps']:7.1f} GB/s  {v['frac_hbm']:.3f}")
print("total", out["total"])

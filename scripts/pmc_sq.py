#!/usr/bin/env python3
"""summarise one rocprofv3 SQ counter pass (scripts/pmc.sh <tag> "SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES") per kernel:
    pmc_sq.py <counter_collection.csv> <out.json>
Ratios only (the raw counters are summed over SEs/XCDs by rocprofv3; their absolute normalisation differs per counter):
  mfma_busy_per_busy   = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES   (compare kernels with each other)
  lds_conflict_frac    = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE    (extra LDS cycles per LDS-array cycle)
  wait_any_frac        = SQ_WAIT_ANY / SQ_WAVE_CYCLES                (wave-cycles spent in any s_waitcnt)
  wait_inst_frac       = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES           (wave-cycles waiting for an instruction slot/dependency)
  wait_lds_frac        = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES"""
import collections
import csv
import json
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)[:70]
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[n].add(r["Dispatch_Id"])
out = {}
for n, c in acc.items():
    g = lambda k: c.get(k, 0.0)
    if g("SQ_WAVE_CYCLES") <= 0 or g("SQ_BUSY_CYCLES") <= 0:
        continue
    out[n] = {"launches": len(launches[n]),
              "mfma_busy_per_busy": round(g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES"), 4),
              "lds_conflict_frac": round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4) if g("SQ_LDS_IDX_ACTIVE") else None,
              "wait_any_frac": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 4),
              "wait_inst_frac": round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 4),
              "wait_lds_frac": round(g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES"), 4),
              "busy_cycles_per_launch": round(g("SQ_BUSY_CYCLES") / len(launches[n]))}
top = dict(sorted(out.items(), key=lambda kv: -kv[1]["busy_cycles_per_launch"] * kv[1]["launches"])[:24])
json.dump({"note": __doc__, "kernels": top}, open(sys.argv[2], "w"), indent=1)
for n, v in top.items():
    print("%-66s n=%3d mfma/busy %.3f  ldsconf %.3f  wait_any %.3f  wait_inst %.3f wait_lds %.3f" % (
        n[:66], v["launches"], v["mfma_busy_per_busy"], v["lds_conflict_frac"] or 0, v["wait_any_frac"], v["wait_inst_frac"], v["wait_lds_frac"]))

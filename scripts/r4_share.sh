#!/bin/bash
# usage (on the one-GPU box): scripts/r4_share.sh <N> [loop] -- rehearsal of `bench.py --gpus N` with all ranks on GPU 0 through gloo
# (HIPSEG_BENCH_SHARE_GPU=1; RCCL refuses two ranks on one device).  Prints what rank 0's JSON line says about the run.
n=${1:-2}; loop=${2:-auto}
out=gpurun_out/share_${n}_${loop}; mkdir -p $out
HIPSEG_BENCH_SHARE_GPU=1 timeout -k 10 600 python bench.py --gpus $n --no-roofline --loop $loop --steps 20 --warmup 5 > $out/line.json 2> $out/err.log
echo "rc=$?"
python - <<PY
import json
d = json.load(open("$out/line.json"))
print(d["n_gpus"], d["value"], d["ms_per_step"], d["distributed"], d.get("fallback_from"))
PY
tail -n 5 $out/err.log

#!/bin/bash
# usage (on the GPU box): scripts/gpu_round.sh <tag>  -> gpurun_out/<tag>/{pytest.log,bench*.json}
tag=${1:-r}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $out/pytest.log | tail -20; exit $rc; }
python bench.py > $out/bench_n1.json 2> $out/bench_n1.err || { tail -5 $out/bench_n1.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$out/bench_n1.json").read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['ms_per_step_event_median'], d.get('roofline',{}).get('frac'), d.get('eager'), d.get('parity'))
for k,v in d.get('kernels',{}).items(): print('  ',k,v)
PY
HIPSEG_BENCH_FORCE_DDP=1 python bench.py --no-cpu-baseline --no-roofline > $out/bench_ddp_evgraph.json 2> $out/bench_ddp_evgraph.err || { tail -5 $out/bench_ddp_evgraph.err; exit 1; }
cut -c1-400 $out/bench_ddp_evgraph.json; grep -o '"distributed".*' $out/bench_ddp_evgraph.json | cut -c1-600
# the ladder, end to end on the GPU: the first loop fails by the env switch, fresh workers run the next one
HIPSEG_BENCH_FORCE_DDP=1 HIPSEG_BENCH_FAIL_LOOP=evgraph python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > $out/bench_ladder_exit.json 2> $out/bench_ladder_exit.err; echo "ladder(exit) rc=$?"
grep -o '"fallback_from".*' $out/bench_ladder_exit.json | cut -c1-300; grep -o '"loop": "[a-z]*", "attempt": [0-9]' $out/bench_ladder_exit.json
HIPSEG_BENCH_FORCE_DDP=1 HIPSEG_BENCH_FAIL_LOOP=evgraph HIPSEG_BENCH_FAIL_MODE=hang HIPSEG_BENCH_ATTEMPT_TIMEOUT=40 python bench.py --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > $out/bench_ladder_hang.json 2> $out/bench_ladder_hang.err; echo "ladder(hang) rc=$?"
grep -o '"fallback_from".*' $out/bench_ladder_hang.json | cut -c1-200; grep -o '"loop": "[a-z]*", "attempt": [0-9]' $out/bench_ladder_hang.json
python bench.py --gpus 2 > $out/bench_gpus2.json 2> $out/bench_gpus2.err; echo "gpus2 rc=$? (expected 2)"; tail -2 $out/bench_gpus2.err
grep -i "warn" $out/bench_n1.err | head -5

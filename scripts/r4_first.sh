#!/bin/bash
# round 4, first GPU call: DDP tests with the capture-before-communicator order, the forced 1-rank evgraph bench, the default bench
set -o pipefail
out=$GRAFT_REPO_ROOT/gpurun_out/r4a
mkdir -p $out
python -m pytest tests/test_gpu_ddp.py -x -q -m gpu > $out/pytest_ddp.log 2>&1; echo "ddp tests rc=$?"; tail -3 $out/pytest_ddp.log
HIPSEG_BENCH_FORCE_DDP=1 python bench.py --no-cpu-baseline --no-eager > $out/bench_ddp1.json 2> $out/bench_ddp1.err; echo "forced ddp rc=$?"
python -c "
import json; d=json.load(open('$out/bench_ddp1.json')); print(d['value'], d['ms_per_step'], d['distributed'])"
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default rc=$?"
python -c "
import json; d=json.load(open('$out/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']); print(d.get('eager')); print(d.get('parity'))"
bash scripts/prof.sh r4a > $out/prof.log 2>&1; tail -3 $out/prof.log

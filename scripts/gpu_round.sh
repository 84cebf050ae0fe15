#!/bin/bash
# usage (on the GPU box): scripts/gpu_round.sh <tag>  -> gpurun_out/<tag>/{pytest.log,bench*.json}
tag=${1:-r}
out=gpurun_out/$tag
mkdir -p $out
python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log

python bench.py --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err || { tail -5 $out/bench_n1.err; exit 1; }
cat $out/bench_n1.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_event_median'], d.get('roofline',{}).get('frac'))"
HIPSEG_BENCH_FORCE_DDP=1 python bench.py --no-cpu-baseline --no-roofline > $out/bench_ddp_graph.json 2> $out/bench_ddp_graph.err || { tail -5 $out/bench_ddp_graph.err; exit 1; }
cat $out/bench_ddp_graph.json
HIPSEG_BENCH_FORCE_DDP=1 python bench.py --no-cpu-baseline --no-roofline --loop eager > $out/bench_ddp_eager.json 2> $out/bench_ddp_eager.err || { tail -5 $out/bench_ddp_eager.err; exit 1; }
cat $out/bench_ddp_eager.json
python bench.py --gpus 2 > $out/bench_gpus2.json 2> $out/bench_gpus2.err; echo "gpus2 rc=$? (expected 2)"; tail -2 $out/bench_gpus2.err
grep -i "warn" $out/bench_n1.err | head -5

#!/bin/bash
# usage (GPU box): scripts/ab_round.sh <tag> <ENV_SWITCH_NAME> [pytest -k expression]
# runs the selected GPU tests, then bench.py alternately with and without the switch (2 x each)
tag=$1; sw=$2; kexpr=${3:-}
out=gpurun_out/$tag
mkdir -p $out
if [ -n "$kexpr" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -x -k "$kexpr" > $out/pytest.log 2>&1; rc=$?
  tail -3 $out/pytest.log
  [ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $out/pytest.log | tail -20; exit $rc; }
fi
for i in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export $sw=1; else unset $sw; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-eager > $out/bench_${v}_$i.json 2> $out/bench_${v}_$i.err || { tail -5 $out/bench_${v}_$i.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$out/bench_${v}_$i.json").read().strip().splitlines()[-1])
k=d.get('kernels',{})
print("$v", d['value'], d['ms_per_step'], ' | '.join(f"{n.split('<')[1][:18]} {x['ms_per_step']:.3f}ms {x['tflops']:.0f}" for n,x in list(k.items())[:3]))
PY
  done
done

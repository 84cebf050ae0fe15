#!/bin/bash
tag=${1:-blk}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $out/pytest.log | tail -20; exit $rc; }
for v in "" 1; do
HIPSEG_NO_BLOCK_CALLS=$v python bench.py --no-cpu-baseline --no-roofline > $out/bench_$v.json 2> $out/bench_$v.err || { tail -5 $out/bench_$v.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$out/bench_$v.json").read().strip().splitlines()[-1])
print("NO_BLOCK_CALLS=$v", d['value'], d['ms_per_step'], d.get('eager'))
PY
done

#!/bin/bash
# per-op Python path (HIPSEG_NO_BLOCK_CALLS=1) against the block calls: parity tests through both, then the bench
out=gpurun_out/${1:-perop}
mkdir -p $out
HIPSEG_NO_BLOCK_CALLS=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -q -x > $out/pytest_perop.log 2>&1; rc=$?
tail -2 $out/pytest_perop.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $out/pytest_perop.log | tail -20; exit $rc; }
python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['eager'])
for k,v in d.get('kernels',{}).items(): print('  ',k,v)
PY

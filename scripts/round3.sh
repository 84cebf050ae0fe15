#!/bin/bash
tag=${1:-r3}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round3.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $out/pytest.log | tail -20; exit $rc; }
echo "--- igemm wstat layers"; MICRO_LAYERS=enc1.c1,dec4.c1 timeout -k 10 300 python scripts/micro_conv.py igemm 2>&1 | grep igemm | tee -a $out/micro.txt
echo "--- wgrad"; timeout -k 10 300 python scripts/micro_conv.py wgrad 2>&1 | grep wgrad | tee -a $out/micro.txt
python bench.py --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err || { tail -5 $out/bench_n1.err; exit 1; }
python - <<PY
import json
d=json.loads(open("$out/bench_n1.json").read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['ms_per_step_event_median'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'), d.get('eager'))
for k,v in d.get('kernels',{}).items(): print('  ',k,v)
PY

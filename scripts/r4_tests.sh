#!/bin/bash
# full GPU test suite + default bench (+ optional A/B env switch list)
set -o pipefail
tag=${1:-t}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "gpu tests rc=$?"; tail -4 $out/pytest.log
python bench.py --no-cpu-baseline --no-eager > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step']); print(d.get('hbm_kernels'));
[print(k, v) for k, v in d['kernels'].items()]"
for sw in "$@"; do
  env $sw python bench.py --no-cpu-baseline --no-eager --no-roofline > $out/bench_$sw.json 2> $out/bench_$sw.err; echo "$sw rc=$?"
  python -c "
import json; d=json.load(open('$out/bench_$sw.json')); print('$sw', d['value'], d['ms_per_step'])"
done

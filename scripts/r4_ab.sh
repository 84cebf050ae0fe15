#!/bin/bash
# usage: scripts/r4_ab.sh <tag> <ENV=VAL> [rounds]  -- alternating default / switched bench runs in one call
tag=$1; sw=$2; n=${3:-2}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --no-eager --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default ', d['value'], d['ms_per_step'])"
  env $sw python bench.py --no-cpu-baseline --no-eager --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$sw', d['value'], d['ms_per_step'])"
done

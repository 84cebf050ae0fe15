#!/bin/bash
# usage: scripts/r4_ab_model.sh <ENV=VAL> <rounds> [bench args] -- alternating default / switched bench runs of another model
sw=$1; n=$2; shift; shift
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --no-eager --no-roofline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default ', d['value'], d['ms_per_step'])"
  env $sw python bench.py --no-cpu-baseline --no-eager --no-roofline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$sw', d['value'], d['ms_per_step'])"
done

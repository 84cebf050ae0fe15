#!/bin/bash
# usage: scripts/r4_ablib.sh <rounds> <libtag> [<libtag> ...] -- alternating bench runs of the default library and A/B builds
# (scripts/build_variant.sh <tag> ...: hipseg/lib/libhipseg_<tag>.so, selected with HIPSEG_LIB)
n=$1; shift
LIBD=$GRAFT_REPO_ROOT/image-segmentation_amd/hipseg/lib
for i in $(seq 1 $n); do
  python bench.py --no-cpu-baseline --no-eager --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('default ', d['value'], d['ms_per_step'])"
  for t in "$@"; do
    HIPSEG_LIB=$LIBD/libhipseg_$t.so python bench.py --no-cpu-baseline --no-eager --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['ms_per_step'])"
  done
done

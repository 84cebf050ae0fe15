#!/bin/bash
# usage: scripts/pmc.sh <tag> "<counters>" [bench args]  -> gpurun_out/pmc_<tag>/
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --loop eager --steps 2 --warmup 1 "$@" > $out/bench.json 2> $out/bench.err
tail -2 $out/bench.err
ls $out

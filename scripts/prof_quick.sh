#!/bin/bash
# usage: scripts/prof_quick.sh <tag> [bench args]  -> gpurun_out/prof_<tag>/kernel_stats.csv (+ top lines on stdout)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
# bench.py as the WORKER itself: under rocprofv3 the supervisor must not spawn a child (the profiler has initialised the GPU)
export HIPSEG_BENCH_WORKER=1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-eager --steps 20 --warmup 5 "$@" > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-160
f=$(find $out -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
python3 $GRAFT_REPO_ROOT/scripts/stats_table.py $out/kernel_stats.csv 28 30

#!/bin/bash
# usage: scripts/prof.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/ (kernel trace + stats CSVs)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
# bench.py as the WORKER itself: under rocprofv3 the supervisor must not spawn a child (the profiler has initialised the GPU)
export HIPSEG_BENCH_WORKER=1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-eager "$@" > $out/bench.json 2> $out/bench.err
tail -2 $out/bench.err; cat $out/bench.json
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -40 {}'

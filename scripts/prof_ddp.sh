#!/bin/bash
# usage: scripts/prof_ddp.sh <tag>  -> kernel-trace stats of the 1-rank forced-DDP graph step vs the plain graph step
tag=$1
cd /tmp && export TMPDIR=/tmp
# bench.py as the WORKER itself: under rocprofv3 the supervisor must not spawn a child (the profiler has initialised the GPU)
export HIPSEG_BENCH_WORKER=1
for mode in plain ddp; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_$mode
  mkdir -p $out
  if [ $mode = ddp ]; then export HIPSEG_BENCH_FORCE_DDP=1; else unset HIPSEG_BENCH_FORCE_DDP; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-eager --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
  tail -1 $out/bench.json | cut -c1-200
  f=$(find $out -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; head -12 $f
done

#!/bin/bash
# usage: scripts/pmc_micro.sh <tag> <igemm|wgrad> <layers> "<counters>"
tag=$1; which=$2; layers=$3; ctrs=$4
cd /tmp && export TMPDIR=/tmp
# bench.py as the WORKER itself: under rocprofv3 the supervisor must not spawn a child (the profiler has initialised the GPU)
export HIPSEG_BENCH_WORKER=1
out=$GRAFT_REPO_ROOT/gpurun_out/pmcm_$tag
mkdir -p $out
MICRO_LAYERS=$layers MICRO_REPS=5 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/scripts/micro_conv.py $which > $out/log.txt 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $f igemm,wgrad_dma

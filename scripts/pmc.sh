#!/bin/bash
# usage: scripts/pmc.sh <tag> "<counters (comma or space separated)>" [bench args]  -> gpurun_out/pmc_<tag>/
#
# One rocprofv3 PMC pass over a short eager bench run (--kernel-trace only: gpurun refuses --pmc combined with the
# sys/runtime/hip/hsa trace domains).  What round 1 learned about this pool's gfx950 counter limits, enforced here so
# it is not hit again:
#   * rocprofv3 aborts with "rocprofiler_create_counter_config ... error code 38: Request exceeds the capabilities of
#     the hardware to collect" (gpurun_out/pmcm_s2/log.txt) when a pass asks for more counters of one block than it has
#     slots.  Passes with up to 8 SQ_* counters worked (gpurun_out/pmcm_s1, pmc_a); the failing request's list was not
#     recorded, so the rule kept is the envelope that is known good: at most 8 counters per pass.
#   * FETCH_SIZE and WRITE_SIZE (TCC) go in passes of their own (the guide's HBM section: separate passes; FETCH_SIZE
#     x2 on gfx950) -- never together, never mixed with SQ_* counters.
#   * a pass with TA_* / TCP_* counters hung the box in round 1 (no log survived, the cause is unknown, no product
#     kernel was implicated: the same kernels ran clean under every other pass).  Refused here; do not retry on a
#     shared pool.
# The command line is saved next to the results (cmd.txt) so a failure can be traced to its counter set.
tag=$1; ctrs=$(echo "$2" | tr ',' ' '); shift; shift
n=$(echo $ctrs | wc -w)
if [ "$n" -lt 1 ] || [ "$n" -gt 8 ]; then echo "pmc.sh: $n counters requested; 1..8 per pass (error code 38 beyond the block's slots)"; exit 2; fi
for c in $ctrs; do
  case $c in
    TA_*|TCP_*) echo "pmc.sh: $c refused: a TA/TCP pass hung the box in round 1 (see header)"; exit 2;;
    FETCH_SIZE|WRITE_SIZE|TCC_*) if [ "$n" -ne 1 ]; then echo "pmc.sh: $c must be the only counter of its pass"; exit 2; fi;;
  esac
done
cd /tmp && export TMPDIR=/tmp
# bench.py as the WORKER itself: under rocprofv3 the supervisor must not spawn a child (the profiler has initialised the GPU)
export HIPSEG_BENCH_WORKER=1
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
echo "rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o pmc -- python3 bench.py --no-cpu-baseline --no-roofline --no-eager --loop eager --optimizer hip --steps 2 --warmup 1 $@" > $out/cmd.txt
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-roofline --no-eager --loop eager --steps 2 --warmup 1 "$@" > $out/bench.json 2> $out/bench.err
tail -2 $out/bench.err
ls $out

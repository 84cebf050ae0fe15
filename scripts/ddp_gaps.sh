#!/bin/bash
# forced 1-rank DDP (evgraph) under rocprofv3 --kernel-trace: step table + the trace for scripts/trace_gaps.py
export HIPSEG_BENCH_FORCE_DDP=1
bash scripts/prof_quick.sh ddpgaps > gpurun_out/prof_ddpgaps_table.txt 2>&1
head -3 gpurun_out/prof_ddpgaps_table.txt | cut -c1-200

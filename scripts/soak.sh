#!/bin/bash
# usage (on the GPU box): scripts/soak.sh [steps] -- long replays of the default step and of the forced 1-rank DDP loop:
# throughput, final loss (a fixed synthetic batch: it must fall and stay finite through GradScaler growth events)
n=${1:-6000}
python bench.py --steps $n --warmup 8 --no-cpu-baseline --no-eager --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('graph  ', d['value'], d['ms_per_step'], 'final loss', d['config']['final_loss'])"
HIPSEG_BENCH_FORCE_DDP=1 python bench.py --steps $n --warmup 8 --no-cpu-baseline --no-eager --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('evgraph', d['value'], d['ms_per_step'], 'final loss', d['config']['final_loss'], 'in sync', d['distributed']['ranks_in_sync'], 'capture attempts', d['distributed']['capture_attempts'])"

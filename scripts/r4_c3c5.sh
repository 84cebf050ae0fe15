#!/bin/bash
# C3 / C5 bench lines (with roofline) + rocprofv3 step tables
out=$GRAFT_REPO_ROOT/gpurun_out/r4e; mkdir -p $out
python bench.py --no-cpu-baseline --no-eager --model LargeUNet --batch 8 --size 512 > $out/bench_c3.json 2> $out/bench_c3.err; echo "c3 rc=$?"
python bench.py --no-cpu-baseline --no-eager --model ClipUnet --batch 32 --size 224 > $out/bench_c5.json 2> $out/bench_c5.err; echo "c5 rc=$?"
python -c "
import json
for t in ('c3','c5'):
    d=json.load(open('$out/bench_%s.json'%t)); print(t, d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d.get('hbm_kernels'))"
bash scripts/prof_quick.sh r4_c3 --model LargeUNet --batch 8 --size 512 > $out/c3_step_table.txt 2>&1; head -45 $out/c3_step_table.txt
bash scripts/prof_quick.sh r4_c5 --model ClipUnet --batch 32 --size 224 > $out/c5_step_table.txt 2>&1; head -60 $out/c5_step_table.txt

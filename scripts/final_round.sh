#!/bin/bash
# usage (on the GPU box): scripts/final_round.sh <tag>  -> gpurun_out/final_<tag>/: GPU tests, default bench, C3/C5, forced-DDP
# rehearsal, rocprofv3 kernel stats and the two PMC traffic passes of the SAME sources
tag=${1:-r}
out=gpurun_out/final_$tag
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; tail -2 $out/pytest.log
python bench.py > $out/bench_default.json 2> $out/bench_default.err; cut -c1-200 $out/bench_default.json
python bench.py --no-cpu-baseline --no-roofline --model LargeUNet --batch 8 --size 512 > $out/bench_c3.json 2>/dev/null; cut -c1-200 $out/bench_c3.json
python bench.py --no-cpu-baseline --no-roofline --model ClipUnet --batch 32 --size 224 > $out/bench_c5.json 2>/dev/null; cut -c1-200 $out/bench_c5.json
HIPSEG_BENCH_FORCE_DDP=1 python bench.py --no-cpu-baseline --no-roofline > $out/bench_ddp1.json 2>/dev/null; cut -c1-200 $out/bench_ddp1.json
HIPSEG_BENCH_FORCE_DDP=1 python bench.py --no-cpu-baseline --no-roofline --loop eager > $out/bench_ddp1_eager.json 2>/dev/null; cut -c1-200 $out/bench_ddp1_eager.json
HIPSEG_BENCH_FORCE_DDP=1 python bench.py --no-cpu-baseline --no-roofline --loop graph > $out/bench_ddp1_graph.json 2>/dev/null; cut -c1-200 $out/bench_ddp1_graph.json
scripts/prof_quick.sh final_$tag > $out/step_table.txt 2>&1; head -4 $out/step_table.txt
scripts/pmc.sh ffetch_$tag FETCH_SIZE > /dev/null 2>&1 && scripts/pmc.sh fwrite_$tag WRITE_SIZE > /dev/null 2>&1 && \
  python3 scripts/pmc_traffic.py $(find gpurun_out/pmc_ffetch_$tag -name "*counter_collection.csv") $(find gpurun_out/pmc_fwrite_$tag -name "*counter_collection.csv") $out/pmc_traffic.json | head -3

#!/bin/bash
# usage (on the GPU box): scripts/r4_final.sh <tag>  -> gpurun_out/final_<tag>/
# GPU tests, default bench (the driver's command), C3 / C5 lines with roofline, forced 1-rank DDP rehearsals of the three
# loops, rocprofv3 kernel tables of C2 / C3 / C5, and the two PMC traffic passes (FETCH_SIZE, WRITE_SIZE) of C2 and C3.
tag=${1:-r4}
out=gpurun_out/final_$tag
mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; tail -2 $out/pytest.log
python bench.py > $out/bench_default.json 2> $out/bench_default.err; cut -c1-160 $out/bench_default.json
python bench.py --no-cpu-baseline --no-eager --model LargeUNet --batch 8 --size 512 > $out/bench_c3.json 2> $out/bench_c3.err; cut -c1-160 $out/bench_c3.json
python bench.py --no-cpu-baseline --no-eager --model ClipUnet --batch 32 --size 224 > $out/bench_c5.json 2> $out/bench_c5.err; cut -c1-160 $out/bench_c5.json
HIPSEG_BENCH_FORCE_DDP=1 python bench.py --no-cpu-baseline --no-roofline > $out/bench_ddp1.json 2> $out/bench_ddp1.err; cut -c1-160 $out/bench_ddp1.json
HIPSEG_BENCH_FORCE_DDP=1 python bench.py --no-cpu-baseline --no-roofline --loop eager > $out/bench_ddp1_eager.json 2>/dev/null; cut -c1-160 $out/bench_ddp1_eager.json
HIPSEG_BENCH_FORCE_DDP=1 python bench.py --no-cpu-baseline --no-roofline --loop graph > $out/bench_ddp1_graph.json 2>/dev/null; cut -c1-160 $out/bench_ddp1_graph.json
echo "--- step tables"
scripts/prof_quick.sh final_${tag}_c2 > $out/c2_step_table.txt 2>&1; head -3 $out/c2_step_table.txt
scripts/prof_quick.sh final_${tag}_c3 --model LargeUNet --batch 8 --size 512 > $out/c3_step_table.txt 2>&1; head -3 $out/c3_step_table.txt
scripts/prof_quick.sh final_${tag}_c5 --model ClipUnet --batch 32 --size 224 > $out/c5_step_table.txt 2>&1; head -3 $out/c5_step_table.txt
echo "--- PMC traffic"
scripts/pmc.sh ffetch_$tag FETCH_SIZE > /dev/null 2>&1 && scripts/pmc.sh fwrite_$tag WRITE_SIZE > /dev/null 2>&1 && \
  python3 scripts/pmc_traffic.py $(find gpurun_out/pmc_ffetch_$tag -name "*counter_collection.csv") $(find gpurun_out/pmc_fwrite_$tag -name "*counter_collection.csv") $out/pmc_traffic.json | head -4
scripts/pmc.sh ffetch_${tag}_c3 FETCH_SIZE --model LargeUNet --batch 8 --size 512 > /dev/null 2>&1 && scripts/pmc.sh fwrite_${tag}_c3 WRITE_SIZE --model LargeUNet --batch 8 --size 512 > /dev/null 2>&1 && \
  python3 scripts/pmc_traffic.py $(find gpurun_out/pmc_ffetch_${tag}_c3 -name "*counter_collection.csv") $(find gpurun_out/pmc_fwrite_${tag}_c3 -name "*counter_collection.csv") $out/pmc_traffic_c3.json | head -4
echo "--- SQ counters"
scripts/pmc.sh fsq_$tag "SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" > /dev/null 2>&1 && \
  python3 scripts/pmc_sq.py $(find gpurun_out/pmc_fsq_$tag -name "*counter_collection.csv") $out/pmc_sq.json | head -12

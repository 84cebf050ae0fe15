#!/bin/bash
# usage (on the GPU box): scripts/r4_up.sh <tag> -- tests of the ConvBlock + ConvTranspose2d node, then alternating bench runs
tag=${1:-up1}
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py -q -x -k "convT_data or one_node or per_op or head" > $out/pytest_up.log 2>&1; tail -5 $out/pytest_up.log
grep -q "failed\|error" $out/pytest_up.log && exit 1
bash scripts/r4_ab.sh $tag HIPSEG_NO_UP_FUSE=1 2 | tee $out/ab.txt

#!/bin/bash
# usage (on the GPU box): scripts/r4_head.sh <tag> -- tests of the fused last-block + head node, then alternating bench runs
tag=${1:-head1}
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py -q -x -k "head or per_op or alias" > $out/pytest_head.log 2>&1; tail -5 $out/pytest_head.log
grep -q "failed\|error" $out/pytest_head.log && exit 1
bash scripts/r4_ab.sh $tag HIPSEG_NO_HEAD_FUSE=1 2 | tee $out/ab.txt

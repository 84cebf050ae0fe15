#!/bin/bash
# usage (GPU box): scripts/m16_round.sh <tag>: conv kernel tests, then an A/B micro-benchmark old vs new kernel
tag=${1:-m16}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv3" > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
L=enc2.c0,enc2.c1,enc3.c0,enc3.c1,bott.c0,bott.c1,dec1.c0,dec1.c1,dec2.c0,dec2.c1,enc3.c0^T,bott.c0^T,dec1.c0^T,dec2.c0^T
for i in 1 2; do
  echo "--- new (m16)"; MICRO_LAYERS=$L timeout -k 10 300 python scripts/micro_conv.py igemm 2>&1 | tee -a $out/new.txt || exit 1
  echo "--- old (ring)"; HIPSEG_NO_M16=1 MICRO_LAYERS=$L timeout -k 10 300 python scripts/micro_conv.py igemm 2>&1 | tee -a $out/old.txt || exit 1
done

#!/bin/bash
tag=${1:-m16b}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv3" > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && exit $rc
HIPSEG_M16_ROWS=8 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "conv3 and bf16" > $out/pytest_r8.log 2>&1; rc=$?
tail -3 $out/pytest_r8.log
[ $rc -ne 0 ] && exit $rc
L=enc2.c0,enc2.c1,enc3.c0,enc3.c1,bott.c0,bott.c1,dec1.c0,dec1.c1,dec2.c0,dec2.c1,enc3.c0^T,bott.c0^T,dec1.c0^T,dec2.c0^T
run() { echo "--- $1"; shift; env "$@" MICRO_LAYERS=$L timeout -k 10 300 python scripts/micro_conv.py igemm 2>&1 | grep igemm | tee -a $out/all.txt || exit 1; }
for i in 1 2; do
run "default (16 / 8+loader)"
run "rows=8 + loader everywhere" HIPSEG_M16_ROWS=8
run "rows=8 no loader" HIPSEG_M16_ROWS=8 HIPSEG_M16_NO_LOADER=1
run "rows=16 everywhere" HIPSEG_M16_ROWS=16
done

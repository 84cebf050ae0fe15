#!/bin/bash
tag=${1:-ct}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "convT" > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $out/pytest.log | tail -20; exit $rc; }
for i in 1 2; do
echo "--- streaming"; timeout -k 10 300 python scripts/micro_convt.py 2>&1 | grep convT | tee -a $out/new.txt || exit 1
echo "--- gemm kernels"; HIPSEG_NO_CONVT_STREAM=1 timeout -k 10 300 python scripts/micro_convt.py 2>&1 | grep convT | tee -a $out/old.txt || exit 1
done

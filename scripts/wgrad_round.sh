#!/bin/bash
tag=${1:-wg}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "dgrad_wgrad" > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $out/pytest.log | tail -20; exit $rc; }
for i in 1 2; do
echo "--- tr16"; timeout -k 10 300 python scripts/micro_conv.py wgrad 2>&1 | grep wgrad | tee -a $out/new.txt || exit 1
done

#!/bin/bash
tag=${1:-wg}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "dgrad_wgrad" > $out/pytest.log 2>&1; rc=$?
tail -3 $out/pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|error\|assert" $out/pytest.log | tail -20; exit $rc; }
for i in 1 2; do
echo "--- new (tr16)"; timeout -k 10 300 python scripts/micro_conv.py wgrad 2>&1 | grep wgrad | tee -a $out/new.txt || exit 1
echo "--- old"; HIPSEG_NO_WGRAD_TR16=1 timeout -k 10 300 python scripts/micro_conv.py wgrad 2>&1 | grep wgrad | tee -a $out/old.txt || exit 1
done
HIPSEG_LIB=$PWD/image-segmentation_amd/hipseg/lib/libhipseg_wgstamp.so timeout -k 10 300 python scripts/micro_wgstamp.py 2>&1 | grep -v amdgpu.ids | tee $out/stamp.txt

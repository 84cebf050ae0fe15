#!/bin/bash
tag=${1:-stamp}
out=gpurun_out/$tag
mkdir -p $out
LIBD=image-segmentation_amd/hipseg/lib
for rows in 16 8; do
echo "=== rows=$rows"
HIPSEG_M16_ROWS=$rows HIPSEG_LIB=$PWD/$LIBD/libhipseg_stamp.so timeout -k 10 300 python scripts/micro_stamp.py 2>&1 | grep -v amdgpu.ids | tee -a $out/stamp.txt || exit 1
done

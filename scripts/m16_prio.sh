#!/bin/bash
tag=${1:-prio}
out=gpurun_out/$tag
mkdir -p $out
LIBD=$PWD/image-segmentation_amd/hipseg/lib
L=enc2.c0,enc2.c1,enc3.c0,enc3.c1,bott.c1,dec2.c0,dec2.c1,dec1.c0^T,dec2.c0^T
run() { echo "--- $1"; shift; env "$@" MICRO_LAYERS=$L timeout -k 10 300 python scripts/micro_conv.py igemm 2>&1 | grep igemm | tee -a $out/all.txt || exit 1; }
for i in 1 2; do
run "base"
run "prio2" HIPSEG_LIB=$LIBD/libhipseg_prio.so
done
echo "=== stamps base"; MICRO_LAYERS=enc3.c1,enc2.c1 HIPSEG_LIB=$LIBD/libhipseg_stamp.so timeout -k 10 300 python scripts/micro_stamp.py 2>&1 | grep -v amdgpu.ids | tee -a $out/stamp.txt
echo "=== stamps prio"; MICRO_LAYERS=enc3.c1,enc2.c1 HIPSEG_LIB=$LIBD/libhipseg_stampprio.so timeout -k 10 300 python scripts/micro_stamp.py 2>&1 | grep -v amdgpu.ids | tee -a $out/stamp.txt
echo "=== wgrad micro"; timeout -k 10 300 python scripts/micro_conv.py wgrad 2>&1 | grep wgrad | tee -a $out/wgrad.txt

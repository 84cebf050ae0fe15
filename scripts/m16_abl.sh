#!/bin/bash
# usage (GPU box): scripts/m16_abl.sh <tag>: ablations of the 16x16x32 conv kernel (ablation build: libhipseg_abl.so)
tag=${1:-m16abl}
out=gpurun_out/$tag
mkdir -p $out
LIBD=image-segmentation_amd/hipseg/lib
L=${ABL_LAYERS:-enc2.c1,enc3.c1,bott.c1}
run() { echo "--- $1"; shift; env "$@" MICRO_LAYERS=$L timeout -k 10 300 python scripts/micro_conv.py igemm 2>&1 | grep igemm | tee -a $out/all.txt || exit 1; }
for rows in 16 8; do
for b in 0 1 2 4 5 6 7 8 12 13 16 20 24 28; do
  run "rows=$rows ablate dbg=$b" HIPSEG_M16_ROWS=$rows HIPSEG_LIB=$PWD/$LIBD/libhipseg_abl.so HIPSEG_IGEMM_DEBUG=$b
done
done

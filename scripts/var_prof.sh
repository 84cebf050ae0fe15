#!/bin/bash
# usage (GPU box): scripts/var_prof.sh <tag> <variant> <grep pattern>: rocprofv3 step tables of the shipped library and of
# libhipseg_<variant>.so (scripts/build_variant.sh), alternating, two rounds
tag=$1; var=$2; pat=$3
out=gpurun_out/$tag
mkdir -p $out
LIBD=$PWD/image-segmentation_amd/hipseg/lib
for rep in 1 2; do
for v in base $var; do
  unset HIPSEG_LIB
  [ $v != base ] && export HIPSEG_LIB=$LIBD/libhipseg_$v.so
  bash scripts/prof_quick.sh ${tag}_${v}_$rep > $out/table_${v}_$rep.txt 2>&1
  echo "--- $v $rep"; sed -n 1,2p $out/table_${v}_$rep.txt | cut -c1-150; grep "$pat" $out/table_${v}_$rep.txt | cut -c1-120
done
done

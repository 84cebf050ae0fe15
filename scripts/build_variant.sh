#!/bin/bash
# usage: scripts/build_variant.sh <tag> <file.hip> [-DFLAG=..]...  -> hipseg/lib/libhipseg_<tag>.so
# Rebuilds ONE translation unit with extra defines and links it with the other (already built) objects: an A/B
# build for a single gpurun call (select it with HIPSEG_LIB=<path>).
set -e
cd "$(dirname "$0")/../image-segmentation_amd"
tag=$1; src=$2; shift 2
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value "$@" -c csrc/$base.hip -o hipseg/lib/${base}_$tag.o
objs=""
for o in pack bn pointwise loss records augment optim sync conv_igemm conv3_m16 conv_wgrad; do
  if [ $o = $base ]; then objs="$objs hipseg/lib/${base}_$tag.o"; else objs="$objs hipseg/lib/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hipseg/lib/libhipseg_$tag.so $objs
echo built hipseg/lib/libhipseg_$tag.so

#!/bin/bash
# usage: scripts/build_variant.sh <tag> <file.hip>[,<file2.hip>...] [-DFLAG=..]...  -> hipseg/lib/libhipseg_<tag>.so
# Rebuilds the named translation units with extra defines and links them with the other (already built) objects: an
# A/B build for a single gpurun call (select it with HIPSEG_LIB=<path>).
# Ablation builds (-DHIPSEG_ABLATE: HIPSEG_IGEMM_DEBUG / HIPSEG_WGRAD_DEBUG bits, wrong results by design) of the
# convolution kernels need conv_igemm.hip in the list: it reads the environment switch for conv3_m16.hip too.
set -e
cd "$(dirname "$0")/../image-segmentation_amd"
tag=$1; srcs=$2; shift 2
bases=""
for src in ${srcs//,/ }; do
  base=$(basename $src .hip)
  bases="$bases $base"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm "$@" -c csrc/$base.hip -o hipseg/lib/${base}_$tag.o &
done
wait
objs=""
for o in pack bn pointwise loss records augment optim sync conv_igemm conv3_m16 convt_stream conv_wgrad convt_wgrad block; do
  if [[ " $bases " == *" $o "* ]]; then objs="$objs hipseg/lib/${o}_$tag.o"; else objs="$objs hipseg/lib/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hipseg/lib/libhipseg_$tag.so $objs -ldl
echo built hipseg/lib/libhipseg_$tag.so

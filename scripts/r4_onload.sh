#!/bin/bash
# the cost side of "BatchNorm + ReLU on load", measured on the layer where it would pay most
# (enc1.c1: 64 -> 64 at 256 x 256, B = 16): forward conv and weight gradient with the in-LDS rewrite vs without, next to
# the bn_relu_apply pass it would remove.
LIBD=$GRAFT_REPO_ROOT/image-segmentation_amd/hipseg/lib
out=$GRAFT_REPO_ROOT/gpurun_out/r4_onload; mkdir -p $out
for r in 1 2; do
echo "--- round $r: forward conv (weights-stationary kernel)"
HIPSEG_LIB=$LIBD/libhipseg_wsstamp.so python scripts/micro_wsstamp.py | grep -A1 "enc1.c1\|dec4.c1"
echo "--- with the on-load rewrite"
HIPSEG_LIB=$LIBD/libhipseg_wsonload.so python scripts/micro_wsstamp.py | grep -A1 "enc1.c1\|dec4.c1"
echo "--- weight gradient (tr16 kernel), enc1.c1"
MICRO_LAYERS=enc1.c1 python scripts/micro_conv.py wgrad
echo "--- with the on-load rewrite of its P operand"
HIPSEG_LIB=$LIBD/libhipseg_wgonload.so MICRO_LAYERS=enc1.c1 python scripts/micro_conv.py wgrad
done
echo "--- the pass it would remove (bn_relu_apply 64 ch @ 256^2: enc1 row; 32 ch: dec4 row)"
python scripts/micro_bn.py | grep "enc1 \|dec4"

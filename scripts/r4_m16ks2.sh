#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4d; mkdir -p $out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round4.py -x -q -m gpu > $out/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $out/pytest.log
export MICRO_LAYERS=dec1.c0,dec1.c1,bott.c0^T,dec2.c1
for i in 1 2; do
python scripts/micro_conv.py igemm
HIPSEG_NO_M16_KS2=1 python scripts/micro_conv.py igemm | sed 's/^/NO_KS2 /'
done

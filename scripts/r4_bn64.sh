#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4f; mkdir -p $out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round4.py tests/test_gpu_parity.py -x -q -m gpu > $out/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $out/pytest.log
python scripts/micro_bn64.py
HIPSEG_NO_M16_BN64=1 python scripts/micro_bn64.py
python scripts/micro_bn64.py
HIPSEG_NO_M16_BN64=1 python scripts/micro_bn64.py

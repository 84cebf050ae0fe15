#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p $out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round4.py tests/test_gpu_round2.py tests/test_gpu_parity.py -x -q -m gpu > $out/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $out/pytest.log
for sw in "X=1" "HIPSEG_NO_WGRAD_RAGGED=1" "X=2" "HIPSEG_NO_WGRAD_RAGGED=1 HIPSEG_NO_CONVT_WGRAD=1"; do
  env $sw python bench.py --no-cpu-baseline --no-eager --no-roofline --model ClipUnet --batch 32 --size 224 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$sw', d['value'], d['ms_per_step'])"
done

#!/bin/bash
# round 5, GPU call F: gradient through hT (second pass on the lockstep plan), use_rnn with masked data
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5f
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_torch_op.py tests/test_hip_generic.py tests/test_hip_call_lifetime.py \
  tests/test_hip_masked_return_path.py tests/test_hip_parity.py tests/test_hip_properties.py -q -m gpu 2>&1 | tail -40 > $OUT/pytest.log
NJODE_GENERIC=1 timeout 900 python -m pytest tests/test_hip_torch_op.py -q -m gpu -k "hT" 2>&1 | tail -15 > $OUT/pytest_generic_hT.log
tail -5 $OUT/pytest.log; tail -5 $OUT/pytest_generic_hT.log

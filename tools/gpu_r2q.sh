#!/usr/bin/env bash
# 's' cones in the device loops: new tests + regression of the l/q loops and the ldl/chol singular-S switch
export PYTHONPATH=.
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_sdp.py -q -m gpu -x 2>&1 | tail -40 ) > $O/r2q_sdp.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_round2.py tests/test_gpu_kkt.py -q -m gpu 2>&1 | tail -15 ) > $O/r2q_regress.log 2>&1
echo done

#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( SDP_MANY=1 timeout 200 python tests/sdp_time_dev.py 20 2>&1 | grep -v amdgpu | tail -5 ) >> $O/r3b.log 2>&1
( timeout 300 python -m pytest tests/test_gpu_sdp.py -v -m gpu -x 2>&1 | grep -E "PASSED|FAILED|passed|failed|fault|Error" | tail -20 ) >> $O/r3b.log 2>&1
echo done

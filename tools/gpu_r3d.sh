#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
for rep in 1 2 3; do
( timeout 600 python -m pytest tests/test_gpu_sdp.py tests/test_gpu_sdp_ops.py -q -m gpu 2>&1 | grep -E "passed|failed|fault|Error|assert" | tail -8 ) >> $O/r3d.log 2>&1
done
echo done

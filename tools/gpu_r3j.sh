#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( timeout 600 python -m pytest tests/test_gpu_sdp_ops.py tests/test_gpu_sdp.py -q -m gpu 2>&1 | tail -4 ) > $O/r3j_tests.log 2>&1
SDP_MANY=1 timeout 300 python tools/dev/sdp_time_dev.py 20 > $O/r3j_time.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_sdp.py -q -m gpu 2>&1 | tail -2 ) >> $O/r3j_tests.log 2>&1
echo done

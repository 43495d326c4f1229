#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( timeout 300 python -m pytest tests/test_gpu_sdp.py -q -m gpu -x 2>&1 | tail -15 ) > $O/r3a_sdp.log 2>&1
SDP_MANY=1 timeout 300 python tests/sdp_time_dev.py 20 > $O/r3a_time.log 2>&1
echo done

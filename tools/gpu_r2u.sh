#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_sdp.py tests/test_gpu_solvers.py -q -m gpu 2>&1 | tail -8 ) > $O/r2u_tests.log 2>&1
timeout 600 python tests/sdp_time_dev.py 60 100 200 > $O/r2u_time.log 2>&1
echo done

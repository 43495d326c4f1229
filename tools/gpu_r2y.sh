#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_batch.py -q -m gpu -x 2>&1 | tail -25 ) > $O/r2y_tests.log 2>&1
echo done

#!/usr/bin/env bash
# second-order cones in the batched engine: new tests + the loops that share the kernels
export PYTHONPATH=.
O=gpurun_out
( timeout 600 python -m pytest tests/test_gpu_batch.py -q -m gpu 2>&1 | tail -40 ) > $O/r3l_batch.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_resident.py -q -m gpu -x 2>&1 | tail -8 ) > $O/r3l_resident.log 2>&1
echo done

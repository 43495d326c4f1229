#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x 2>&1 | tail -4 ) > $O/r3h_tests.log 2>&1
for h in 1 0 1 0; do
  MI355KKT_POTRF_HALF=$h timeout 120 python tools/dev/bench_potrf_dev.py 8192 2>&1 | tail -1 >> $O/r3h_potrf.log
done
for n in 4096 2048; do for h in 1 0; do MI355KKT_POTRF_HALF=$h timeout 120 python tools/dev/bench_potrf_dev.py $n 2>&1 | tail -1 >> $O/r3h_potrf.log; done; done
echo done

#!/usr/bin/env bash
# round-3 call 24: does a GPU pytest process crash at interpreter exit?  (normal exit forced, exit codes and the end of stderr kept)
export PYTHONPATH=.
O=gpurun_out/c24; mkdir -p $O
for r in 1 2 3; do
MI355KKT_TEST_NORMAL_EXIT=1 timeout 40 python -m pytest tests/test_gpu_batch.py tests/test_gpu_round3.py -q -m gpu -x > $O/run_$r.log 2>&1
echo "run $r exit $?" >> $O/summary.log
done
echo done

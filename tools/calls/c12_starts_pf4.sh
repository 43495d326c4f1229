#!/usr/bin/env bash
# round-3 call 12: primalstart / dualstart / initvals in the device loops; four steps of operand look-ahead in the tile Cholesky bulk
export PYTHONPATH=.
O=gpurun_out/c12; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_resident.py -q -m gpu 2>&1 | tail -8 ) > $O/tests.log 2>&1
{
for b in 16 164 16 164; do MI355KKT_POTRF_BK=$b timeout 300 python tools/dev/bench_potrf_dev.py 8192; done
for b in 16 164; do MI355KKT_POTRF_BK=$b timeout 300 python tools/dev/bench_potrf_dev.py 4096; done
} > $O/potrf.log 2>&1
( MI355KKT_POTRF_BK=164 timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k potrf 2>&1 | tail -3 ) > $O/tests_pf4.log 2>&1
echo done

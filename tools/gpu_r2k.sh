#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_round2.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -30 ) > $O/r2k_new_tests.log 2>&1
timeout 900 python bench.py --no-cpu-baseline > $O/r2k_bench.json 2> $O/r2k_bench.err
MI355KKT_TRSV_NOINV=1 timeout 900 python bench.py --no-cpu-baseline --steps 5 > $O/r2k_bench_noinv.json 2> $O/r2k_bench_noinv.err
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > $O/r2k_gpu_tests.log 2>&1
echo done

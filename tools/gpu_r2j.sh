#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -30 ) > $O/r2j_new_tests.log 2>&1
timeout 900 python bench.py > $O/r2j_bench.json 2> $O/r2j_bench.err
timeout 600 python bench.py --workload socp --no-cpu-baseline > $O/r2j_bench_socp.json 2> $O/r2j_bench_socp.err
timeout 600 python bench.py --workload sparse --no-cpu-baseline > $O/r2j_bench_sparse.json 2> $O/r2j_bench_sparse.err
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > $O/r2j_gpu_tests.log 2>&1
echo done

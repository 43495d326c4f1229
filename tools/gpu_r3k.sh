#!/usr/bin/env bash
# maximum-size test + the triangular solves with double-buffered LDS vectors and split FMA chains
export PYTHONPATH=.
O=gpurun_out
( timeout 400 python -m pytest tests/test_gpu_maxsize.py -q -m gpu 2>&1 | tail -15 ) > $O/r3k_maxsize.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_sparse.py -q -m gpu -x 2>&1 | tail -6 ) > $O/r3k_tests.log 2>&1
timeout 300 python bench.py --no-cpu-baseline > $O/r3k_bench.json 2> $O/r3k_bench.err
timeout 300 python bench.py --workload sparse --no-cpu-baseline --steps 10 > $O/r3k_sparse.json 2> $O/r3k_sparse.err
echo done

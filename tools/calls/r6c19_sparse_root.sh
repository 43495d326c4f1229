#!/usr/bin/env bash
# round-6 call 19: the sparse engine's dense root through the 512-row solves: sparse tests + the sparse bench line + unpoisoned smoke
export PYTHONPATH=.
O=gpurun_out/r6c19; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_sparse_big.py tests/test_gpu_lifecycle.py -m gpu -q -x > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
( timeout 600 python bench.py --workload sparse --no-cpu-baseline ) > $O/bench_sparse.json 2> $O/bench_sparse.err
cut -c1-1400 $O/bench_sparse.json

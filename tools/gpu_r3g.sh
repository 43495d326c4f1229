#!/usr/bin/env bash
# sparse engine after the host-analysis changes: tests, bench lines (symbolic + first factor time), end-to-end sparse coneqp
export PYTHONPATH=.
O=gpurun_out
( timeout 600 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -3 ) > $O/r3g_tests.log 2>&1
MI355KKT_SPARSE_DEBUG=1 timeout 300 python bench.py --workload sparse --no-cpu-baseline --steps 10 > $O/r3g_sparse46.json 2> $O/r3g_sparse46.err
timeout 300 python bench.py --workload sparse --grid 64 --no-cpu-baseline --steps 5 > $O/r3g_sparse64.json 2> $O/r3g_sparse64.err
echo done

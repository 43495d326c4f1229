#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_round2.py -q -m gpu -x 2>&1 | tail -12 ) > $O/r2w_tests.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k sparse 2>&1 | tail -5 ) >> $O/r2w_tests.log 2>&1
timeout 300 python bench.py --workload sparse --no-cpu-baseline --steps 10 > $O/r2w_sparse46.json 2> $O/r2w_sparse46.err
MI355KKT_SN_MAXW=256 timeout 300 python bench.py --workload sparse --no-cpu-baseline --steps 10 > $O/r2w_sparse46_w256.json 2> $O/r2w_sparse46_w256.err
timeout 300 python bench.py --workload sparse --mesh tet --grid 39 --no-cpu-baseline --steps 10 > $O/r2w_sparse_tet.json 2> $O/r2w_sparse_tet.err
timeout 300 python bench.py --workload sparse --grid 64 --no-cpu-baseline --steps 5 > $O/r2w_sparse64.json 2> $O/r2w_sparse64.err
echo done
( timeout 600 python bench.py --workload sparse --grid 100 --no-cpu-baseline --steps 3 --warmup 1 > $O/r2w_sparse100.json 2> $O/r2w_sparse100.err ; echo "rc=$?" >> $O/r2w_sparse100.err )

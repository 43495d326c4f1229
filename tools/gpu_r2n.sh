#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_round2.py -q -m gpu 2>&1 | tail -15 ) > $O/r2n_tests.log 2>&1
for v in 1 0; do
  MI355KKT_SPARSE_TILES=$v timeout 300 python bench.py --workload sparse --no-cpu-baseline --steps 10 > $O/r2n_sparse46_t$v.json 2> $O/r2n_sparse46_t$v.err
done
timeout 300 python bench.py --workload sparse --mesh tet --grid 39 --no-cpu-baseline --steps 10 > $O/r2n_sparse_tet.json 2> $O/r2n_sparse_tet.err
timeout 300 python bench.py --workload sparse --grid 64 --no-cpu-baseline --steps 5 > $O/r2n_sparse64.json 2> $O/r2n_sparse64.err
( timeout 600 python bench.py --workload sparse --grid 100 --no-cpu-baseline --steps 3 --warmup 1 > $O/r2n_sparse100.json 2> $O/r2n_sparse100.err ; echo "rc=$?" >> $O/r2n_sparse100.err )
echo done

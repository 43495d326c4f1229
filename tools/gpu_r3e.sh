#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
for w in 256 128 64; do
  MI355KKT_SP_WIDE=$w timeout 300 python bench.py --workload sparse --no-cpu-baseline --steps 10 > $O/r3e_sparse46_w$w.json 2> $O/r3e_sparse46_w$w.err
done
( MI355KKT_SP_WIDE=128 timeout 600 python -m pytest tests/test_gpu_sparse.py -q -m gpu 2>&1 | tail -3 ) > $O/r3e_tests.log 2>&1
echo done

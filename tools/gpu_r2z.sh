#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_resident.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -6 ) > $O/r2z_tests.log 2>&1
for r in 4 8 64; do
  timeout 300 python bench.py --workload socp --cone-dim $r --no-cpu-baseline > $O/r2z_socp$r.json 2> $O/r2z_socp$r.err
done
echo done

#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_kkt.py tests/test_gpu_resident.py tests/test_gpu_fullsize.py tests/test_gpu_round2.py -q -m gpu 2>&1 | tail -4 ) > $O/r3i_tests.log 2>&1
timeout 400 python bench.py --no-cpu-baseline > $O/r3i_bench.json 2> $O/r3i_bench.err
timeout 300 python bench.py --workload socp --no-cpu-baseline > $O/r3i_socp.json 2> $O/r3i_socp.err
echo done

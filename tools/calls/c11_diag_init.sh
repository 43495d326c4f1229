#!/usr/bin/env bash
# round-3 call 11: balanced diagonal tiles in the SYRK, initvals in the device coneqp loop
export PYTHONPATH=.
O=gpurun_out/c11; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_solvers.py tests/test_gpu_batch.py -q -m gpu 2>&1 | tail -8 ) > $O/tests.log 2>&1
timeout 300 python tools/dev/syrk_diag_dev.py > $O/syrk_diag.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --workload batch --steps 3 --warmup 1 --no-cpu-baseline > $O/batch.json 2> $O/batch.err
timeout 600 python bench.py --workload socp --steps 10 --warmup 3 --no-cpu-baseline > $O/socp.json 2> $O/socp.err
echo done

#!/usr/bin/env bash
# round-3 call 8: iterative refinement of the ldl flavours (late-iteration test), tightened full-size bounds, trsv_z with LDS-only barriers
export PYTHONPATH=.
O=gpurun_out/c08; mkdir -p $O
export MI355KKT_PARITY_REPORT=$PWD/$O/parity_report.json
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_kkt.py tests/test_gpu_fullsize.py tests/test_gpu_cvxprog.py tests/test_gpu_solvers.py -q -m gpu 2>&1 | tail -12 ) > $O/tests.log 2>&1
unset MI355KKT_PARITY_REPORT
( MI355KKT_TRSV=z timeout 600 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5 ) > $O/tests_z.log 2>&1
timeout 300 python tools/dev/prof_trsvz_dev.py > $O/trsvz.log 2>&1
for v in z inv; do
MI355KKT_TRSV=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads > $O/bench_$v.json 2> $O/bench_$v.err
done
echo done

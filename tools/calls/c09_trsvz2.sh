#!/usr/bin/env bash
# round-3 call 9: trsv_z with strip register reuse + LDS-only barriers; ldl refinement with two steps
export PYTHONPATH=.
O=gpurun_out/c09; mkdir -p $O
export MI355KKT_PARITY_REPORT=$PWD/$O/parity_report.json
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_kkt.py -q -m gpu 2>&1 | tail -8 ) > $O/tests.log 2>&1
unset MI355KKT_PARITY_REPORT
timeout 300 python tools/dev/prof_trsvz_dev.py > $O/trsvz.log 2>&1
( MI355KKT_TRSV=z timeout 600 python -m pytest tests/test_gpu_kkt.py -x -q -m gpu 2>&1 | tail -5 ) > $O/tests_z.log 2>&1
for v in z inv; do
MI355KKT_TRSV=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads > $O/bench_$v.json 2> $O/bench_$v.err
done
( MI355KKT_ROCTX=1 timeout 300 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k pinned 2>&1 | tail -3 ) > $O/roctx.log 2>&1
echo done

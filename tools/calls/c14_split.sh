#!/usr/bin/env bash
# round-3 call 14: helper tickets (split-K of late tiles) in the dense tile Cholesky
export PYTHONPATH=.
O=gpurun_out/c14; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_ops.py -x -q -m gpu -k potrf 2>&1 | tail -8 ) > $O/tests.log 2>&1
for d in 0 16 12 8 0 16; do
for n in 8192 4096; do
MI355KKT_POTRF_SPLIT=$d timeout 300 python tools/dev/bench_potrf_dev.py $n >> $O/potrf.log 2>&1
done
done
for d in 0 16 0 16; do
MI355KKT_POTRF_SPLIT=$d timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads > $O/bench_$d.json 2> $O/bench_$d.err
python -c "import json; d=json.load(open('$O/bench_$d.json')); print('$d', d['ms_per_step'], d['phases_ms'])" >> $O/summary.log
done
echo done

#!/usr/bin/env bash
# round-3 call 22: SYRK plans of matrices with fewer tiles than slots cut into equal (tile, k) segments: SOCP timing, op tests
export PYTHONPATH=.
O=gpurun_out/c22; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k syrk 2>&1 | tail -3 ) > $O/tests.log 2>&1
for r in 1 2; do
timeout 300 python bench.py --workload socp --steps 20 --warmup 5 --no-cpu-baseline > $O/socp_$r.json 2> $O/socp_$r.err
python -c "import json; d=json.load(open('$O/socp_$r.json')); print(d['ms_per_step'], d['phases_ms'])" >> $O/summary.log
done
echo done

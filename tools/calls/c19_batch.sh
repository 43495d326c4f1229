#!/usr/bin/env bash
# round-3 call 19: batched engine -- residual products in one pass over G, SYRK work items grouped by XCD: tests + A/B
export PYTHONPATH=.
O=gpurun_out/c19; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -5 ) > $O/tests.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5 or batch" 2>&1 | tail -5 ) >> $O/tests.log 2>&1
for cfg in "0 0" "1 0" "0 1" "1 1" "0 0" "1 1"; do
set -- $cfg
MI355KKT_BATCH_XCD=$1 MI355KKT_BATCH_FUSED_PRODUCTS=$2 timeout 300 python bench.py --workload batch --steps 3 --warmup 1 --no-cpu-baseline > $O/b_$1_$2.json 2> $O/b_$1_$2.err
python -c "import json; d=json.load(open('$O/b_$1_$2.json')); print('xcd $1 fused $2', d['value'], d['ms_per_step'])" >> $O/summary.log
done
echo done

#!/usr/bin/env bash
# round-3 call 13: triangular solves with four threads per row (trsv_q_kernel)
export PYTHONPATH=.
O=gpurun_out/c13; mkdir -p $O
( MI355KKT_TRSV=quad timeout 600 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5 ) > $O/tests_q.log 2>&1
for v in quad inv quad inv; do
MI355KKT_TRSV=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads > $O/bench_$v.json 2> $O/bench_$v.err
python -c "import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['ms_per_step'], d['phases_ms'])" >> $O/summary.log
done
for v in quad inv; do
MI355KKT_TRSV=$v timeout 600 python bench.py --workload socp --steps 10 --warmup 3 --no-cpu-baseline > $O/socp_$v.json 2> $O/socp_$v.err
python -c "import json; d=json.load(open('$O/socp_$v.json')); print('socp $v', d['ms_per_step'], d['phases_ms'])" >> $O/summary.log
done
echo done

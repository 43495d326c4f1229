#!/usr/bin/env bash
# round-3 call 7: 32-deep LDS stages in the tile Cholesky's bulk; 16-column streaming inside the sparse engine's fronts
export PYTHONPATH=.
O=gpurun_out/c07; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_sparse.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests.log 2>&1
( MI355KKT_POTRF_BK=32 timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_kkt.py -x -q -m gpu 2>&1 | tail -5 ) > $O/tests_bk32.log 2>&1
{
for n in 8192 4096 2048; do for b in 16 32; do
  MI355KKT_POTRF_BK=$b timeout 300 python tools/dev/bench_potrf_dev.py $n
done; done
} > $O/potrf.log 2>&1
for v in 1 0; do
MI355KKT_SPARSE_STREAM=$v timeout 600 python bench.py --workload sparse --steps 10 --warmup 3 --no-cpu-baseline > $O/sparse46_$v.json 2> $O/sparse46_$v.err
MI355KKT_SPARSE_STREAM=$v timeout 600 python bench.py --workload sparse --grid 64 --steps 6 --warmup 2 --no-cpu-baseline > $O/sparse64_$v.json 2> $O/sparse64_$v.err
done
timeout 300 python tools/dev/prof_trsvz_dev.py > $O/trsvz.log 2>&1
MI355KKT_POTRF_BK=32 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads > $O/bench_bk32.json 2> $O/bench_bk32.err
echo done

#!/usr/bin/env bash
# round-3 call 4: tile Cholesky with 16-column streaming of L(j,j) and of tile (j+1,j) (MI355KKT_POTRF_HALF=2) vs the half-tile hand-off
export PYTHONPATH=.
O=gpurun_out/c04; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k potrf 2>&1 | tail -8 ) > $O/ops_potrf.log 2>&1
{
for n in 8192 4096 2048 1024; do for m in 1 2; do
  MI355KKT_POTRF_HALF=$m timeout 300 python tools/dev/bench_potrf_dev.py $n
done; done
} > $O/potrf.log 2>&1
MI355KKT_POTRF_HALF=2 timeout 300 python tools/dev/prof_tiles_dev.py 8192 > $O/tiles_stream.txt 2>&1
MI355KKT_POTRF_HALF=1 timeout 300 python tools/dev/prof_tiles_dev.py 8192 > $O/tiles_half.txt 2>&1
( timeout 1200 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_ops.py tests/test_gpu_sparse.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests.log 2>&1
for v in 2 1; do
MI355KKT_POTRF_HALF=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads > $O/bench_$v.json 2> $O/bench_$v.err
done
echo done

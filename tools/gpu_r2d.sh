#!/usr/bin/env bash
# round-2 GPU call D: persistent tile Cholesky: correctness at all sizes + timing vs the launch chain
export PYTHONPATH=.
O=gpurun_out
( MI355KKT_TILES_MIN_N=1 timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k potrf 2>&1 | tail -15 ) > $O/r2d_ops_tiles.log 2>&1
( timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -5 ) > $O/r2d_ops.log 2>&1
{
for n in 8192 4096 2048 1024 512 256; do
  MI355KKT_TILES_MIN_N=1 timeout 120 python tools/dev/bench_potrf_dev.py $n 2>&1 | tail -1
  MI355KKT_POTRF=streams timeout 120 python tools/dev/bench_potrf_dev.py $n 2>&1 | tail -1
done
} > $O/r2d_potrf.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r2d_bench.json 2> $O/r2d_bench.err
cat $O/r2d_ops_tiles.log $O/r2d_ops.log $O/r2d_potrf.log; head -c 600 $O/r2d_bench.json

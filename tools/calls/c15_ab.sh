#!/usr/bin/env bash
# round-3 call 15: A/B of the tile Cholesky against the library of the previous commit (build/libbase.so)
export PYTHONPATH=.
O=gpurun_out/c15; mkdir -p $O
for r in 1 2; do
for n in 8192 4096 2048; do
CVXOPT_AMD_LIB=$PWD/build/libbase.so timeout 300 python tools/dev/bench_potrf_dev.py $n 2>&1 | grep potrf | sed 's/^/base /' >> $O/potrf.log
MI355KKT_POTRF_SPLIT=0 timeout 300 python tools/dev/bench_potrf_dev.py $n 2>&1 | grep potrf | sed 's/^/new0 /' >> $O/potrf.log
MI355KKT_POTRF_SPLIT=16 timeout 300 python tools/dev/bench_potrf_dev.py $n 2>&1 | grep potrf | sed 's/^/new16 /' >> $O/potrf.log
done
done
echo done

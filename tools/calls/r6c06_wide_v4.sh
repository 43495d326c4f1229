#!/usr/bin/env bash
# round-6 call 6: wide solves v4 (waits start on one granule of the block)
export PYTHONPATH=.
O=gpurun_out/r6c06; mkdir -p $O
timeout 600 python tools/dev/trsv_wide_dev.py > $O/wide_dev.txt 2>&1
grep -v amdgpu.ids $O/wide_dev.txt
for n in 2048 8192; do
  CVXOPT_AMD_LIB=$PWD/cvxopt_amd/libmi355kkt_debug.so timeout 300 python tools/dev/wide_stamps_dev.py $n > $O/stamps_$n.txt 2>&1
done
head -60 $O/stamps_2048.txt

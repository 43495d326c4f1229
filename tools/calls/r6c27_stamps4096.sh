#!/usr/bin/env bash
# round-6 call 27: stamps of the shipped 512-row solve at n = 4096 (16 rows per workgroup, one slice each)
export PYTHONPATH=.
O=gpurun_out/r6c27; mkdir -p $O
for n in 4096 2048; do
CVXOPT_AMD_LIB=$PWD/cvxopt_amd/libmi355kkt_debug.so timeout 300 python tools/dev/wide_stamps_dev.py $n > $O/stamps_$n.txt 2>&1
done
grep -v amdgpu $O/stamps_4096.txt | sed -n 1,60p

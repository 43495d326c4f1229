#!/usr/bin/env bash
# round-6 call 3: where a 512-row hop's time goes (debug library, 100 MHz stamps per workgroup)
export PYTHONPATH=.
O=gpurun_out/r6c03; mkdir -p $O
for n in 2048 4096; do
  CVXOPT_AMD_LIB=$PWD/cvxopt_amd/libmi355kkt_debug.so timeout 300 python tools/dev/wide_stamps_dev.py $n > $O/stamps_$n.txt 2>&1
done
cat $O/stamps_2048.txt

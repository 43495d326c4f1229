#!/usr/bin/env bash
# round-6 call 10: what a far-field step of the second wave costs (stamps after its wait and after its load request)
export PYTHONPATH=.
O=gpurun_out/r6c10; mkdir -p $O
CVXOPT_AMD_LIB=$PWD/cvxopt_amd/libmi355kkt_debug.so timeout 300 python tools/dev/wide_stamps_dev.py 8192 > $O/stamps_8192.txt 2>&1
grep -v amdgpu $O/stamps_8192.txt | awk '/^block/{b=$2} /entry|Ms in|far/{print b, $0}' | awk '$1>=7'

#!/usr/bin/env bash
# round-6 call 9: wide solves v7 (far field one block behind with d: three rotating strip buffers)
export PYTHONPATH=.
O=gpurun_out/r6c09; mkdir -p $O
timeout 600 python tools/dev/trsv_wide_dev.py > $O/wide_dev.txt 2>&1
grep -v amdgpu.ids $O/wide_dev.txt
for n in 2048 8192; do
  CVXOPT_AMD_LIB=$PWD/cvxopt_amd/libmi355kkt_debug.so timeout 300 python tools/dev/wide_stamps_dev.py $n > $O/stamps_$n.txt 2>&1
done
head -60 $O/stamps_2048.txt
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -m gpu -q > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt

#!/usr/bin/env bash
# round-6 call 11: wide solves v8 = v6 (two strip buffers + diagonal rows) with unconditional loads + ONE-launch MFMA formation of the
# 512 x 512 inverses; A/B probe, solve tests, the headline / SOCP lines
export PYTHONPATH=.
O=gpurun_out/r6c11; mkdir -p $O
timeout 600 python tools/dev/trsv_wide_dev.py > $O/wide_dev.txt 2>&1
grep -v amdgpu.ids $O/wide_dev.txt
for n in 2048 8192; do
  CVXOPT_AMD_LIB=$PWD/cvxopt_amd/libmi355kkt_debug.so timeout 300 python tools/dev/wide_stamps_dev.py $n > $O/stamps_$n.txt 2>&1
done
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py tests/test_gpu_kkt.py tests/test_gpu_stress.py -m gpu -q -x > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
( timeout 600 python bench.py --no-cpu-baseline --no-side-workloads ) > $O/bench_dense.json 2> $O/bench_dense.err
( timeout 300 python bench.py --workload socp --no-cpu-baseline ) > $O/bench_socp.json 2> $O/bench_socp.err
cut -c1-900 $O/bench_dense.json $O/bench_socp.json

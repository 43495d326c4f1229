#!/usr/bin/env bash
# round-5 call 11: the files that exercise the round's last two kernel-side changes (zero-filled inverse blocks beyond a ragged
# block; S's inverses kept across the factorisation of K: two-sweep solves with equality constraints) under the GPU fence
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
O=gpurun_out/r5c11; mkdir -p $O
( MI355KKT_TEST_ALLOC_GUARD=1 MI355KKT_STRESS_ITERS=20 timeout 500 python -m pytest tests/test_gpu_round5.py tests/test_gpu_stress.py tests/test_gpu_kkt.py tests/test_gpu_lifecycle.py -m gpu -q -s -p no:cacheprovider ) > $O/fence.log 2>&1
echo "fence rc=$? $(grep -h 'mi355kkt guard\|passed\|failed' $O/fence.log | tail -2 | tr '\n' ' ')" > $O/summary.txt
cat $O/summary.txt

#!/usr/bin/env bash
# round-6 call 36: the GPU electric fence (every device block ends where its own mapping ends) under the round's new kernels:
# 512-row solves incl. ragged orders and the sparse dense root, formation of the inverses, IPC transport, conditional refinement
export PYTHONPATH=.
O=gpurun_out/r6c36; mkdir -p $O
( MI355KKT_TEST_ALLOC_GUARD=1 MI355KKT_STRESS_ITERS=20 MI355KKT_PIN_CHURN_CYCLES=60 MI355KKT_CHURN_CYCLES=100 timeout 2400 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py tests/test_gpu_stress.py tests/test_gpu_kkt.py tests/test_gpu_sparse.py tests/test_gpu_lifecycle.py tests/test_gpu_churn.py -m gpu -q -s -p no:cacheprovider ) > $O/fence.log 2>&1
echo "rc=$?"; tail -5 $O/fence.log | cut -c1-300
grep -i "overwritten\|guard\|fault" $O/fence.log | tail -5

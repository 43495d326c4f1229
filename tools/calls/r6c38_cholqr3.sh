#!/usr/bin/env bash
# round-6 call 38: shifted CholeskyQR3 where chol(Gs'Gs) breaks down: probe table + solver suites
export PYTHONPATH=.
O=gpurun_out/r6c38; mkdir -p $O
timeout 900 python tests/run_qr_cond_probe.py > $O/qr_probe.txt 2>&1
grep -v amdgpu $O/qr_probe.txt | grep -A7 "n=40" | grep "n=40\|reference qr\|backend qr"
timeout 2400 python -m pytest tests/test_gpu_round6.py tests/test_gpu_reference_examples.py tests/test_gpu_reference_suite.py tests/test_gpu_sdp.py tests/test_gpu_solvers.py tests/test_gpu_resident.py -m gpu -q -x -k "not ipc and not wide and not orders and not info" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt

#!/usr/bin/env bash
# round-6 call 16: the 'qr' mapping with conditional refinement: probe table + tests; reference-example suites (they run 'qr' by default for socp / sdp)
export PYTHONPATH=.
O=gpurun_out/r6c16; mkdir -p $O
timeout 900 python tests/run_qr_cond_probe.py > $O/qr_probe.txt 2>&1
grep -v amdgpu $O/qr_probe.txt | head -50
timeout 2400 python -m pytest tests/test_gpu_round6.py tests/test_gpu_reference_examples.py tests/test_gpu_reference_suite.py tests/test_gpu_sdp.py tests/test_gpu_solvers.py tests/test_gpu_resident.py -m gpu -q -x > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt

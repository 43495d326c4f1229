#!/usr/bin/env bash
# round-6 call 23: ragged orders through the 512-row solves by default: the dense suites that factor n >= 1024
export PYTHONPATH=.
O=gpurun_out/r6c23; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_kkt.py tests/test_gpu_stress.py tests/test_gpu_fullsize.py tests/test_gpu_round3.py tests/test_gpu_round2.py tests/test_gpu_maxsize.py -m gpu -q -x > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt

#!/usr/bin/env bash
# round-6 call 2: first run of the 512-row all-CU triangular solves (trsv512.hip): A/B probe + the solve tests
export PYTHONPATH=.
O=gpurun_out/r6c02; mkdir -p $O
timeout 600 python tools/dev/trsv_wide_dev.py > $O/wide_dev.txt 2>&1
tail -30 $O/wide_dev.txt
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_kkt.py -m gpu -x -q > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt

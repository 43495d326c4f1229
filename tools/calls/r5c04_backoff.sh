#!/usr/bin/env bash
# round-5 call 4: distance-proportional poll backoff in the triangular solves (every workgroup of a launch waits for the same block, the
# front's): two-sweep kernel without / with backoff, wide kernel with backoff; the round's solve tests
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
O=gpurun_out/r5c04; mkdir -p $O
( timeout 300 python tools/dev/trsv_wide_dev.py ) > $O/trsv_backoff.log 2>&1
grep -v amdgpu.ids $O/trsv_backoff.log > $O/summary.txt
( timeout 400 python -m pytest tests/test_gpu_round5.py tests/test_gpu_stress.py -m gpu -q -x -p no:cacheprovider ) > $O/tests.log 2>&1
echo "tests rc=$? $(tail -1 $O/tests.log | cut -c1-200)" >> $O/summary.txt
cat $O/summary.txt

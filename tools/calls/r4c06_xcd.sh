#!/usr/bin/env bash
# round-4 call 6: XCD-grouped placement and L2-local polling of the two-sweep triangular solves
export PYTHONPATH=.
O=gpurun_out/r4c06; mkdir -p $O
( timeout 300 python tools/dev/trsv_xcd_dev.py ) > $O/xcd.log 2>&1
grep -v amdgpu.ids $O/xcd.log

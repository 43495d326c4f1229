#!/usr/bin/env bash
# round-6 call 12: formation of the 512 x 512 inverses with 64-column operand chunks (A/B probe only)
export PYTHONPATH=.
O=gpurun_out/r6c12; mkdir -p $O
timeout 600 python tools/dev/trsv_wide_dev.py > $O/wide_dev.txt 2>&1
grep -v amdgpu.ids $O/wide_dev.txt | grep factor

#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
timeout 300 python tests/sdp_ops_dev.py > $O/r3c.log 2>&1
echo done

#!/usr/bin/env bash
# timing of the batched engine with second-order cones, then the validation run of the final build (tools/gpu_r3f.sh)
export PYTHONPATH=.
O=gpurun_out
timeout 300 python tools/dev/bench_batch_q_dev.py > $O/r3m_batch_q.log 2>&1
bash tools/gpu_r3f.sh

#!/usr/bin/env bash
# round-6 call 15: pinned-upload churn test, the kkt_qr conditioning probe (reference 'qr' / 'chol' / 'ldl' against the backend's
# mappings), the sharded bench line at world size 1 with --transport auto / ipc
export PYTHONPATH=.
O=gpurun_out/r6c15; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_churn.py -m gpu -q -x -k pinned > $O/pytest_churn.txt 2>&1
tail -5 $O/pytest_churn.txt
timeout 900 python tests/run_qr_cond_probe.py > $O/qr_probe.txt 2>&1
grep -v amdgpu $O/qr_probe.txt | tail -100
( timeout 600 python bench.py --workload sharded --gpus 1 --total-batch 1024 --steps 3 --warmup 1 --no-cpu-baseline ) > $O/bench_sharded.json 2> $O/bench_sharded.err
cut -c1-1200 $O/bench_sharded.json; tail -3 $O/bench_sharded.err

#!/usr/bin/env bash
# round-6 call 24: IPC transport after keeping the mapped views on the owning device; sharded bench line at world size 1 (auto)
export PYTHONPATH=.
O=gpurun_out/r6c24; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -x -k ipc > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29562 bench.py --workload sharded --gpus 1 --total-batch 1024 --steps 3 --warmup 1 ) > $O/bench_sharded.json 2> $O/bench_sharded.err
cut -c1-700 $O/bench_sharded.json

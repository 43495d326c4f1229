#!/usr/bin/env bash
# round-6 call 17: the sharded bench line at world size 1 under torch.distributed.run, --transport auto (-> ipc) and rccl
export PYTHONPATH=.
O=gpurun_out/r6c17; mkdir -p $O
for t in auto rccl; do
  ( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --workload sharded --gpus 1 --total-batch 1024 --steps 3 --warmup 1 --transport $t ) > $O/bench_sharded_$t.json 2> $O/bench_sharded_$t.err
  cut -c1-1500 $O/bench_sharded_$t.json; tail -3 $O/bench_sharded_$t.err
done

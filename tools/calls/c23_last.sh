#!/usr/bin/env bash
# round-3 call 23: the committed tree once more -- full GPU suite, smoke, the N-rank line at world size 1 with the whole config-5 batch
# (4096 problems resident on one GPU: the root's memory footprint of the 8-GPU run), the default bench line
export PYTHONPATH=.
O=gpurun_out/c23; mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 ) > $O/tests.log 2>&1
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) > $O/smoke.log 2>&1
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --workload sharded --total-batch 4096 --steps 2 --warmup 1 > $O/sharded4096.json 2> $O/sharded4096.err
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
echo done

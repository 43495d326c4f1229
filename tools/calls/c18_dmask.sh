#!/usr/bin/env bash
# round-3 call 18: block masks on the diagonal SYRK tiles of short contractions (the batch): tests, A/B, headline unchanged?
export PYTHONPATH=.
O=gpurun_out/c18; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -5 ) > $O/tests.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5 or batch" 2>&1 | tail -5 ) >> $O/tests.log 2>&1
timeout 600 python tools/dev/bench_batch_mask_dev.py 512 > $O/mask.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads > $O/bench.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['phases_ms'])" > $O/summary.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --workload sharded --total-batch 1024 --steps 2 --warmup 1 > $O/sharded.json 2> $O/sharded.err
echo done

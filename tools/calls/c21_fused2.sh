#!/usr/bin/env bash
# round-3 call 21: fused residual products with two columns in flight per trip; rebuild without warnings (unused variables removed): tests + batch timing
export PYTHONPATH=.
O=gpurun_out/c21; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_batch.py tests/test_gpu_resident.py tests/test_gpu_sparse.py -x -q -m gpu 2>&1 | tail -5 ) > $O/tests.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "potrf or trsm" 2>&1 | tail -3 ) >> $O/tests.log 2>&1
for r in 1 2; do
timeout 300 python bench.py --workload batch --steps 3 --warmup 1 --no-cpu-baseline > $O/b_$r.json 2> $O/b_$r.err
python -c "import json; d=json.load(open('$O/b_$r.json')); print(d['value'], d['ms_per_step'])" >> $O/summary.log
done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_batch -o r03b -- python $GRAFT_REPO_ROOT/bench.py --workload batch --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_batch.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_batch -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r03_batch_kernel_stats.md > /dev/null 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads > $O/bench.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['phases_ms'], d['roofline']['traffic'])" >> $O/summary.log
echo done

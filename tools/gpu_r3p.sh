#!/usr/bin/env bash
# rocprofv3 kernel statistics of the two batched workloads: config 5 (LP cone) and the batch with second-order cones
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_b5 -o b5 -- python $GRAFT_REPO_ROOT/bench.py --workload batch --steps 3 --warmup 1 --no-cpu-baseline > $O/r3p_b5.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_bq -o bq -- python $GRAFT_REPO_ROOT/tools/dev/bench_batch_q_dev.py > $O/r3p_bq.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py stats $(find /tmp/prof_b5 -name '*results.db' | head -1) $O/r02_batch_kernel_stats.md > /dev/null 2>&1
python tools/rocpd_summary.py stats $(find /tmp/prof_bq -name '*results.db' | head -1) $O/r02_batch_socp_kernel_stats.md > /dev/null 2>&1
echo done

#!/usr/bin/env bash
# rocprofv3 kernel statistics of the batch with second-order cones after the scaling kernels were tuned
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof_bq -o bq -- python $GRAFT_REPO_ROOT/tools/dev/bench_batch_q_dev.py > $O/r3r_bq.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py stats $(find /tmp/prof_bq -name '*results.db' | head -1) $O/r02_batch_socp_kernel_stats.md > /dev/null 2>&1
echo done

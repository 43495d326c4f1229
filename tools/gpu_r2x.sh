#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_sp -o sp -- python $GRAFT_REPO_ROOT/bench.py --workload sparse --no-cpu-baseline --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/r2x_prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_sp -name '*results.db' | head -1)
python tools/sparse_timeline.py $DB > $O/r2x_sparse_timeline.txt 2>&1
echo done

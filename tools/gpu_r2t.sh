#!/usr/bin/env bash
export PYTHONPATH=.
O=gpurun_out
( timeout 600 python -m pytest tests/test_gpu_sdp.py -q -m gpu 2>&1 | tail -8 ) > $O/r2t_sdp.log 2>&1
timeout 600 python tools/dev/sdp_time_dev.py 20 60 100 200 > $O/r2t_time.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sdp -o sdp -- python $GRAFT_REPO_ROOT/tools/dev/sdp_time_dev.py 100 > $GRAFT_REPO_ROOT/$O/r2t_prof.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_sdp -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r02_sdp_mc100_kernel_stats.md > /dev/null 2>&1
echo done

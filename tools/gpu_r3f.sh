#!/usr/bin/env bash
# final validation of the round-2 build: full GPU suite, smoke, bench lines, refreshed rocprof summaries
export PYTHONPATH=.
O=gpurun_out
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) > $O/r3f_tests.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/r3f_smoke.log 2>&1
timeout 400 python bench.py > $O/r3f_bench.json 2> $O/r3f_bench.err
timeout 300 python bench.py --workload socp --no-cpu-baseline > $O/r3f_socp.json 2> $O/r3f_socp.err
timeout 300 python bench.py --workload sparse --no-cpu-baseline --steps 10 > $O/r3f_sparse.json 2> $O/r3f_sparse.err
timeout 300 python bench.py --workload sdp --steps 10 > $O/r3f_sdp.json 2> $O/r3f_sdp.err
timeout 300 python bench.py --workload batch --no-cpu-baseline > $O/r3f_batch.json 2> $O/r3f_batch.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_dense -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/r3f_prof_dense.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_dense -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r02_kernel_stats.md > /dev/null 2>&1
python tools/potrf_timeline.py $DB 3 70 > $O/r02_step_timeline.txt 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_socp -o r02s -- python $GRAFT_REPO_ROOT/bench.py --workload socp --steps 4 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/r3f_prof_socp.log 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_socp -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r02_socp8_kernel_stats.md > /dev/null 2>&1
echo done

#!/usr/bin/env bash
# round-4 call 15: step timelines (rocprofv3 --kernel-trace) of the SOCP workload, the headline and one GPU's share of the batch
export PYTHONPATH=.
R=$PWD
O=gpurun_out/r4c15; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_socp -o q -- python $R/bench.py --workload socp --steps 6 --warmup 2 --no-cpu-baseline > $R/$O/prof_socp.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_dense -o d -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-side-workloads > $R/$O/prof_dense.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_batch -o b -- python $R/bench.py --workload batch --steps 1 --warmup 1 --no-cpu-baseline > $R/$O/prof_batch.log 2>&1
cd $R
DB=$(find /tmp/prof_socp -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r04_socp8_kernel_stats.md > /dev/null 2>&1
python tools/potrf_timeline.py $DB 4 140 > $O/r04_socp8_step_timeline.txt 2>&1
DB=$(find /tmp/prof_dense -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r04_kernel_stats.md > /dev/null 2>&1
python tools/potrf_timeline.py $DB 3 70 > $O/r04_step_timeline.txt 2>&1
DB=$(find /tmp/prof_batch -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r04_batch_kernel_stats.md > /dev/null 2>&1
python tools/potrf_timeline.py $DB 12 160 > $O/r04_batch_iteration_timeline.txt 2>&1
tail -3 $O/prof_socp.log $O/prof_dense.log $O/prof_batch.log

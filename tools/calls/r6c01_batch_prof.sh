#!/usr/bin/env bash
# round-6 call 1: fresh evidence for the batched engine BEFORE touching it: kernel statistics, the lock-step
# iteration's timeline, and FETCH_SIZE / WRITE_SIZE per kernel name (each counter in its own pass); plus the headline and the SOCP
# line on today's box as the round's starting numbers
export PYTHONPATH=.
R=$PWD
O=gpurun_out/r6c01; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BB="python $R/bench.py --workload batch --steps 1 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_batch -o b -- $BB > $R/$O/prof_batch.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_bf -o f -- $BB > $R/$O/pmc_bf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_bw -o w -- $BB > $R/$O/pmc_bw.log 2>&1
cd $R
DB=$(find /tmp/prof_batch -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r06_batch_kernel_stats_before.md > /dev/null 2>&1
python tools/potrf_timeline.py $DB 12 200 > $O/r06_batch_iteration_timeline_before.txt 2>&1
FD=$(find /tmp/pmc_bf -name '*results.db' | head -1); WD=$(find /tmp/pmc_bw -name '*results.db' | head -1)
python tools/rocpd_summary.py pmctable $FD $WD $O/r06_batch_pmc_traffic_before.md > $O/pmctable.log 2>&1
( timeout 600 python bench.py --no-cpu-baseline --no-side-workloads ) > $O/bench_dense.json 2> $O/bench_dense.err
( timeout 300 python bench.py --workload socp --no-cpu-baseline ) > $O/bench_socp.json 2> $O/bench_socp.err
( timeout 300 python bench.py --workload batch --no-cpu-baseline ) > $O/bench_batch.json 2> $O/bench_batch.err
( timeout 300 python bench.py --workload sparse --no-cpu-baseline ) > $O/bench_sparse.json 2> $O/bench_sparse.err
tail -2 $O/prof_batch.log; head -20 $O/r06_batch_kernel_stats_before.md; head -16 $O/r06_batch_pmc_traffic_before.md
cut -c1-600 $O/bench_dense.json $O/bench_socp.json $O/bench_batch.json $O/bench_sparse.json

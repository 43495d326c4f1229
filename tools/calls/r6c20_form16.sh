#!/usr/bin/env bash
# round-6 call 20: formation with sixteen waves per 64 x 64 tile, on the side stream beside the mirror; sparse dense root through the wide solves
export PYTHONPATH=.
R=$PWD
O=gpurun_out/r6c20; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py tests/test_gpu_stress.py tests/test_gpu_kkt.py -m gpu -q -x > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
for n in 2048 8192; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_w$n -o w -- python $R/tools/dev/wide_prof_dev.py $n > $R/$O/prof_$n.log 2>&1
  DB=$(find /tmp/prof_w$n -name '*results.db' | head -1)
  python $R/tools/rocpd_summary.py stats $DB $R/$O/wide_kernel_stats_$n.md > /dev/null 2>&1
  grep "inverse512\|trsv_wide\|mirror" $R/$O/wide_kernel_stats_$n.md; tail -1 $R/$O/prof_$n.log
done
cd $R
( timeout 600 python bench.py --no-cpu-baseline --no-side-workloads ) > $O/bench_dense.json 2> $O/bench_dense.err
( timeout 300 python bench.py --workload socp --no-cpu-baseline ) > $O/bench_socp.json 2> $O/bench_socp.err
cut -c1-700 $O/bench_dense.json $O/bench_socp.json

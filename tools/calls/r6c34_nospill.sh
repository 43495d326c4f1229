#!/usr/bin/env bash
# round-6 call 34: 16 rows per workgroup without the early t_0 (no scratch): durations + solve tests
export PYTHONPATH=.
R=$PWD
O=gpurun_out/r6c34; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py tests/test_gpu_stress.py -m gpu -q -x -k "not ipc and not churn and not qr and not info" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
for n in 4096 8192; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_w$n -o w -- python $R/tools/dev/wide_prof_dev.py $n > $R/$O/prof_$n.log 2>&1
  DB=$(find /tmp/prof_w$n -name '*results.db' | head -1)
  python $R/tools/rocpd_summary.py stats $DB $R/$O/wide_kernel_stats_$n.md > /dev/null 2>&1
  grep "trsv_wide" $R/$O/wide_kernel_stats_$n.md | cut -c1-150
done

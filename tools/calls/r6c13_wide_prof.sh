#!/usr/bin/env bash
# round-6 call 13: kernel durations of the wide solves and the formation of their inverses (rocprofv3 kernel trace), n = 2048 / 8192;
# then the IPC-transport test of the sharded batch
export PYTHONPATH=.
R=$PWD
O=gpurun_out/r6c13; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 2048 8192; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_w$n -o w -- python $R/tools/dev/wide_prof_dev.py $n > $R/$O/prof_$n.log 2>&1
  DB=$(find /tmp/prof_w$n -name '*results.db' | head -1)
  python $R/tools/rocpd_summary.py stats $DB $R/$O/wide_kernel_stats_$n.md > /dev/null 2>&1
  head -14 $R/$O/wide_kernel_stats_$n.md
done
cd $R
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_stress.py -m gpu -q -x > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt

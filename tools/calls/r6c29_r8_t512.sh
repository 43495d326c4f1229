#!/usr/bin/env bash
# round-6 call 29: 512 threads for the 8-row shape too (8 columns per thread); 8 rows at every order with two workgroups per compute
# unit (knob value 8) against 16 rows: tests + kernel durations
export PYTHONPATH=.
R=$PWD
O=gpurun_out/r6c29; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py tests/test_gpu_stress.py -m gpu -q -x -k "not ipc and not churn" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
for cfg in "2048 1" "4096 1" "4096 8" "8192 1" "8192 8"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_w$1_$2 -o w -- python $R/tools/dev/wide_prof_dev.py $1 $2 > $R/$O/prof_$1_$2.log 2>&1
  DB=$(find /tmp/prof_w$1_$2 -name '*results.db' | head -1)
  python $R/tools/rocpd_summary.py stats $DB $R/$O/wide_kernel_stats_$1_$2.md > /dev/null 2>&1
  echo "n=$1 knob=$2"; grep "trsv_wide" $R/$O/wide_kernel_stats_$1_$2.md | cut -c1-140
done

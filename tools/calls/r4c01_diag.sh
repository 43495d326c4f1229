#!/usr/bin/env bash
# round-4 call 1: diagnosis of the intermittent core dump of the full GPU suite.
# Full suite three times with the WHOLE log kept, faulthandler on, core dumps enabled + rocgdb backtrace of any core,
# AMD_LOG_LEVEL raised on a rerun of the failing test; then the soak test at 2000 repetitions.
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
R=$PWD
O=gpurun_out/r4c01; mkdir -p $O
ulimit -c unlimited
echo "core_pattern: $(cat /proc/sys/kernel/core_pattern)" > $O/env.txt
rocm-smi --showproductname --showclocks >> $O/env.txt 2>&1
for i in 1 2; do
  rm -f core core.* /tmp/core*
  ( cd $R && timeout 900 python -X faulthandler -m pytest tests -q -m gpu -p no:cacheprovider -x --deselect tests/test_gpu_stress.py ) > $O/suite_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc last_test=$(cat gpurun_out/pytest_last_test.txt 2>/dev/null)" >> $O/summary.txt
  tail -3 $O/suite_$i.log >> $O/summary.txt
  if [ $rc -ne 0 ]; then
    dmesg 2>/dev/null | tail -40 > $O/dmesg_$i.txt
    for c in core core.* /tmp/core*; do
      if [ -f "$c" ]; then
        timeout 300 /opt/rocm/bin/rocgdb -batch -ex "thread apply all bt" "$(which python)" "$c" > $O/core_bt_$i.txt 2>&1
        break
      fi
    done
  fi
done
( MI355KKT_STRESS_ITERS=2000 timeout 900 python -X faulthandler -m pytest tests/test_gpu_stress.py -q -x -p no:cacheprovider ) > $O/stress.log 2>&1
echo "stress rc=$?" >> $O/summary.txt
tail -3 $O/stress.log >> $O/summary.txt
cat $O/summary.txt

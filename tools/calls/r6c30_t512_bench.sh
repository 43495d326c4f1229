#!/usr/bin/env bash
# round-6 call 30: the 512-thread shapes of the 512-row solves: dense suites, stamps, kernel durations, the bench lines
export PYTHONPATH=.
R=$PWD
O=gpurun_out/r6c30; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_kkt.py tests/test_gpu_stress.py tests/test_gpu_fullsize.py tests/test_gpu_sparse.py -m gpu -q -x > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
for n in 2048 8192; do
  CVXOPT_AMD_LIB=$PWD/cvxopt_amd/libmi355kkt_debug.so timeout 300 python tools/dev/wide_stamps_dev.py $n > $O/stamps_$n.txt 2>&1
done
timeout 600 python tools/dev/trsv_wide_dev.py > $O/wide_dev.txt 2>&1
grep -v amdgpu $O/wide_dev.txt | grep "solve"
cd /tmp && export TMPDIR=/tmp
for n in 2048 4096 8192; do
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_w$n -o w -- python $R/tools/dev/wide_prof_dev.py $n > $R/$O/prof_$n.log 2>&1
  DB=$(find /tmp/prof_w$n -name '*results.db' | head -1)
  python $R/tools/rocpd_summary.py stats $DB $R/$O/wide_kernel_stats_$n.md > /dev/null 2>&1
done
cd $R
( timeout 900 python bench.py --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6c30/bench.json").read().strip().splitlines()[-1])
print("headline", d["ms_per_step"], d["phases_ms"], d["roofline"]["frac"])
for k, v in d.get("side_workloads", {}).items():
    print(k, v.get("ms_per_step"), v.get("value"), v.get("phases_ms"), (v.get("roofline") or {}).get("frac"))
PY

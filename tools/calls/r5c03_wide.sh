#!/usr/bin/env bash
# round-5 call 3: the wide triangular solves (trsv_wide.hip) and the matrix-core micro panel of potf2 (potrf.hip, v2):
# parity (new test file + the suites that go through factor / solve), soak, timings against the round-4 kernels, stamps, bench
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
R=$PWD
O=gpurun_out/r5c03; mkdir -p $O
( timeout 300 python tools/dev/pin_fault_dev.py ) > $O/pin_fault.log 2>&1
cat $O/pin_fault.log > $O/summary.txt
( timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_ops.py tests/test_gpu_kkt.py tests/test_gpu_stress.py tests/test_gpu_sparse.py \
    tests/test_gpu_resident.py -m gpu -q -x -p no:cacheprovider ) > $O/tests.log 2>&1
TRC=$?
echo "tests rc=$TRC $(tail -1 $O/tests.log | cut -c1-200)" >> $O/summary.txt
grep -h "^FAILED\|^ERROR\|Error\|assert" $O/tests.log | head -20 >> $O/summary.txt
( timeout 300 python tools/dev/trsv_wide_dev.py ) > $O/trsv_wide.log 2>&1
cat $O/trsv_wide.log >> $O/summary.txt
for n in 1024 2048 4096 8192; do ( timeout 120 python tools/dev/bench_potrf_dev.py $n ) >> $O/potrf.log 2>&1; done
grep potrf $O/potrf.log >> $O/summary.txt
( CVXOPT_AMD_LIB=$R/cvxopt_amd/libmi355kkt_debug.so timeout 120 python tools/dev/prof_tiles_dev.py 2048 ) > $O/tiles_2048.log 2>&1
head -22 $O/tiles_2048.log >> $O/summary.txt
if [ $TRC != 0 ]; then tail -60 $O/tests.log | cut -c1-250 >> $O/summary.txt; cat $O/summary.txt; exit 0; fi
( MI355KKT_PARITY_REPORT=$R/$O/parity_report.json timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider ) > $O/fullsize.log 2>&1
echo "fullsize rc=$? $(tail -1 $O/fullsize.log | cut -c1-200)" >> $O/summary.txt
( timeout 900 python bench.py --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/summary.txt
python - <<'PY' >> gpurun_out/r5c03/summary.txt
import json
try:
    d = json.load(open("gpurun_out/r5c03/bench.json"))
    print("headline", d["ms_per_step"], d["phases_ms"], "hook", d.get("hook_ms_per_step"), "roofline", d["roofline"]["frac"])
    for k, v in d.get("side_workloads", {}).items():
        print(k, v.get("ms_per_step"), v.get("value"), (v.get("roofline") or {}).get("frac"), v.get("phases_ms"), v.get("error"))
except Exception as e:
    print("bench parse error", e)
PY
cat $O/summary.txt

#!/usr/bin/env bash
# round-4 call 2: the new parity tests (options as inputs, the reference's own test-suite through the backend), the soak test at
# 2000 repetitions on the library with host-synchronous copies, then the full GPU suite once (whole log kept) and a short headline
# bench (SYRK time after the ablation switch became a compile-time constant).
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
O=gpurun_out/r4c02; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_options.py tests/test_gpu_reference_suite.py -q -p no:cacheprovider ) > $O/new_tests.log 2>&1
echo "new tests rc=$?" > $O/summary.txt; tail -4 $O/new_tests.log >> $O/summary.txt
( MI355KKT_STRESS_ITERS=2000 timeout 900 python -m pytest tests/test_gpu_stress.py -q -p no:cacheprovider ) > $O/stress.log 2>&1
echo "stress rc=$?" >> $O/summary.txt; tail -4 $O/stress.log >> $O/summary.txt
( timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_stress.py ) > $O/suite.log 2>&1
echo "suite rc=$? last_test=$(cat gpurun_out/pytest_last_test.txt 2>/dev/null)" >> $O/summary.txt; tail -4 $O/suite.log >> $O/summary.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads > $O/bench.json 2> $O/bench.err
python - <<'PY' >> gpurun_out/r4c02/summary.txt
import json
try:
    d = json.load(open("gpurun_out/r4c02/bench.json"))
    print("bench ms_per_step", d["ms_per_step"], "phases", d["phases_ms"], "hook", d["hook_ms_per_step"], "frac", d["roofline"]["frac"])
except Exception as e:
    print("bench parse error", e)
PY
cat $O/summary.txt

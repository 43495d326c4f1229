#!/usr/bin/env bash
# round-4 call 7: the reference's examples/book through the backend (results compared), the full GPU suite (whole log, parity
# report), smoke, and the driver's command: python bench.py (headline + side workloads + CPU baselines)
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
O=gpurun_out/r4c07; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_reference_examples.py -q -s -p no:cacheprovider ) > $O/book.log 2>&1
echo "book rc=$?" > $O/summary.txt; grep -E "matrices|passed|failed" $O/book.log | tail -30 >> $O/summary.txt
export MI355KKT_PARITY_REPORT=$PWD/$O/parity_report.json
( timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_reference_examples.py ) > $O/suite.log 2>&1
echo "suite rc=$? last_test=$(cat gpurun_out/pytest_last_test.txt 2>/dev/null)" >> $O/summary.txt; tail -3 $O/suite.log >> $O/summary.txt
unset MI355KKT_PARITY_REPORT
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1
tail -2 $O/smoke.log >> $O/summary.txt
( timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/summary.txt
python - <<'PY' >> gpurun_out/r4c07/summary.txt
import json
try:
    d = json.load(open("gpurun_out/r4c07/bench.json"))
    print("headline", d["ms_per_step"], d["phases_ms"], "hook", d["hook_ms_per_step"], "frac", d["roofline"]["frac"], "cpu", d.get("cpu_baseline", {}).get("value"))
    for k, v in d.get("side_workloads", {}).items():
        print(k, v.get("ms_per_step"), v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("cpu_baseline") or {}).get("value"), v.get("error"))
except Exception as e:
    print("bench parse error", e)
PY
cat $O/summary.txt

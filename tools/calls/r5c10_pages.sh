#!/usr/bin/env bash
# round-5 call 10: set_H_dense_async pins only whole pages that belong to H alone (page-aligned interior; the partial pages at the ends
# are copied synchronously).  The whole GPU suite once more in ONE process (the driver's command: a second sample of the
# one-process run of the final build) and the headline with its hook-boundary leg (the path that uses the pinned upload).
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
O=gpurun_out/r5c10; mkdir -p $O
( timeout 1200 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider ) > $O/suite.log 2>&1
echo "suite rc=$? $(tail -1 $O/suite.log | cut -c1-200)" > $O/summary.txt
grep -h "^FAILED\|^ERROR" $O/suite.log | head -10 >> $O/summary.txt
( timeout 400 python bench.py --no-cpu-baseline --no-side-workloads ) > $O/bench.json 2> $O/bench.err
python - <<'PY' >> gpurun_out/r5c10/summary.txt
import json
try:
    d = json.load(open("gpurun_out/r5c10/bench.json"))
    print("headline", d["ms_per_step"], "hook", d.get("hook_ms_per_step"), d.get("hook"))
except Exception as e:
    print("bench parse error", e)
PY
cat $O/summary.txt

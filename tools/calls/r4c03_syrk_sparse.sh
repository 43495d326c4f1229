#!/usr/bin/env bash
# round-4 call 3: (a) the reference's own suite through cvxopt_amd.solvers after the solver='default' fix, (b) config-4 class at
# 64^3 / 100^3 against the independent reference solutions, (c) the SYRK k-split sweep at small n, (d) the new side-workload lines
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
O=gpurun_out/r4c03; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_reference_suite.py tests/test_gpu_sparse_big.py -q -p no:cacheprovider ) > $O/tests.log 2>&1
echo "tests rc=$?" > $O/summary.txt; tail -5 $O/tests.log >> $O/summary.txt
( timeout 300 python tools/dev/syrk_split_dev.py ) > $O/syrk_split.log 2>&1
cat $O/syrk_split.log >> $O/summary.txt
( timeout 600 python bench.py --workload socp --steps 20 --warmup 3 --no-cpu-baseline ) > $O/bench_socp.json 2> $O/bench_socp.err
( timeout 600 python bench.py --workload sparse --steps 10 --warmup 2 --no-cpu-baseline ) > $O/bench_sparse.json 2> $O/bench_sparse.err
python - <<'PY' >> gpurun_out/r4c03/summary.txt
import json
for f in ("bench_socp", "bench_sparse"):
    try:
        d = json.load(open("gpurun_out/r4c03/%s.json" % f))
        print(f, d["ms_per_step"], d.get("phases_ms"), (d.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(f, "parse error", e)
PY
cat $O/summary.txt

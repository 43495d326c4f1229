#!/usr/bin/env bash
# round-4 call 13: sparse solves with the shortened in-LDS substitution chains (reciprocal off the chain, chain length = block
# width): parity tests of the sparse engine, bench lines at 64^3 and 46^3 (before: solves 3.3 / 1.7 ms)
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
O=gpurun_out/r4c13; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_sparse_big.py -q -x -p no:cacheprovider ) > $O/tests.log 2>&1
echo "tests rc=$?" > $O/summary.txt; tail -4 $O/tests.log >> $O/summary.txt
( timeout 400 python bench.py --workload sparse --steps 10 --warmup 2 --no-cpu-baseline ) > $O/bench_sparse64.json 2> $O/bench_sparse64.err
( timeout 400 python bench.py --workload sparse --grid 46 --steps 10 --warmup 2 --no-cpu-baseline ) > $O/bench_sparse46.json 2> $O/bench_sparse46.err
python - <<'PY' >> gpurun_out/r4c13/summary.txt
import json
for f in ("bench_sparse64", "bench_sparse46"):
    try:
        d = json.load(open("gpurun_out/r4c13/%s.json" % f))
        print(f, d["ms_per_step"], d.get("phases_ms"), (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("solve"))
    except Exception as e:
        print(f, "parse error", e)
PY
cat $O/summary.txt

#!/usr/bin/env bash
# round-4 call 14: sparse solves with more loads in flight (remainder products, children gathers, four columns per wave in the
# backward products): parity tests, bench lines at 64^3 / 46^3, rocprofv3 kernel statistics + timeline of the 64^3 run
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
R=$PWD
O=gpurun_out/r4c14; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_sparse_big.py -q -x -p no:cacheprovider ) > $O/tests.log 2>&1
echo "tests rc=$?" > $O/summary.txt; tail -4 $O/tests.log >> $O/summary.txt
( timeout 400 python bench.py --workload sparse --steps 10 --warmup 2 --no-cpu-baseline ) > $O/bench_sparse64.json 2> $O/bench_sparse64.err
( timeout 400 python bench.py --workload sparse --grid 46 --steps 10 --warmup 2 --no-cpu-baseline ) > $O/bench_sparse46.json 2> $O/bench_sparse46.err
python - <<'PY' >> gpurun_out/r4c14/summary.txt
import json
for f in ("bench_sparse64", "bench_sparse46"):
    try:
        d = json.load(open("gpurun_out/r4c14/%s.json" % f))
        print(f, d["ms_per_step"], d.get("phases_ms"), (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("solve"))
    except Exception as e:
        print(f, "parse error", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sparse -o r04s -- python $R/bench.py --workload sparse --steps 4 --warmup 2 --no-cpu-baseline > $R/$O/prof_sparse.log 2>&1
cd $R
DB=$(find /tmp/prof_sparse -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r04_sparse64_kernel_stats.md > /dev/null 2>&1
python tools/sparse_timeline.py $DB > $O/r04_sparse64_timeline.txt 2>&1
cat $O/summary.txt

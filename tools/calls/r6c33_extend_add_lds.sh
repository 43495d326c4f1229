#!/usr/bin/env bash
# round-6 call 33: extend-add with multi-child tiles accumulated in LDS: sparse suites (incl. the NaN-poisoned store), bench lines, kernel stats
export PYTHONPATH=.
R=$PWD
O=gpurun_out/r6c33; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_sparse_big.py tests/test_gpu_round6.py tests/test_gpu_fullsize.py -m gpu -q -x -k "sparse or config4 or elasticity or config3" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sparse -o s -- python $R/bench.py --workload sparse --steps 4 --warmup 2 --no-cpu-baseline > $R/$O/prof_sparse.log 2>&1
cd $R
DB=$(find /tmp/prof_sparse -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/sparse64_kernel_stats.md > /dev/null 2>&1
python tools/sparse_timeline.py $DB > $O/sparse64_timeline.txt 2>&1
grep "last factor" -A6 $O/sparse64_timeline.txt
( timeout 900 python bench.py --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6c33/bench.json").read().strip().splitlines()[-1])
for k, v in d.get("side_workloads", {}).items():
    if 'sparse' in k: print(k, v.get("ms_per_step"), v.get("phases_ms"), (v.get("roofline") or {}).get("frac"))
PY

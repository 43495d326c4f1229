#!/usr/bin/env bash
# round-4 call 4: same-box A/B of the round-3 library (libmi355kkt_r3.so, built from 8b19aea) against the current one on the
# latency-bound kernels (potrf at small n, solve), the XCD-grouped trsv mapping experiment, config 5 on one GPU
export PYTHONPATH=.
O=gpurun_out/r4c04; mkdir -p $O
for rep in 1 2; do
  ( timeout 200 python tools/dev/ab_lib_dev.py cvxopt_amd/libmi355kkt_r3.so r3 ) >> $O/ab.log 2>&1
  ( timeout 200 python tools/dev/ab_lib_dev.py cvxopt_amd/libmi355kkt.so r4 ) >> $O/ab.log 2>&1
done
( timeout 300 python tools/dev/trsv_xmap_dev.py ) > $O/xmap.log 2>&1
( timeout 300 python bench.py --workload batch --steps 3 --warmup 1 --no-cpu-baseline ) > $O/bench_batch.json 2> $O/bench_batch.err
grep -v amdgpu.ids $O/ab.log; grep -v amdgpu.ids $O/xmap.log
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r4c04/bench_batch.json"))
    print("batch", d["value"], d["ms_per_step"], d.get("roofline", {}).get("frac"))
except Exception as e:
    print("batch parse error", e)
PY

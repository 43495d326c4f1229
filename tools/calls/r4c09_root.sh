#!/usr/bin/env bash
# round-4 call 9: the sparse engine with the root front on the dense tile kernel + two-sweep root solves: parity tests of the sparse
# path (incl. 64^3 / 100^3, singular / Schur-complement cases, the soak case), bench lines at 64^3 and 46^3
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
O=gpurun_out/r4c09; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_sparse_big.py tests/test_gpu_round2.py tests/test_gpu_stress.py -q -p no:cacheprovider -k "sparse or Sparse" ) > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log
( timeout 300 python -m pytest tests/test_gpu_fullsize.py -q -p no:cacheprovider -k config4 ) > $O/tests2.log 2>&1
echo "fullsize rc=$?"; tail -2 $O/tests2.log
for g in 64 46; do
  ( timeout 300 python bench.py --workload sparse --grid $g --steps 10 --warmup 2 --no-cpu-baseline ) > $O/bench_sparse$g.json 2> $O/bench_sparse$g.err
done
python - <<'PY'
import json
for g in (64, 46):
    try:
        d = json.load(open("gpurun_out/r4c09/bench_sparse%d.json" % g))
        print(g, d["ms_per_step"], d.get("phases_ms"), (d.get("roofline") or {}))
    except Exception as e:
        print(g, "parse error", e)
PY

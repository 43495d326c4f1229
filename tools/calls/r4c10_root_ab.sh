#!/usr/bin/env bash
# round-4 call 10: same-box A/B of the dense-root path of the sparse engine (knob MI355KKT_SPARSE_NO_DENSE_ROOT through the API)
export PYTHONPATH=.
O=gpurun_out/r4c10; mkdir -p $O
for rep in 1 2; do
for knob in 1 0; do
python - <<PY 2> /dev/null | tail -1 > $O/line_${knob}_${rep}.json
import sys, runpy
from cvxopt_amd import _capi
if $knob: _capi.set_knob('MI355KKT_SPARSE_NO_DENSE_ROOT', '1')
sys.argv = ['bench.py', '--workload', 'sparse', '--grid', '64', '--steps', '10', '--warmup', '2', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')
PY
python - <<PY
import json
d = json.load(open("$O/line_${knob}_${rep}.json"))
print("no_dense_root=$knob rep=$rep", d["ms_per_step"], d["phases_ms"]["factor_ms"], d["phases_ms"]["solve_ms"])
PY
done
done

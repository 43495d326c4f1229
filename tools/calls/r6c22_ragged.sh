#!/usr/bin/env bash
# round-6 call 22: the 512-row solves for orders that are not multiples of 128 (the sparse engine's dense root; dense engine through
# test knob value 2): new tests, sparse suites, sparse bench lines
export PYTHONPATH=.
O=gpurun_out/r6c22; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_round6.py tests/test_gpu_sparse.py tests/test_gpu_sparse_big.py tests/test_gpu_stress.py -m gpu -q -x > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
( timeout 600 python bench.py --workload sparse --no-cpu-baseline ) > $O/bench_sparse.json 2> $O/bench_sparse.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6c22/bench_sparse.json").read().strip().splitlines()[-1])
print("sparse", d["ms_per_step"], d["phases_ms"], d["roofline"].get("solve"))
PY
( timeout 600 python bench.py --workload sparse --mesh elasticity --no-cpu-baseline ) > $O/bench_elast.json 2> $O/bench_elast.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6c22/bench_elast.json").read().strip().splitlines()[-1])
print("elasticity", d["ms_per_step"], d["phases_ms"], d["roofline"].get("solve"))
PY

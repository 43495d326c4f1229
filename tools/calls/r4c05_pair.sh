#!/usr/bin/env bash
# round-4 call 5: the two-sweep triangular solves (trsv_pair_kernel, default on in this build): A/B timing + residuals, the soak
# test, the GPU tests that exercise solves (dense parity, full size, stress), SOCP bench line
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
O=gpurun_out/r4c05; mkdir -p $O
( timeout 300 python tools/dev/trsv_pair_dev.py ) > $O/pair.log 2>&1
grep -v amdgpu.ids $O/pair.log
( MI355KKT_STRESS_ITERS=600 timeout 600 python -m pytest tests/test_gpu_stress.py -q -p no:cacheprovider ) > $O/stress.log 2>&1
echo "stress rc=$?"; tail -3 $O/stress.log
( timeout 900 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_fullsize.py tests/test_gpu_resident.py tests/test_gpu_round3.py tests/test_gpu_solvers.py -q -p no:cacheprovider ) > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log
( timeout 300 python bench.py --workload socp --steps 20 --warmup 3 --no-cpu-baseline ) > $O/bench_socp.json 2> $O/bench_socp.err
( timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads ) > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
for f in ("bench_socp", "bench"):
    try:
        d = json.load(open("gpurun_out/r4c05/%s.json" % f))
        print(f, d["ms_per_step"], d.get("phases_ms"))
    except Exception as e:
        print(f, "parse error", e)
PY

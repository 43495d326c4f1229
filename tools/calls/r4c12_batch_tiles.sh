#!/usr/bin/env bash
# round-4 call 12: the batch's Cholesky as one persistent launch of the variable-batched tile kernel against the launch chain it
# replaces (knob MI355KKT_BATCH_TILES=0): parity tests of the batched engine, A/B timing of one GPU's share of config 5, and the
# headline / SOCP lines with the final SYRK (stream-K remainder round, no synchronisation)
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
O=gpurun_out/r4c12; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_batch.py -q -x -p no:cacheprovider ) > $O/tests.log 2>&1
echo "tests rc=$?" > $O/summary.txt; tail -4 $O/tests.log >> $O/summary.txt
( timeout 300 python tools/dev/bench_batch_dev.py 512 ) > $O/batch_tiles.log 2>&1
( timeout 300 python tools/dev/bench_batch_dev.py 512 MI355KKT_BATCH_TILES=0 ) > $O/batch_chain.log 2>&1
( timeout 300 python tools/dev/bench_batch_dev.py 512 ) > $O/batch_tiles2.log 2>&1
for f in batch_tiles batch_chain batch_tiles2; do echo "== $f" >> $O/summary.txt; grep -E "factor|resident|coneqp_batch" $O/$f.log >> $O/summary.txt; done
( timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads ) > $O/bench.json 2> $O/bench.err
( timeout 300 python bench.py --workload socp --steps 20 --warmup 3 --no-cpu-baseline ) > $O/bench_socp.json 2> $O/bench_socp.err
( timeout 300 python bench.py --workload batch --steps 2 --warmup 1 --no-cpu-baseline ) > $O/bench_batch.json 2> $O/bench_batch.err
python - <<'PY' >> gpurun_out/r4c12/summary.txt
import json
for f in ("bench", "bench_socp", "bench_batch"):
    try:
        d = json.load(open("gpurun_out/r4c12/%s.json" % f))
        print(f, d["ms_per_step"], d.get("value"), d.get("phases_ms"), (d.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(f, "parse error", e)
PY
cat $O/summary.txt

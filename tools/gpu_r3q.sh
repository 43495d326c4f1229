#!/usr/bin/env bash
# validation of the build with the faster batched cone scaling: full GPU suite, smoke, headline bench line, batched timings
export PYTHONPATH=.
O=gpurun_out
( timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) > $O/r3q_tests.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/r3q_smoke.log 2>&1
timeout 200 python tools/dev/bench_batch_q_dev.py > $O/r3q_batch_q.log 2>&1
timeout 300 python bench.py --no-cpu-baseline > $O/r3q_bench.json 2> $O/r3q_bench.err
echo done

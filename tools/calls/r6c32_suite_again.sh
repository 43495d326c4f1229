#!/usr/bin/env bash
# round-6 call 32: the driver's own commands once more on a fresh box (flakiness check of the final build)
export PYTHONPATH=.
O=gpurun_out/r6c32; mkdir -p $O
( timeout 2400 python -m pytest tests/ -x -q -m gpu ) > $O/pytest.txt 2>&1
echo "rc=$?"; tail -3 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -1
( timeout 900 python bench.py ) > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; cut -c1-400 $O/bench.json

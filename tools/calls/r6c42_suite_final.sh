#!/usr/bin/env bash
# round-6 call 42: the driver's test command on the final tree, once more on a fresh box
export PYTHONPATH=.
O=gpurun_out/r6c42; mkdir -p $O
( timeout 2400 python -m pytest tests/ -x -q -m gpu ) > $O/pytest.txt 2>&1
echo "rc=$?"; tail -3 $O/pytest.txt | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -1

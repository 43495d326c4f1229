#!/usr/bin/env bash
# round-6 call 14: the whole GPU suite in one process with the wide solves as the default + smoke + the four bench lines
export PYTHONPATH=.
O=gpurun_out/r6c14; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
( timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err
cut -c1-700 $O/bench.json

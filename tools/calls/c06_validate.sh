#!/usr/bin/env bash
# round-3 call 6: full GPU suite with the parity report, smoke, the default bench line (side workloads + CPU baselines)
export PYTHONPATH=.
O=gpurun_out/c06; mkdir -p $O
export MI355KKT_PARITY_REPORT=$PWD/$O/parity_report.json
( timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > $O/tests.log 2>&1
unset MI355KKT_PARITY_REPORT
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/smoke.log 2>&1
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
echo done

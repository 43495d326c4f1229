#!/usr/bin/env bash
# full validation of the current build: GPU suite, smoke, headline bench, socp + sparse lines
export PYTHONPATH=.
O=gpurun_out
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 ) > $O/r2p_tests.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/r2p_smoke.log 2>&1
timeout 400 python bench.py > $O/r2p_bench.json 2> $O/r2p_bench.err
timeout 300 python bench.py --workload socp --no-cpu-baseline > $O/r2p_socp.json 2> $O/r2p_socp.err
timeout 300 python bench.py --workload sparse --no-cpu-baseline --steps 10 > $O/r2p_sparse.json 2> $O/r2p_sparse.err
echo done

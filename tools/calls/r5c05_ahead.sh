#!/usr/bin/env bash
# round-5 call 5: the strip of the next step requested AFTER the poll (trsv_pair_kernel<.., AHEAD>) against the round-4 order, same
# factor; then the whole GPU suite in one process with the test defaults (0xff-poisoned device blocks, book examples in process),
# no -x: every test that depended on zero-filled blocks shows up at once
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
O=gpurun_out/r5c05; mkdir -p $O
( timeout 300 python tools/dev/trsv_ahead_dev.py ) > $O/trsv_ahead.log 2>&1
grep -v amdgpu.ids $O/trsv_ahead.log > $O/summary.txt
( timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/suite.log 2>&1
echo "suite rc=$? $(tail -1 $O/suite.log | cut -c1-200)" >> $O/summary.txt
grep -h "^FAILED\|^ERROR" $O/suite.log | head -40 >> $O/summary.txt
cat $O/summary.txt

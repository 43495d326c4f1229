#!/usr/bin/env bash
# round-4 call 8: the book examples against the host fixture (every number compared), rocprofv3 kernel statistics + timeline of the
# sparse engine at 64^3
export PYTHONPATH=.
R=$PWD
O=gpurun_out/r4c08; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_reference_examples.py -q -s -p no:cacheprovider ) > $O/book.log 2>&1
echo "book rc=$?"; grep -E "variables|passed|failed|Error" $O/book.log | tail -30
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sparse -o r04s -- python $R/bench.py --workload sparse --steps 4 --warmup 2 --no-cpu-baseline > $R/$O/prof_sparse.log 2>&1
cd $R
DB=$(find /tmp/prof_sparse -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r04_sparse64_kernel_stats.md > /dev/null 2>&1
python tools/sparse_timeline.py $DB > $O/r04_sparse64_timeline.txt 2>&1
head -40 $O/r04_sparse64_kernel_stats.md

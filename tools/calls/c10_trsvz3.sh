#!/usr/bin/env bash
# round-3 call 10: trsv_z with two workgroups per block row + matrix-core preparation of the pre-multiplied blocks
export PYTHONPATH=.
O=gpurun_out/c10; mkdir -p $O
( MI355KKT_TRSV=z timeout 600 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5 ) > $O/tests_z.log 2>&1
( MI355KKT_TRSV=z MI355KKT_TRSVZ_PREP=trsm timeout 600 python -m pytest tests/test_gpu_kkt.py -x -q -m gpu -k "stress or chol2 or socp" 2>&1 | tail -3 ) > $O/tests_z_trsm.log 2>&1
timeout 300 python tools/dev/prof_trsvz_dev.py > $O/trsvz.log 2>&1
for v in z inv; do
MI355KKT_TRSV=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads > $O/bench_$v.json 2> $O/bench_$v.err
MI355KKT_TRSV=$v timeout 600 python bench.py --workload socp --steps 10 --warmup 3 --no-cpu-baseline > $O/socp_$v.json 2> $O/socp_$v.err
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
MI355KKT_TRSV=z timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o z -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-side-workloads > $O/prof.log 2>&1
DB=$(find $O/prof -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/kernel_stats.md > /dev/null 2>&1
rm -rf $O/prof
echo done

#!/usr/bin/env bash
# round-3 call 3: triangular solves with the pre-multiplied sub-diagonal blocks (trsv_z_kernel): parity tests, solve phase, kernel times
export PYTHONPATH=.
O=gpurun_out/c03; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_ops.py tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests_z.log 2>&1
for v in z inv; do
MI355KKT_TRSV=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads > $O/bench_$v.json 2> $O/bench_$v.err
MI355KKT_TRSV=$v timeout 600 python bench.py --workload socp --steps 10 --warmup 3 --no-cpu-baseline > $O/socp_$v.json 2> $O/socp_$v.err
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof -o z -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-side-workloads > $O/prof.log 2>&1
DB=$(find $O/prof -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/kernel_stats.md > /dev/null 2>&1
python tools/potrf_timeline.py $DB 3 400 > $O/timeline.txt 2>&1
rm -rf $O/prof
( timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_resident.py tests/test_gpu_solvers.py -x -q -m gpu 2>&1 | tail -8 ) > $O/tests_full.log 2>&1
echo done

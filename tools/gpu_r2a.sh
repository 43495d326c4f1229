#!/usr/bin/env bash
# round-2 GPU call A: new look-ahead potf2 + granule trsv: correctness, A/B timings, step timeline
export PYTHONPATH=.
O=gpurun_out
mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -15 ) > $O/r2a_ops.log 2>&1
{
for v in "old 2" "new 2" "new 3"; do set -- $v
  MI355KKT_POTF2=$1 MI355KKT_POTRF_STREAMS=$2 timeout 300 python tools/dev/bench_potrf_dev.py 8192
done
for n in 4096 2048 1024; do
  MI355KKT_POTF2=old timeout 300 python tools/dev/bench_potrf_dev.py $n
  MI355KKT_POTF2=new timeout 300 python tools/dev/bench_potrf_dev.py $n
done
} > $O/r2a_potrf.log 2>&1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r2a_bench_new.json 2> $O/r2a_bench_new.err
MI355KKT_TRSV=flag MI355KKT_POTF2=old timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r2a_bench_old.json 2> $O/r2a_bench_old.err
MI355KKT_POTRF_STREAMS=3 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r2a_bench_s3.json 2> $O/r2a_bench_s3.err
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_r2a -o r2a -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/r2a_prof.log 2>&1
DB=$(find $O/prof_r2a -name '*results.db' | head -1)
python tools/potrf_timeline.py $DB 3 400 > $O/r2a_timeline.txt 2>&1
python tools/rocpd_summary.py stats $DB $O/r2a_kernel_stats.md > /dev/null 2>&1
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > $O/r2a_gpu_tests.log 2>&1
echo done

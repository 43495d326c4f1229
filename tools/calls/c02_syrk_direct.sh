#!/usr/bin/env bash
# round-3 call 2: the barrier-free register-staged SYRK (syrk_tn_direct_kernel) against the LDS-staged one: op tests, timings, MFMA duty
export PYTHONPATH=.
O=gpurun_out/c02; mkdir -p $O
for v in direct412 direct411 direct414; do
  ( MI355KKT_SYRK=$v timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k syrk 2>&1 | tail -5 ) > $O/ops_$v.log 2>&1
done
( timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -8 ) > $O/ops_default.log 2>&1
{
for rep in 1 2; do
for v in lds direct411 direct412 direct414; do
  echo "== $v"; MI355KKT_SYRK=$v timeout 300 python tools/dev/syrk_prof_dev.py 6
done; done
} > $O/times.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in direct412 direct414; do
MI355KKT_SYRK=$v timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/p_$v -o a -- python tools/dev/syrk_prof_dev.py 4 > $O/p_$v.log 2>&1
DB=$(find $O/p_$v -name '*results.db' | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py pmcany $DB syrk_tn >> $O/pmc_summary.jsonl 2>&1
rm -rf $O/p_$v
done
MI355KKT_SYRK=direct412 timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $O/pf -o a -- python tools/dev/syrk_prof_dev.py 4 > $O/pf.log 2>&1
DB=$(find $O/pf -name '*results.db' | head -1); [ -n "$DB" ] && python tools/rocpd_summary.py pmcany $DB syrk_tn >> $O/pmc_summary.jsonl 2>&1
rm -rf $O/pf
for v in lds direct412; do
MI355KKT_SYRK=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
done
echo done

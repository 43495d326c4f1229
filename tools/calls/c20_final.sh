#!/usr/bin/env bash
# round-3 call 20: validation of the round's FINAL build (after the batched-engine changes of calls 18-19) -- full GPU suite with the parity report, smoke, rocprofv3 kernel statistics +
# step timeline of the headline command, PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy, each on its own), then the default
# bench line (side workloads + CPU baselines; roofline.traffic from the PMC file written just before)
export PYTHONPATH=.
R=$PWD
O=gpurun_out/c20; mkdir -p $O
export MI355KKT_PARITY_REPORT=$R/$O/parity_report.json
( timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) > $O/tests.log 2>&1
unset MI355KKT_PARITY_REPORT
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 ) > $O/smoke.log 2>&1
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-side-workloads"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_dense -o r03 -- $B > $R/$O/prof_dense.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_fetch -o f -- $B > $R/$O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_write -o w -- $B > $R/$O/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_mfma -o m -- $B > $R/$O/pmc_mfma.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_batch -o r03b -- python $R/bench.py --workload batch --steps 1 --warmup 1 --no-cpu-baseline > $R/$O/prof_batch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_socp -o r03q -- python $R/bench.py --workload socp --steps 4 --warmup 2 --no-cpu-baseline > $R/$O/prof_socp.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_sparse -o r03s -- python $R/bench.py --workload sparse --steps 4 --warmup 2 --no-cpu-baseline > $R/$O/prof_sparse.log 2>&1
cd $R
DB=$(find /tmp/prof_dense -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r03_kernel_stats.md > /dev/null 2>&1
python tools/potrf_timeline.py $DB 3 70 > $O/r03_step_timeline.txt 2>&1
FD=$(find /tmp/pmc_fetch -name '*results.db' | head -1); WD=$(find /tmp/pmc_write -name '*results.db' | head -1); MD=$(find /tmp/pmc_mfma -name '*results.db' | head -1)
python tools/rocpd_summary.py pmc $FD $WD syrk_tn_kernel $O/r03_pmc_syrk.json 8192 16384 > $O/pmc_syrk.log 2>&1
python tools/rocpd_summary.py pmc $FD $WD potrf_tiles_kernel $O/r03_pmc_potrf_tiles.json 8192 16384 > $O/pmc_potrf.log 2>&1
python tools/rocpd_summary.py pmcany $MD syrk_tn_kernel > $O/r03_pmc_mfma.jsonl 2>&1
python tools/rocpd_summary.py pmcany $MD potrf_tiles_kernel >> $O/r03_pmc_mfma.jsonl 2>&1
cp $O/r03_pmc_syrk.json profiles/r03_pmc_syrk.json
DB=$(find /tmp/prof_batch -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r03_batch_kernel_stats.md > /dev/null 2>&1
DB=$(find /tmp/prof_socp -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r03_socp8_kernel_stats.md > /dev/null 2>&1
DB=$(find /tmp/prof_sparse -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r03_sparse46_kernel_stats.md > /dev/null 2>&1
python tools/sparse_timeline.py $DB > $O/r03_sparse46_timeline.txt 2>&1
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
echo done

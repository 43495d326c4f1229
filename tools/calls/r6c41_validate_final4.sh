#!/usr/bin/env bash
# round-6 validation of the build at this commit, the way the driver runs it: the WHOLE GPU suite in ONE process (book examples in
# process, 0xff-poisoned device allocations: the defaults of tests/), parity report, smoke; rocprofv3 kernel statistics + timelines of
# the headline / SOCP / sparse commands; PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy, each on its own); then the driver's
# command: python bench.py (headline + side workloads + CPU baselines; roofline.traffic from the PMC file)
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
R=$PWD
O=gpurun_out/r6c41; mkdir -p $O
export MI355KKT_PARITY_REPORT=$R/$O/r06_parity_report.json
( timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider ) > $O/r06_final_gpu_tests.log 2>&1
echo "suite rc=$? last_test=$(cat gpurun_out/pytest_last_test.txt 2>/dev/null)" > $O/summary.txt; tail -3 $O/r06_final_gpu_tests.log | cut -c1-300 >> $O/summary.txt
grep -h "^FAILED\|^ERROR" $O/r06_final_gpu_tests.log | head -20 >> $O/summary.txt
unset MI355KKT_PARITY_REPORT
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1
tail -2 $O/smoke.log >> $O/summary.txt
B="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-side-workloads"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_dense -o r06 -- $B > $R/$O/prof_dense.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_socp -o q -- python $R/bench.py --workload socp --steps 6 --warmup 2 --no-cpu-baseline > $R/$O/prof_socp.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sparse -o s -- python $R/bench.py --workload sparse --steps 4 --warmup 2 --no-cpu-baseline > $R/$O/prof_sparse.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_fetch -o f -- $B > $R/$O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_write -o w -- $B > $R/$O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pmc_mfma -o m -- $B > $R/$O/pmc_mfma.log 2>&1
cd $R
DB=$(find /tmp/prof_dense -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r06_kernel_stats.md > /dev/null 2>&1
python tools/potrf_timeline.py $DB 3 70 > $O/r06_step_timeline.txt 2>&1
DB=$(find /tmp/prof_socp -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r06_socp8_kernel_stats.md > /dev/null 2>&1
python tools/potrf_timeline.py $DB 4 140 > $O/r06_socp8_step_timeline.txt 2>&1
DB=$(find /tmp/prof_sparse -name '*results.db' | head -1)
python tools/rocpd_summary.py stats $DB $O/r06_sparse64_kernel_stats.md > /dev/null 2>&1
python tools/sparse_timeline.py $DB > $O/r06_sparse64_timeline.txt 2>&1
FD=$(find /tmp/pmc_fetch -name '*results.db' | head -1); WD=$(find /tmp/pmc_write -name '*results.db' | head -1); MD=$(find /tmp/pmc_mfma -name '*results.db' | head -1)
python tools/rocpd_summary.py pmc $FD $WD syrk_tn_kernel $O/r06_pmc_syrk.json 8192 16384 > $O/pmc_syrk.log 2>&1
python tools/rocpd_summary.py pmc $FD $WD potrf_tiles_kernel $O/r06_pmc_potrf_tiles.json 8192 16384 > $O/pmc_potrf.log 2>&1
python tools/rocpd_summary.py pmcany $MD syrk_tn_kernel > $O/r06_pmc_mfma.jsonl 2>&1
python tools/rocpd_summary.py pmcany $MD potrf_tiles_kernel >> $O/r06_pmc_mfma.jsonl 2>&1
cp $O/r06_pmc_syrk.json profiles/pmc_syrk_latest.json; cp $O/r06_pmc_syrk.json profiles/r06_pmc_syrk.json
( timeout 1500 python bench.py ) > $O/r06_final_bench.json 2> $O/bench.err
echo "bench rc=$?" >> $O/summary.txt
python - <<'PY' >> gpurun_out/r6c41/summary.txt
import json
try:
    d = json.load(open("gpurun_out/r6c41/r06_final_bench.json"))
    print("headline", d["ms_per_step"], d["phases_ms"], "hook", d.get("hook_ms_per_step"), "roofline", d["roofline"], "cpu", d.get("cpu_baseline", {}).get("value"))
    for k, v in d.get("side_workloads", {}).items():
        print(k, v.get("ms_per_step"), v.get("value"), (v.get("roofline") or {}).get("frac"), (v.get("cpu_baseline") or {}).get("value"), v.get("error"))
except Exception as e:
    print("bench parse error", e)
PY
head -14 $O/r06_kernel_stats.md >> $O/summary.txt
cat $O/summary.txt

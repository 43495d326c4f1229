#!/usr/bin/env bash
# round-4 call 11: the scaled SYRK with the stream-K remainder round and the XCD-local k synchronisation:
# correctness against NumPy / the free-running kernel, a sweep of (announce interval, lag), PMC traffic + L2 hit rate + MFMA busy /
# clock of the free-running and the synchronised kernel (each counter set in its own pass), the SOCP bench line, the KKT parity tests
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
R=$PWD
O=gpurun_out/r4c11; mkdir -p $O
( timeout 420 python tools/dev/syrk_sync_dev.py 7 ) > $O/sync_dev.log 2>&1
echo "sync_dev rc=$?" > $O/summary.txt; cat $O/sync_dev.log >> $O/summary.txt
cd /tmp && export TMPDIR=/tmp
for mode in off on; do
  K=""; [ "$mode" = off ] && K="off"
  P="python $R/tools/dev/syrk_prof_dev.py 6 $K"
  timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$mode -o f -- $P > $R/$O/pf_$mode.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$mode -o w -- $P > $R/$O/pw_$mode.log 2>&1
  timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/pt_$mode -o t -- $P > $R/$O/pt_$mode.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/pm_$mode -o m -- $P > $R/$O/pm_$mode.log 2>&1
done
cd $R
for mode in off on; do
  FD=$(find /tmp/pf_$mode -name '*results.db' | head -1); WD=$(find /tmp/pw_$mode -name '*results.db' | head -1)
  TD=$(find /tmp/pt_$mode -name '*results.db' | head -1); MD=$(find /tmp/pm_$mode -name '*results.db' | head -1)
  python tools/rocpd_summary.py pmc $FD $WD syrk_tn_kernel $O/r04_pmc_syrk_$mode.json 8192 16384 > $O/pmc_$mode.log 2>&1
  python tools/rocpd_summary.py pmcany $TD syrk_tn_kernel > $O/r04_pmc_syrk_l2_$mode.json 2>&1
  python tools/rocpd_summary.py pmcany $MD syrk_tn_kernel > $O/r04_pmc_syrk_mfma_$mode.json 2>&1
  echo "== $mode" >> $O/summary.txt; grep -h "min " $O/pf_$mode.log $O/pm_$mode.log >> $O/summary.txt
  python - $O $mode >> $O/summary.txt <<'PY'
import json, sys
O, mode = sys.argv[1:3]
try:
    d = json.load(open("%s/r04_pmc_syrk_%s.json" % (O, mode)))
    print(mode, "hbm GB per launch %.2f" % (d["hbm_bytes_per_launch"] / 1e9), "fetch KiB", d["FETCH_SIZE_KiB_avg"], "write KiB", d["WRITE_SIZE_KiB_avg"])
    for f in ("l2", "mfma"):
        print(open("%s/r04_pmc_syrk_%s_%s.json" % (O, f, mode)).read().strip()[:400])
except Exception as e:
    print(mode, "pmc parse error", e)
PY
done
( timeout 300 python bench.py --workload socp --steps 20 --warmup 3 --no-cpu-baseline ) > $O/bench_socp.json 2> $O/bench_socp.err
python - <<'PY' >> gpurun_out/r4c11/summary.txt
import json
try:
    d = json.load(open("gpurun_out/r4c11/bench_socp.json"))
    print("socp", d["ms_per_step"], d.get("phases_ms"), (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print("socp parse error", e)
PY
( timeout 500 python -m pytest tests/test_gpu_kkt.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -x -p no:cacheprovider ) > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/summary.txt; tail -4 $O/tests.log >> $O/summary.txt
( timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads ) > $O/bench.json 2> $O/bench.err
python - <<'PY' >> gpurun_out/r4c11/summary.txt
import json
try:
    d = json.load(open("gpurun_out/r4c11/bench.json"))
    print("headline", d["ms_per_step"], d.get("phases_ms"), (d.get("roofline") or {}).get("frac"))
except Exception as e:
    print("headline parse error", e)
PY
cat $O/summary.txt

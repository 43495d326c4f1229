#!/usr/bin/env bash
# round-3 call 17: leading dimensions that are powers of two -- cache-set conflicts in the SYRK / tile Cholesky? (padded copies, timings + L2 hit counters)
export PYTHONPATH=.
O=gpurun_out/c17; mkdir -p $O
timeout 600 python tools/dev/ld_pad_dev.py all 5 > $O/ld.log 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/p -o d -- python tools/dev/ld_pad_dev.py syrk 2 > $O/p.log 2>&1
DB=$(find $O/p -name '*results.db' | head -1)
python - $DB > $O/tcc.log 2>&1 <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
namecol = "kernel_name" if "kernel_name" in cols else "name"
per = {}
for kname, cname, val, did in cur.execute("select %s, counter_name, value, dispatch_id from counters_collection" % namecol):
    if "syrk_tn_kernel" in kname:
        per.setdefault(did, {}).setdefault(cname, 0.0)
        per[did][cname] += float(val)
for did in sorted(per):
    print(did, per[did])
PY
rm -rf $O/p
echo done

#!/usr/bin/env python
"""Prints the kernel timeline of one dense potrf from a rocprofv3 rocpd database (developer aid).
    python tools/potrf_timeline.py <results.db> [which syrk occurrence] [max lines]"""
import sqlite3, sys
db = sys.argv[1]
occ = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 80
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end, stream_id, grid_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if 'syrk_tn' in r[0]]
i0 = idx[occ]
i1 = idx[occ + 1] if occ + 1 < len(idx) else len(rows)
seq = rows[i0:i1]
t0 = seq[0][1]
tot = {}
for r in seq:
    nm = r[0].split('(')[0].replace('mi355kkt::', '').replace('void ', '')[:30]
    tot.setdefault((nm, r[3]), [0, 0.0])
    tot[(nm, r[3])][0] += 1
    tot[(nm, r[3])][1] += (r[2] - r[1]) / 1e3
for r in seq[:nmax]:
    nm = r[0].split('(')[0].replace('mi355kkt::', '').replace('void ', '')[:30]
    print("%-30s start %9.1f  dur %7.1f  end %9.1f  stream %s grid %d" % (nm, (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, (r[2] - t0) / 1e3, r[3], r[4]))
print("---- totals over the step (us):")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-30s stream %s  calls %4d  total %9.1f  avg %7.1f" % (k[0], k[1], v[0], v[1], v[1] / v[0]))
print("step span %.1f us" % ((seq[-1][2] - t0) / 1e3))

#!/usr/bin/env python
"""Prints the kernel timeline of the LAST sparse solve (forward + backward substitution) and of the last numeric factorisation
found in a rocprofv3 rocpd database (developer aid).   python tools/sparse_timeline.py <results.db>"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end, grid_x, grid_y from kernels order by start").fetchall()
short = lambda n: n.split('(')[0].replace('mi355kkt::', '').replace('void ', '')[:34]
# last solve: from the last sp_permute (forward start) pair
perm = [i for i, r in enumerate(rows) if 'sp_permute_kernel' in r[0]]
i1 = perm[-1]; i0 = perm[-2]
seq = rows[i0:i1 + 1]
t0 = seq[0][1]
print("== last solve: %d kernels, span %.1f us" % (len(seq), (seq[-1][2] - t0) / 1e3))
for r in seq:
    print("%-34s start %8.1f dur %7.1f grid %5d x %3d" % (short(r[0]), (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4]))
# last factor: from the last sp_assemble to the kernel before the following sp_permute
asm = [i for i, r in enumerate(rows) if 'sp_assemble_kernel' in r[0]]
a0 = asm[-1]
a1 = min([i for i in perm if i > a0] or [len(rows)])
seq = rows[a0:a1]
t0 = seq[0][1]
print("== last factor: %d kernels, span %.1f us" % (len(seq), (seq[-1][2] - t0) / 1e3))
tot = {}
for r in seq:
    tot.setdefault(short(r[0]), [0, 0.0]); tot[short(r[0])][0] += 1; tot[short(r[0])][1] += (r[2] - r[1]) / 1e3
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print("%-34s calls %4d total %8.1f us" % (k, v[0], v[1]))
busy = sum((r[2] - r[1]) for r in seq) / 1e3
print("busy %.1f us of span %.1f us" % (busy, (seq[-1][2] - t0) / 1e3))
for r in seq:
    print("%-34s start %8.1f dur %7.1f grid %5d x %3d" % (short(r[0]), (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], r[4]))

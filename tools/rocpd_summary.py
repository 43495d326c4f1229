#!/usr/bin/env python
"""Turns rocprofv3's rocpd sqlite output into the text/JSON summaries committed under profiles/.

    python tools/rocpd_summary.py stats  <results.db> <out.md>           # == `--kernel-trace --stats` table
    python tools/rocpd_summary.py pmc    <fetch.db> <write.db> <kernel-substring> <out.json> n m

PMC correction (guide MI355X_MICROARCH.md, section HBM): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read -> doubled here.
"""
import json
import sqlite3
import sys


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, s, a, mn, mx in rows:
        lines.append("| `%s` | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (name[:90], n, s / 1e6, a / 1e3, mn / 1e3,
                                                                          mx / 1e3, 100.0 * s / tot))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))


def pmc_per_kernel(db, counter, substr):
    cur = sqlite3.connect(db).cursor()
    cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    q = "select %s, counter_name, value, dispatch_id from counters_collection" % namecol
    per = {}
    for kname, cname, val, did in cur.execute(q):
        if cname == counter and substr in kname:
            per[did] = per.get(did, 0.0) + float(val)
    vals = list(per.values())
    return vals


def source_id():
    """sha256 over the sources that define the SYRK kernel -- csrc/gemm_f64.hip and the tile-geometry / SYRK block of
    csrc/kkt_common.h (from "// ---- tile geometry" to "// ---- dense Cholesky"; the rest of that header belongs to other
    engines) -- bench.py recomputes it and refuses a PMC file of another version"""
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hsh = hashlib.sha256()
    hsh.update(open(os.path.join(root, "cvxopt_amd/csrc/gemm_f64.hip"), "rb").read())
    txt = open(os.path.join(root, "cvxopt_amd/csrc/kkt_common.h")).read()
    hsh.update(txt[txt.index("// ---- tile geometry of the FP64 MFMA kernels"):txt.index("// ---- dense Cholesky")].encode())
    return hsh.hexdigest()[:16]


def pmc(fetch_db, write_db, substr, out, n, m):
    f = pmc_per_kernel(fetch_db, "FETCH_SIZE", substr)
    w = pmc_per_kernel(write_db, "WRITE_SIZE", substr)
    favg = sum(f) / max(1, len(f))
    wavg = sum(w) / max(1, len(w))
    rec = {"kernel": substr, "n": int(n), "m": int(m), "launches_sampled": [len(f), len(w)],
           "FETCH_SIZE_KiB_avg": favg, "WRITE_SIZE_KiB_avg": wavg,
           "correction": "FETCH_SIZE x2 on gfx950 (guide: counts 128-B requests as 64 B); WRITE_SIZE uncalibrated, x1",
           "hbm_bytes_per_launch": 2.0 * favg * 1024.0 + wavg * 1024.0,
           "kernel_source_id": source_id(),
           "kernel_source_files": ["cvxopt_amd/csrc/gemm_f64.hip", "cvxopt_amd/csrc/kkt_common.h (tile geometry / SYRK block)"]}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


def pmc_any(db, substr):
    """every counter of a --pmc pass, summed over the dimensions of a dispatch, averaged over the dispatches of the kernel"""
    cur = sqlite3.connect(db).cursor()
    cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    per = {}
    for kname, cname, val, did in cur.execute("select %s, counter_name, value, dispatch_id from counters_collection" % namecol):
        if substr in kname:
            per.setdefault(cname, {}).setdefault(did, 0.0)
            per[cname][did] += float(val)
    out = {c: sum(v.values()) / max(1, len(v)) for c, v in per.items()}
    out["_dispatches"] = max([len(v) for v in per.values()] or [0])
    print(json.dumps({"kernel": substr, "db": db, "avg_per_dispatch": out}))


def pmc_table(fetch_db, write_db, out):
    """bytes moved past the L2 per kernel NAME over a whole run: FETCH_SIZE (x2, KiB) and WRITE_SIZE (KiB) passes side by side"""
    def per_name(db, counter):
        cur = sqlite3.connect(db).cursor()
        cols = [c[1] for c in cur.execute("pragma table_info('counters_collection')")]
        namecol = "kernel_name" if "kernel_name" in cols else "name"
        tot, disp = {}, {}
        for kname, cname, val, did in cur.execute("select %s, counter_name, value, dispatch_id from counters_collection" % namecol):
            if cname == counter:
                tot[kname] = tot.get(kname, 0.0) + float(val)
                disp.setdefault(kname, set()).add(did)
        return tot, {k: len(v) for k, v in disp.items()}
    f, fd = per_name(fetch_db, "FETCH_SIZE")
    w, wd = per_name(write_db, "WRITE_SIZE")
    names = sorted(set(f) | set(w), key=lambda k: -(2.0 * f.get(k, 0.0) + w.get(k, 0.0)))
    lines = ["| kernel | launches | fetch GB (FETCH_SIZE x2) | write GB | GB per launch |", "|---|---|---|---|---|"]
    for k in names:
        n = max(fd.get(k, 0), wd.get(k, 0), 1)
        fb, wb = 2.0 * f.get(k, 0.0) * 1024.0 / 1e9, w.get(k, 0.0) * 1024.0 / 1e9
        lines.append("| `%s` | %d | %.3f | %.3f | %.4f |" % (k[:90], n, fb, wb, (fb + wb) / n))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:16]))


if __name__ == "__main__":
    if sys.argv[1] == "pmctable":
        pmc_table(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "pmcany":
        pmc_any(sys.argv[2], sys.argv[3])
    else:
        pmc(*sys.argv[2:9])

import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: d[k] for k in d.files}


def dims_of(rec):
    return {'l': int(rec['dims_l']), 'q': [int(k) for k in rec['dims_q']], 's': [int(k) for k in rec['dims_s']]}


def w_of(rec, dims):
    W = {'d': rec['W_d'].copy(), 'di': rec['W_di'].copy(), 'v': [], 'beta': [float(b) for b in rec['W_beta']],
         'r': [], 'rti': []}
    o = 0
    for mk in dims['q']:
        W['v'].append(rec['W_v'][o:o + mk].copy())
        o += mk
    o = 0
    for nk in dims['s']:
        W['r'].append(rec['W_r'][o:o + nk * nk].reshape(nk, nk, order='F').copy())
        W['rti'].append(rec['W_rti'][o:o + nk * nk].reshape(nk, nk, order='F').copy())
        o += nk * nk
    return W


def relerr(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b)))) if a.size else 0.0


# ---- parity report: achieved errors of the full-size fixtures, written when $MI355KKT_PARITY_REPORT names a file ------------
_REPORT = {}


def record(test, **values):
    """the errors a parity test actually achieved (the test then asserts them against its bounds): kept per test and dumped as
    JSON after every call, so a GPU run of the suite leaves profiles/rNN_parity_report.json behind"""
    path = os.environ.get("MI355KKT_PARITY_REPORT")
    clean = {}
    for k, v in values.items():
        if isinstance(v, (list, tuple, np.ndarray)):
            clean[k] = [float(x) for x in np.asarray(v, dtype=float).ravel()]
        else:
            clean[k] = float(v) if isinstance(v, (float, np.floating)) else (int(v) if isinstance(v, (int, np.integer)) else v)
    _REPORT.setdefault(test, {}).update(clean)
    if path:
        import json
        old = {}
        if os.path.exists(path):
            try:
                old = json.load(open(path))
            except Exception:
                old = {}
        old.update(_REPORT)
        json.dump(old, open(path, "w"), indent=1, sort_keys=True)
    return clean

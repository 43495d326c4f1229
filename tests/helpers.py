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

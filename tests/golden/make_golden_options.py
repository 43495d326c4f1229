"""Fixtures for the solver OPTIONS as inputs (VERDICT r3 "missing" 1): the real reference (oracle/_ref) run with
options['use_correction'] = False, options['refinement'] = 2, options['maxiters'] = 3, explicit tolerances, and with
initvals = {} (the reference starts from x = 0, y = 0, s = z = e then: coneprog.py:2107-2149) on small seeded problems.
tests/test_gpu_options.py runs the device loops with the same options against these.

    python tests/golden/make_golden_options.py        # rewrites tests/golden/options.npz (deterministic, seeded)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refloader            # noqa: E402
from cvxopt_amd import synth            # noqa: E402

refloader.load()
from cvxopt import matrix, solvers      # noqa: E402


def qp_problem(kind):
    """kind -> (P, q, G, h, dims, A, b) as NumPy arrays"""
    if kind == 'lp_cone':
        pr = synth.dense_qp(40, 90, seed=11)
        return pr['P'], pr['q'], pr['G'], pr['h'], pr['dims'], None, None
    if kind == 'lp_cone_eq':
        pr = synth.dense_qp(36, 70, seed=12, p=5)
        return pr['P'], pr['q'], pr['G'], pr['h'], pr['dims'], pr['A'], pr['b']
    if kind == 'soc':
        pr = synth.socp(24, 6, 5, seed=13, ml=10)
        rng = np.random.default_rng(13)
        B = rng.standard_normal((24, 24)) / 5.0
        return np.asfortranarray(B.T @ B + 0.05 * np.eye(24)), -pr['c'], pr['G'], pr['h'], pr['dims'], None, None
    raise KeyError(kind)


CASES = [   # (name, problem, options, initvals)
    ('default', 'lp_cone', {}, None),
    ('nocorr', 'lp_cone', {'use_correction': False}, None),
    ('nocorr_eq', 'lp_cone_eq', {'use_correction': False}, None),
    ('nocorr_soc', 'soc', {'use_correction': False}, None),
    ('refine2', 'lp_cone', {'refinement': 2}, None),
    ('refine2_soc', 'soc', {'refinement': 2}, None),
    ('refine0_soc', 'soc', {'refinement': 0}, None),
    ('maxit3', 'lp_cone', {'maxiters': 3}, None),
    ('maxit3_soc', 'soc', {'maxiters': 3}, None),
    ('tols', 'lp_cone_eq', {'abstol': 1e-9, 'reltol': 1e-8, 'feastol': 1e-8}, None),
    ('abs_only', 'lp_cone', {'abstol': 1e-5, 'reltol': -1.0}, None),
    ('empty_initvals', 'lp_cone', {}, {}),
    ('empty_initvals_soc', 'soc', {}, {}),
    ('nocorr_refine2_soc', 'soc', {'use_correction': False, 'refinement': 2}, None),
]


def main():
    out = {}
    for name, kind, opts, initvals in CASES:
        P, q, G, h, dims, A, b = qp_problem(kind)
        o = dict(opts)
        o['show_progress'] = False
        kw = {}
        if A is not None:
            kw = {'A': matrix(A), 'b': matrix(b)}
        sol = solvers.coneqp(matrix(P), matrix(q), matrix(G), matrix(h), dims, initvals=initvals, options=o, **kw)
        out[name + '_status'] = np.array(sol['status'])
        out[name + '_iterations'] = np.array(sol['iterations'])
        for k in ('x', 'y', 's', 'z'):
            out[name + '_' + k] = np.array(sol[k]).ravel()
        for k, key in (('pobj', 'primal objective'), ('dobj', 'dual objective'), ('gap', 'gap'), ('pres', 'primal infeasibility'),
                       ('dres', 'dual infeasibility')):
            out[name + '_' + k] = np.array(float(sol[key]))
        print("%-22s %-8s %2d iterations  pobj % .12e" % (name, sol['status'], sol['iterations'], sol['primal objective']))
    # conelp: refinement and maxiters are read there too (coneprog.py:435-437, :502-509)
    pr = synth.socp(20, 5, 4, seed=21, ml=8)
    for name, opts in (('lp_default', {}), ('lp_refine2', {'refinement': 2}), ('lp_refine0', {'refinement': 0}), ('lp_maxit4', {'maxiters': 4})):
        o = dict(opts)
        o['show_progress'] = False
        sol = solvers.conelp(matrix(pr['c']), matrix(pr['G']), matrix(pr['h']), pr['dims'], options=o)
        out[name + '_status'] = np.array(sol['status'])
        out[name + '_iterations'] = np.array(sol['iterations'])
        for k in ('x', 's', 'z'):
            out[name + '_' + k] = np.array(sol[k]).ravel()
        out[name + '_pobj'] = np.array(float(sol['primal objective']))
        out[name + '_dobj'] = np.array(float(sol['dual objective']))
        print("%-22s %-8s %2d iterations  pobj % .12e" % (name, sol['status'], sol['iterations'], sol['primal objective']))
    np.savez_compressed(os.path.join(HERE, "options.npz"), **out)


if __name__ == "__main__":
    main()

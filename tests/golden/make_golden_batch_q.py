"""Fixtures for the batched engine with second-order cones: the REAL reference (oracle/_ref) solves every problem of two small
batches with solvers.coneqp(P, q, G, h, dims[, A, b]) (default kktsolver 'chol' with second-order cones, refinement 1):

  batch_q      8 problems, n = 20, dims = {'l': 6, 'q': [5, 3, 12, 40]} (a cone beyond 32 rows: the wave kernel), p = 0
  batch_q_eq   6 problems, n = 18, dims = {'l': 0, 'q': [4, 4, 9]}, p = 3 equality constraints

Also one random scaling W per problem of batch_q with the solution of the KKT system by the reference's kkt_chol (hook level).

    python tests/golden/make_golden_batch_q.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import refloader            # noqa: E402

refloader.load()
from cvxopt import matrix, solvers, misc      # noqa: E402

solvers.options['show_progress'] = False


def interior(rng, dims):
    parts = [rng.random(dims['l']) + 0.5]
    for k in dims['q']:
        u = rng.standard_normal(k - 1)
        parts.append(np.concatenate([[np.linalg.norm(u) + rng.random() + 0.5], u]))
    return np.concatenate(parts)


def problem(rng, n, dims, p):
    m = dims['l'] + sum(dims['q'])
    Bm = rng.standard_normal((n, n))
    P = Bm @ Bm.T / n + 0.1 * np.eye(n)
    G = rng.standard_normal((m, n))
    A = rng.standard_normal((p, n))
    x0 = rng.standard_normal(n)
    z0 = interior(rng, dims)
    y0 = rng.standard_normal(p)
    q = -(P @ x0 + G.T @ z0 + A.T @ y0)
    h = G @ x0 + interior(rng, dims)
    b = A @ x0
    return P, q, G, h, A, b


def make(name, B, n, dims, p, seed, with_kkt=False):
    rng = np.random.default_rng(seed)
    keys = ['P', 'q', 'G', 'h', 'A', 'b', 'x', 'y', 's', 'z', 'iterations', 'pobj', 'dobj']
    if with_kkt:
        keys += ['Wdi', 'Wv', 'Wbeta', 'bx', 'bz', 'ux', 'uz']
    rec = {k: [] for k in keys}
    cd = {'l': dims['l'], 'q': list(dims['q']), 's': []}
    for _ in range(B):
        P, q, G, h, A, b = problem(rng, n, dims, p)
        args = [matrix(P), matrix(q), matrix(G), matrix(h), cd]
        if p:
            args += [matrix(A), matrix(b)]
        sol = solvers.coneqp(*args)
        assert sol['status'] == 'optimal', sol['status']
        for k, v in (('P', P), ('q', q), ('G', G), ('h', h), ('A', A), ('b', b)):
            rec[k].append(v)
        for k in ('x', 'y', 's', 'z'):
            rec[k].append(np.array(sol[k]).ravel())
        rec['iterations'].append(sol['iterations'])
        rec['pobj'].append(sol['primal objective'])
        rec['dobj'].append(sol['dual objective'])
        if with_kkt:
            s0, z0 = matrix(interior(rng, dims)), matrix(interior(rng, dims))
            lmbda = matrix(0.0, (len(s0), 1))
            W = misc.compute_scaling(s0, z0, lmbda, cd)
            f = misc.kkt_chol(matrix(G), cd, matrix(A))(W, matrix(P))
            bx, bz = rng.standard_normal(n), rng.standard_normal(len(s0))
            ux, uy, uz = matrix(bx), matrix(0.0, (p, 1)), matrix(bz)
            f(ux, uy, uz)
            rec['Wdi'].append(np.array(W['di']).ravel())
            rec['Wv'].append(np.concatenate([np.array(v).ravel() for v in W['v']]))
            rec['Wbeta'].append(np.array(W['beta'], dtype=float))
            rec['bx'].append(bx)
            rec['bz'].append(bz)
            rec['ux'].append(np.array(ux).ravel())
            rec['uz'].append(np.array(uz).ravel())
    out = {k: np.array(v) for k, v in rec.items()}
    out['dims_l'] = np.array(dims['l'])
    out['dims_q'] = np.array(dims['q'])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print("wrote %s: iterations %s" % (name, rec['iterations']))


if __name__ == "__main__":
    make('batch_q', 8, 20, {'l': 6, 'q': [5, 3, 12, 40]}, 0, 0, with_kkt=True)
    make('batch_q_eq', 6, 18, {'l': 0, 'q': [4, 4, 9]}, 3, 1)

"""Fixtures for the batched engine with equality constraints: the REAL reference (oracle/_ref) solves every problem of two
small batches with solvers.coneqp(P, q, G, h, A=A, b=b) (default kktsolver chol2 for the LP cone):

  batch_eq           8 problems, n = 24, m = 40, p = 5, P positive definite
  batch_eq_singular  6 problems, n = 16, m = 6, p = 6, rank(P) = 6: S = P + G'D^2G is singular for every scaling, the reference
                     switches to S + A'A at its first factorisation (misc.py:1433-1447)

    python tests/golden/make_golden_batch_eq.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import refloader            # noqa: E402

refloader.load()
from cvxopt import matrix, solvers      # noqa: E402

solvers.options['show_progress'] = False


def problem(rng, n, m, p, rank):
    Bm = rng.standard_normal((n, rank))
    P = Bm @ Bm.T / rank + (0.1 * np.eye(n) if rank == n else 0.0)
    G = rng.standard_normal((m, n))
    A = rng.standard_normal((p, n))
    x0 = rng.standard_normal(n)
    z0 = rng.random(m) + 0.5
    y0 = rng.standard_normal(p)
    q = -(P @ x0 + G.T @ z0 + A.T @ y0)          # (z0, y0) dual feasible
    h = G @ x0 + rng.random(m) + 0.5              # x0 strictly primal feasible
    b = A @ x0
    return P, q, G, h, A, b


def make(name, B, n, m, p, rank, seed):
    rng = np.random.default_rng(seed)
    rec = {k: [] for k in ('P', 'q', 'G', 'h', 'A', 'b', 'x', 'y', 's', 'z', 'iterations', 'pobj', 'dobj')}
    for _ in range(B):
        P, q, G, h, A, b = problem(rng, n, m, p, rank)
        sol = solvers.coneqp(matrix(P), matrix(q), matrix(G), matrix(h), None, matrix(A), matrix(b))
        assert sol['status'] == 'optimal', sol['status']
        for k, v in (('P', P), ('q', q), ('G', G), ('h', h), ('A', A), ('b', b)):
            rec[k].append(v)
        for k in ('x', 'y', 's', 'z'):
            rec[k].append(np.array(sol[k]).ravel())
        rec['iterations'].append(sol['iterations'])
        rec['pobj'].append(sol['primal objective'])
        rec['dobj'].append(sol['dual objective'])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **{k: np.array(v) for k, v in rec.items()})
    print("wrote %s: iterations %s" % (name, rec['iterations']))


if __name__ == "__main__":
    make('batch_eq', 8, 24, 40, 5, 24, 0)
    make('batch_eq_singular', 6, 16, 6, 6, 6, 1)

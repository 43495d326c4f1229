"""Semidefinite-cone fixtures: the REAL reference (cvxopt built from /root/reference by oracle/build_ref.sh) on cone programs
with 's' blocks, run in the build container and committed with this script (tests/golden/sdp_*.npz):

    python tests/golden/make_golden_sdp.py

  sdp_doc        the SDP of the reference's documentation, examples/doc/chap8/sdp.py (two blocks, 2 and 3)
  sdp_mc20/60/150 the max-cut relaxation of examples/doc/chap8/mcsdp.py as a plain cone LP: min 1'x s.t. w + diag(x) >= 0
                 (the example's own data generator: w = normal(n, n) symmetrised), solved with the default kktsolver
  sdp_mixed      a random feasible cone LP over R^l_+ x two second-order cones x three 's' blocks with equality constraints
  sdp_qp         a cone QP (solvers.coneqp) with an LP block, a second-order cone and two 's' blocks, equality constraints
  sdp_pinf       a primal infeasible cone LP with an 's' block (certificate run)

Every fixture holds the problem data (c/q, P, G, h, dims, A, b), the reference's solution (x, y, s, z, objectives, gap,
status, iteration count) and the per-iteration table it prints with options['show_progress'].
"""
import contextlib
import io
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refloader            # noqa: E402

cvx = refloader.load()
from cvxopt import matrix, solvers      # noqa: E402

LINE = re.compile(r"^\s*(\d+):\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)(?:\s+(\S+))?\s*$")


def logged(fn):
    buf = io.StringIO()
    solvers.options['show_progress'] = True
    try:
        with contextlib.redirect_stdout(buf):
            sol = fn()
    finally:
        solvers.options['show_progress'] = False
    rows = []
    for ln in buf.getvalue().splitlines():
        mm = LINE.match(ln)
        if mm:
            rows.append([float(v) if v is not None else np.nan for v in mm.groups()[1:]])
    return sol, np.array(rows)


def arr(v):
    return np.zeros(0) if v is None else np.array(v, dtype=float).ravel(order='F')


def save(name, data, sol, table):
    rec = dict(data)
    num = lambda v: np.nan if v is None else float(v)
    rec.update({'status': str(sol['status']), 'iterations': int(sol['iterations']), 'x': arr(sol['x']), 'y': arr(sol['y']),
                's': arr(sol['s']), 'z': arr(sol['z']), 'pobj': num(sol['primal objective']), 'dobj': num(sol['dual objective']),
                'gap': num(sol['gap']), 'table': table})
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **rec)
    print("wrote %s: %s in %d iterations, pobj %r" % (name, sol['status'], sol['iterations'], sol['primal objective']))


def dims_arrays(dims):
    return {'dims_l': np.array(dims['l']), 'dims_q': np.array(dims['q'], dtype=int), 'dims_s': np.array(dims['s'], dtype=int)}


def conelp_case(name, c, G, h, dims, A=None, b=None):
    kw = {}
    if A is not None:
        kw = {'A': matrix(A), 'b': matrix(b)}
    sol, table = logged(lambda: solvers.conelp(matrix(c), matrix(G), matrix(h), dims, **kw))
    data = {'c': c, 'G': G, 'h': h, 'A': A if A is not None else np.zeros((0, len(c))), 'b': b if b is not None else np.zeros(0)}
    data.update(dims_arrays(dims))
    save(name, data, sol, table)


def doc_sdp():
    c = np.array([1., -1., 1.])
    G0 = np.array([[-7., -11., -11., 3.], [7., -18., -18., 8.], [-2., -8., -8., 1.]]).T
    G1 = np.array([[-21., -11., 0., -11., 10., 8., 0., 8., 5.], [0., 10., 16., 10., -10., -10., 16., -10., 3.],
                   [-5., 2., -17., 2., -6., 8., -17., -7., 6.]]).T
    h0 = np.array([[33., -9.], [-9., 26.]])
    h1 = np.array([[14., 9., 40.], [9., 91., 10.], [40., 10., 15.]])
    G = np.vstack([G0, G1])
    h = np.concatenate([h0.ravel(order='F'), h1.ravel(order='F')])
    conelp_case('sdp_doc', c, G, h, {'l': 0, 'q': [], 's': [2, 3]})


def maxcut(n, seed):
    rng = np.random.default_rng(seed)
    w = rng.standard_normal((n, n))
    w = 0.5 * (w + w.T)
    G = np.zeros((n * n, n))
    for j in range(n):
        G[j * (n + 1), j] = -1.0
    conelp_case('sdp_mc%d' % n, np.ones(n), G, w.ravel(order='F'), {'l': 0, 'q': [], 's': [n]})


def _rand_sym(rng, m):
    a = rng.standard_normal((m, m))
    return 0.5 * (a + a.T)


def _cone_interior(rng, dims):
    """a strictly interior point of the cone, in the unpacked storage"""
    parts = [rng.random(dims['l']) + 0.5]
    for mk in dims['q']:
        v = rng.standard_normal(mk)
        v[0] = np.linalg.norm(v[1:]) + 0.5 + rng.random()
        parts.append(v)
    for mk in dims['s']:
        a = rng.standard_normal((mk, mk))
        parts.append((a @ a.T / mk + 0.5 * np.eye(mk)).ravel(order='F'))
    return np.concatenate(parts)


def _sym_columns(rng, dims, n):
    """G whose 's' rows are vec's of symmetric matrices (like every SDP in standard form)"""
    cols = []
    for _ in range(n):
        parts = [rng.standard_normal(dims['l'])] + [rng.standard_normal(mk) for mk in dims['q']]
        parts += [_rand_sym(rng, mk).ravel(order='F') for mk in dims['s']]
        cols.append(np.concatenate(parts))
    return np.array(cols).T


def mixed(seed=1):
    rng = np.random.default_rng(seed)
    dims = {'l': 6, 'q': [4, 3], 's': [3, 5, 2]}
    n, p = 9, 2
    G = _sym_columns(rng, dims, n)
    A = rng.standard_normal((p, n))
    x0 = rng.standard_normal(n)
    s0, z0 = _cone_interior(rng, dims), _cone_interior(rng, dims)
    h = G @ x0 + s0                                        # primal strictly feasible
    b = A @ x0
    # weights of the inner product <z, Gx>: the 's' parts count off-diagonal entries twice through both triangles already
    c = -(G.T @ z0) - A.T @ rng.standard_normal(p)         # dual strictly feasible
    conelp_case('sdp_mixed', c, G, h, dims, A, b)


def qp(seed=2):
    rng = np.random.default_rng(seed)
    dims = {'l': 5, 'q': [4], 's': [4, 3]}
    n, p = 8, 2
    G = _sym_columns(rng, dims, n)
    A = rng.standard_normal((p, n))
    x0 = rng.standard_normal(n)
    h = G @ x0 + _cone_interior(rng, dims)
    b = A @ x0
    B = rng.standard_normal((n, n))
    P = B @ B.T / n + 0.1 * np.eye(n)
    q = rng.standard_normal(n)
    sol, table = logged(lambda: solvers.coneqp(matrix(P), matrix(q), matrix(G), matrix(h), dims, matrix(A), matrix(b)))
    data = {'P': P, 'q': q, 'G': G, 'h': h, 'A': A, 'b': b}
    data.update(dims_arrays(dims))
    save('sdp_qp', data, sol, table)


def pinf():
    # x >= 0 together with x I <= -I (3 x 3 block): primal infeasible
    dims = {'l': 1, 'q': [], 's': [3]}
    G = np.concatenate([[-1.0], np.eye(3).ravel(order='F')]).reshape(10, 1)
    h = np.concatenate([[0.0], (-np.eye(3)).ravel(order='F')])
    conelp_case('sdp_pinf', np.array([1.0]), G, h, dims)


if __name__ == "__main__":
    doc_sdp()
    maxcut(20, 0)
    maxcut(60, 1)
    maxcut(150, 2)          # beyond the LDS-resident Jacobi of the device code (order <= 101 / 142)
    mixed()
    qp()
    pinf()

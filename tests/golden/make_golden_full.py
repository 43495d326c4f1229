"""Full-size fixtures: the REAL reference (cvxopt built from /root/reference by oracle/build_ref.sh) run on the BASELINE
configurations at the sizes the GPU tests use, in the build container.  Takes several minutes; committed together with
its outputs (tests/golden/full_*.npz) so that the GPU box can check the device loops and the hook against reference-
produced numbers instead of hand-typed constants:

    python tests/golden/make_golden_full.py [qp8192] [socp] [sparse46] [elasticity] [batch64] [batch512] [qp2048]

For every run the fixture holds the iteration count, the final objectives at full precision, the final x (and z / a
sample of them), the per-iteration table the reference driver prints with options['show_progress'] (pcost, dcost, gap,
pres, dres [, k/t]: 5 significant digits), and a full-precision digest of the Nesterov-Todd scaling of every factor()
call (the 2-norm of W['di'], the sum of W['beta']) taken by wrapping the reference's own factory.
"""
import contextlib
import io
import os
import re
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refloader            # noqa: E402
from cvxopt_amd import synth            # noqa: E402

cvx = refloader.load()
from cvxopt import matrix, spmatrix, solvers, misc, blas   # noqa: E402

LINE = re.compile(r"^\s*(\d+):\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)(?:\s+(\S+))?\s*$")


def run_logged(fn, factory_name):
    """fn() with show_progress on, stdout parsed into the per-iteration table; the named misc.kkt_* factory is wrapped to
    record a digest of W at every factor(W, ...)."""
    digests = []
    orig = getattr(misc, factory_name)

    def wrapped(*a, **k):
        fac = orig(*a, **k)

        def factor(W, *rest):
            d = [float(blas.nrm2(W['di'])) if len(W['di']) else 0.0, float(sum(W['beta'])) if W['beta'] else 0.0]
            digests.append(d)
            return fac(W, *rest)
        return factor
    setattr(misc, factory_name, wrapped)
    buf = io.StringIO()
    old = solvers.options.get('show_progress', True)
    solvers.options['show_progress'] = True
    t = time.perf_counter()
    try:
        with contextlib.redirect_stdout(buf):
            sol = fn()
    finally:
        setattr(misc, factory_name, orig)
        solvers.options['show_progress'] = old
    t = time.perf_counter() - t
    rows = []
    for ln in buf.getvalue().splitlines():
        mm = LINE.match(ln)
        if mm:
            rows.append([float(v) if v is not None else np.nan for v in mm.groups()[1:]])
    return sol, np.array(rows), np.array(digests), t


def spm(A):
    A = A.tocoo()
    return spmatrix(A.data.tolist(), A.row.tolist(), A.col.tolist(), A.shape)


def qp_dense(n, m, name):
    pr = synth.dense_qp(n, m, seed=0)
    sol, table, dig, t = run_logged(lambda: solvers.coneqp(matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']),
                                                           kktsolver='chol2'), 'kkt_chol2')
    rec = {'n': n, 'm': m, 'seed': 0, 'iterations': sol['iterations'], 'pobj': sol['primal objective'],
           'dobj': sol['dual objective'], 'gap': sol['gap'], 'status_optimal': int(sol['status'] == 'optimal'),
           'x': np.array(sol['x']).ravel(), 'z': np.array(sol['z']).ravel(), 's': np.array(sol['s']).ravel(),
           'table': table, 'w_digest': dig, 'reference_seconds': t}
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **rec)
    print("wrote %s: %d iterations, pobj %.12e, %.1f s" % (name, sol['iterations'], sol['primal objective'], t))


def socp_full():
    n, N, r = 2048, 1024, 8
    pr = synth.socp(n=n, ncones=N, r=r, seed=0)
    sol, table, dig, t = run_logged(lambda: solvers.conelp(matrix(pr['c']), matrix(pr['G']), matrix(pr['h']), pr['dims'],
                                                           kktsolver='chol'), 'kkt_chol')
    rec = {'n': n, 'N': N, 'r': r, 'seed': 0, 'iterations': sol['iterations'], 'pobj': sol['primal objective'],
           'dobj': sol['dual objective'], 'gap': sol['gap'], 'status_optimal': int(sol['status'] == 'optimal'),
           'x': np.array(sol['x']).ravel(), 'z': np.array(sol['z']).ravel(), 's': np.array(sol['s']).ravel(),
           'table': table, 'w_digest': dig, 'reference_seconds': t}
    np.savez_compressed(os.path.join(HERE, 'full_socp2048.npz'), **rec)
    print("wrote full_socp2048: %d iterations, pobj %.12e, %.1f s" % (sol['iterations'], sol['primal objective'], t))


def elasticity():
    """the irregular stand-in of config 4 (round 5): box QP on the 3-dof stiffness matrix of a 3000-node tetrahedral mesh (n = 9000)
    through the reference's sparse kkt_chol2 branch (CHOLMOD replaced by the SuperLU shim: solutions are ordering independent)"""
    import scipy.sparse as sp
    nodes = 3000
    P = synth.tet_mesh_elasticity(nodes, seed=7)
    n = P.shape[0]
    rng = np.random.default_rng(7)
    q = rng.standard_normal(n)
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()
    h = np.ones(2 * n)
    sol, table, dig, t = run_logged(lambda: solvers.coneqp(spm(sp.tril(P)), matrix(q), spm(G), matrix(h), kktsolver='chol2'),
                                    'kkt_chol2')
    x = np.array(sol['x']).ravel()
    z = np.array(sol['z']).ravel()
    rec = {'nodes': nodes, 'n': n, 'seed': 7, 'iterations': sol['iterations'], 'pobj': sol['primal objective'],
           'dobj': sol['dual objective'], 'gap': sol['gap'], 'status_optimal': int(sol['status'] == 'optimal'),
           'x': x, 'z': z, 'table': table, 'w_digest': dig, 'reference_seconds': t,
           'note': "reference sparse branch of kkt_chol2 with oracle/cholmod_shim.py (SuperLU) in place of CHOLMOD"}
    np.savez_compressed(os.path.join(HERE, 'full_elasticity9000.npz'), **rec)
    print("wrote full_elasticity9000: %d iterations, pobj %.12e, %.1f s" % (sol['iterations'], sol['primal objective'], t))


def sparse46():
    import scipy.sparse as sp
    k = 46
    n = k ** 3
    P = synth.grid_laplacian(k)
    rng = np.random.default_rng(0)
    q = rng.standard_normal(n)
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()
    h = np.ones(2 * n)
    sol, table, dig, t = run_logged(lambda: solvers.coneqp(spm(sp.tril(P)), matrix(q), spm(G), matrix(h), kktsolver='chol2'),
                                    'kkt_chol2')
    x = np.array(sol['x']).ravel()
    z = np.array(sol['z']).ravel()
    rec = {'k': k, 'n': n, 'seed': 0, 'iterations': sol['iterations'], 'pobj': sol['primal objective'],
           'dobj': sol['dual objective'], 'gap': sol['gap'], 'status_optimal': int(sol['status'] == 'optimal'),
           'x': x, 'z_sample': z[::97].copy(), 'z_norm': float(np.linalg.norm(z)), 'table': table, 'w_digest': dig,
           'reference_seconds': t,
           'note': "reference sparse branch of kkt_chol2 with oracle/cholmod_shim.py (SuperLU) in place of CHOLMOD"}
    np.savez_compressed(os.path.join(HERE, 'full_sparse46.npz'), **rec)
    print("wrote full_sparse46: %d iterations, pobj %.12e, %.1f s" % (sol['iterations'], sol['primal objective'], t))


def batch64():
    n, m, B = 512, 1024, 64
    its, pobj, dobj, xs = [], [], [], []
    solvers.options['show_progress'] = False
    t = time.perf_counter()
    for i in range(B):
        pr = synth.dense_qp(n, m, seed=i)
        sol = solvers.coneqp(matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']), kktsolver='chol2')
        assert sol['status'] == 'optimal'
        its.append(sol['iterations'])
        pobj.append(sol['primal objective'])
        dobj.append(sol['dual objective'])
        xs.append(np.array(sol['x']).ravel())
    t = time.perf_counter() - t
    np.savez_compressed(os.path.join(HERE, 'full_batch64.npz'), n=n, m=m, B=B, iterations=np.array(its),
                        pobj=np.array(pobj), dobj=np.array(dobj), x=np.array(xs), reference_seconds=t)
    print("wrote full_batch64: %d problems, iterations %s..., %.1f s" % (B, its[:6], t))


def batch512():
    """BASELINE configs[4], one GPU's share: 512 individual reference solves (seed = index).  Per problem: iteration count, both
    objectives, ||x||_2 and every 8th entry of x (the full x of the first 64 is in full_batch64)."""
    n, m, B = 512, 1024, 512
    its, pobj, dobj, xs, xn = [], [], [], [], []
    solvers.options['show_progress'] = False
    t = time.perf_counter()
    for i in range(B):
        pr = synth.dense_qp(n, m, seed=i)
        sol = solvers.coneqp(matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']), kktsolver='chol2')
        assert sol['status'] == 'optimal'
        x = np.array(sol['x']).ravel()
        its.append(sol['iterations'])
        pobj.append(sol['primal objective'])
        dobj.append(sol['dual objective'])
        xs.append(x[::8].copy())
        xn.append(float(np.linalg.norm(x)))
    t = time.perf_counter() - t
    np.savez_compressed(os.path.join(HERE, 'full_batch512.npz'), n=n, m=m, B=B, iterations=np.array(its),
                        pobj=np.array(pobj), dobj=np.array(dobj), x_sample=np.array(xs), x_norm=np.array(xn),
                        reference_seconds=t)
    print("wrote full_batch512: %d problems, iterations min %d max %d, %.1f s" % (B, min(its), max(its), t))


if __name__ == "__main__":
    which = sys.argv[1:] or ['qp2048', 'batch64', 'socp', 'qp8192', 'sparse46']
    for w in which:
        if w == 'qp8192':
            qp_dense(8192, 16384, 'full_qp8192')
        elif w == 'qp2048':
            qp_dense(2048, 4096, 'full_qp2048')
        elif w == 'socp':
            socp_full()
        elif w == 'sparse46':
            sparse46()
        elif w == 'elasticity':
            elasticity()
        elif w == 'batch64':
            batch64()
        elif w == 'batch512':
            batch512()
        else:
            raise SystemExit("unknown fixture " + w)

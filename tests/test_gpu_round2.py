"""Round-2 additions at the drop-in boundary: H re-upload on every factor(), the sparse engine's S + A'A fallback and
sparse-A Schur complement, `cvxopt_amd.solvers` with kktsolver=None / a callable / kktreg, options['show_progress'] in the
device-resident loops.  Parity against the NumPy oracle, the dense engine and the real reference (oracle/_ref)."""
import contextlib
import io
import re

import numpy as np
import pytest
import scipy.sparse as sp

import cvxopt_amd
from cvxopt_amd import kkt, synth
from helpers import relerr
from oracle import kkt_oracle as ko

pytestmark = pytest.mark.gpu
LINE = re.compile(r"^\s*(\d+):\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)(?:\s+(\S+))?\s*$")


class FakeSp(object):
    """minimal stand-in for cvxopt.spmatrix (size + CCS)"""

    def __init__(self, A):
        A = sp.csc_matrix(A)
        A.sort_indices()
        self.size = A.shape
        self.CCS = (A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(float))


# ---- H is re-read at every factor(W, H) ------------------------------------------------------------------
@pytest.mark.parametrize("n,m", [(300, 420), (1100, 1500)])
def test_inplace_edit_of_H_between_factor_calls_is_seen(n, m):
    """One off-diagonal entry of P is rewritten in place between two factor() calls (round 1 sampled ~65k entries and the
    trace: this edit was invisible and the stale HBM copy was reused)."""
    pr = synth.dense_qp(n, m, seed=5)
    P = pr['P'].copy(order='F')
    W = synth.random_scaling(pr['dims'], seed=6, spread=1.0)
    rng = np.random.default_rng(0)
    bx, bz = rng.standard_normal(n), rng.standard_normal(m)
    f = kkt.kkt_chol2(pr['G'], pr['dims'], np.zeros((0, n)))
    oracle = ko.KktChol2(pr['G'], pr['dims'], np.zeros((0, n)))
    for step in range(3):
        if step == 1:
            P[n - 3, 1] += 0.37                     # lower triangle, far from the diagonal, not on any sampling stride
        if step == 2:
            P[5, 5] += 1.5
        x, y, z = bx.copy(), np.zeros(0), bz.copy()
        f(W, P)(x, y, z)
        xo, yo, zo = bx.copy(), np.zeros(0), bz.copy()
        oracle.factor(W, P)(xo, yo, zo)
        assert relerr(x, xo) < 1e-9 and relerr(z, zo) < 1e-9, step
    # the opt-out for callers that guarantee immutability: the same object is not uploaded again
    kkt.options["assume_constant_H"] = True
    try:
        x1, z1 = bx.copy(), bz.copy()
        f(W, P)(x1, np.zeros(0), z1)
        P[n - 3, 1] -= 0.37                         # now (deliberately) invisible
        x2, z2 = bx.copy(), bz.copy()
        f(W, P)(x2, np.zeros(0), z2)
        assert np.array_equal(x1, x2)
    finally:
        kkt.options["assume_constant_H"] = False
    f.engine.close()


def test_sparse_H_in_dense_mode_and_type_switch_is_refused():
    n, m = 60, 90
    pr = synth.dense_qp(n, m, seed=1)
    W = synth.random_scaling(pr['dims'], seed=2, spread=1.0)
    Hs = sp.tril(sp.csc_matrix(np.where(np.abs(pr['P']) > 0.02, pr['P'], 0.0)))
    bx, bz = np.ones(n), np.ones(m)
    f = kkt.kkt_chol2(pr['G'], pr['dims'], np.zeros((0, n)))          # dense G -> dense engine, sparse H densified
    x, z = bx.copy(), bz.copy()
    f(W, FakeSp(Hs))(x, np.zeros(0), z)
    xo, zo = bx.copy(), bz.copy()
    ko.KktChol2(pr['G'], pr['dims'], np.zeros((0, n))).factor(W, np.asfortranarray(Hs.toarray()))(xo, np.zeros(0), zo)
    assert relerr(x, xo) < 1e-9
    f.engine.close()
    # sparse G + sparse H -> sparse engine; a dense H afterwards cannot be honoured by that handle
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()
    dims = {'l': 2 * n, 'q': [], 's': []}
    f = kkt.kkt_chol2(FakeSp(G), dims, np.zeros((0, n)))
    W2 = synth.random_scaling(dims, seed=3, spread=1.0)
    f(W2, FakeSp(Hs))
    with pytest.raises(TypeError):
        f(W2, np.asfortranarray(Hs.toarray()))
    f.engine.close()


# ---- sparse engine: S + A'A fallback (misc.py:1433-1447) and sparse A (misc.py:1483-1487) ------------------
def _free_variable_lp(n, m, p, seed):
    """inequalities touch only the first m < n variables; the rest are pinned by A alone: S = G'D^2G is singular"""
    rng = np.random.default_rng(seed)
    G = sp.hstack([sp.vstack([sp.eye(m), -sp.eye(m)]), sp.csc_matrix((2 * m, n - m))]).tocsc()
    A = sp.random(p, n, density=0.2, random_state=seed, format='csc') + sp.hstack(
        [sp.csc_matrix((p, n - p)), sp.eye(p)]).tocsc()
    x0 = rng.uniform(-0.5, 0.5, n)
    h = np.ones(2 * m)
    b = A @ x0
    c = rng.standard_normal(n)
    c[m:] = 0.0                                       # bounded: the free variables carry no cost
    return c, G.tocsc(), h, sp.csc_matrix(A), b


def test_sparse_singular_S_falls_back_to_S_plus_AtA_hook_level():
    n, m, p = 90, 50, 40
    c, G, h, A, b = _free_variable_lp(n, m, p, seed=3)
    dims = {'l': 2 * m, 'q': [], 's': []}
    W = synth.random_scaling(dims, seed=1, spread=1.0)
    rng = np.random.default_rng(1)
    bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(2 * m)
    fs = kkt.kkt_chol2(FakeSp(G), dims, FakeSp(A))                  # sparse engine, sparse A
    x, y, z = bx.copy(), by.copy(), bz.copy()
    fs(W)(x, y, z)
    assert fs.engine._mode == "sparse" and fs.engine._sparse_singular
    Gd, Ad = np.asfortranarray(G.toarray()), np.asfortranarray(A.toarray())
    xo, yo, zo = bx.copy(), by.copy(), bz.copy()
    ko.KktChol2(Gd, dims, Ad).factor(W, None)(xo, yo, zo)
    assert relerr(x, xo) < 1e-8 and relerr(y, yo) < 1e-8 and relerr(z, zo) < 1e-8
    res = ko.kkt_residual(None, Ad, Gd, W, dims, bx, by, bz, x, y, z)
    assert res < 1e-10, res
    # a second factorisation with another scaling stays in the S + A'A mode
    W2 = synth.random_scaling(dims, seed=9, spread=1.0)
    x, y, z = bx.copy(), by.copy(), bz.copy()
    fs(W2)(x, y, z)
    assert ko.kkt_residual(None, Ad, Gd, W2, dims, bx, by, bz, x, y, z) < 1e-10
    fs.engine.close()


def test_sparse_singular_S_device_loop_matches_reference(ref_cvxopt):
    """the LP of ADVICE r1: free variables pinned only by A.  conelp_device used to raise 'Rank(A) < p or Rank([G; A]) < n'."""
    from cvxopt import matrix, spmatrix, solvers
    n, m, p = 70, 40, 30
    c, G, h, A, b = _free_variable_lp(n, m, p, seed=11)

    def spm(M):
        M = M.tocoo()
        return spmatrix(M.data.tolist(), M.row.tolist(), M.col.tolist(), M.shape)
    ref = solvers.conelp(matrix(c), spm(G), matrix(h), A=spm(A), b=matrix(b), kktsolver='chol2')
    sol = cvxopt_amd.conelp_device(c, FakeSp(G), h, None, FakeSp(A), b, kktsolver='chol2')
    assert sol['status'] == ref['status'] == 'optimal'
    assert sol['iterations'] == ref['iterations']
    assert abs(sol['primal objective'] - ref['primal objective']) <= 1e-8 * max(1.0, abs(ref['primal objective']))
    assert relerr(sol['x'][:m], np.array(ref['x']).ravel()[:m]) < 1e-5


def test_sparse_A_schur_complement_beyond_the_old_cap():
    """p = 700 sparse equality constraints (round 1 densified A and refused p > 512): sparse engine vs the dense engine"""
    k = 12
    n = k ** 3
    P = synth.grid_laplacian(k)
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()
    p = 700
    rng = np.random.default_rng(2)
    rows = np.repeat(np.arange(p), 3)
    cols = rng.integers(0, n, size=3 * p)
    A = sp.csc_matrix((rng.standard_normal(3 * p), (rows, cols)), shape=(p, n)) + sp.csc_matrix(
        (np.ones(p), (np.arange(p), rng.permutation(n)[:p])), shape=(p, n))
    dims = {'l': 2 * n, 'q': [], 's': []}
    W = synth.random_scaling(dims, seed=4, spread=1.0)
    bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(2 * n)
    fs = kkt.kkt_chol2(FakeSp(G), dims, FakeSp(A))
    x, y, z = bx.copy(), by.copy(), bz.copy()
    fs(W, FakeSp(sp.tril(P)))(x, y, z)
    assert fs.engine._mode == "sparse"
    fd = kkt.kkt_chol2(np.asfortranarray(G.toarray()), dims, np.asfortranarray(A.toarray()))
    xd, yd, zd = bx.copy(), by.copy(), bz.copy()
    fd(W, np.asfortranarray(P.toarray()))(xd, yd, zd)
    assert relerr(x, xd) < 1e-8 and relerr(y, yd) < 1e-7 and relerr(z, zd) < 1e-8
    # operator form of a sparse A (cvxopt_amd.solvers' Af closure)
    v = rng.standard_normal(n)
    assert relerr(fs.engine.product(1, False, v), A @ v) < 1e-12
    u = rng.standard_normal(p)
    assert relerr(fs.engine.product(1, True, u), A.T @ u) < 1e-12
    fs.engine.close()
    fd.engine.close()


# ---- cvxopt_amd.solvers: kktsolver=None / callable / kktreg, show_progress ---------------------------------
def test_solvers_default_and_callable_kktsolver(ref_cvxopt):
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    pr = synth.dense_qp(40, 70, seed=3, p=4)
    args = (matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']))
    kw = dict(A=matrix(pr['A']), b=matrix(pr['b']))
    ref = solvers.coneqp(*args, **kw)
    for ks in (None, 'ldl', kkt.kktsolver_qp(args[2], pr['dims'], kw['A'], args[0])):
        sol = gs.coneqp(*args, kktsolver=ks, **kw)
        assert sol['status'] == 'optimal' and sol['iterations'] == ref['iterations']
        assert abs(sol['primal objective'] - ref['primal objective']) <= 1e-9 * max(1.0, abs(ref['primal objective']))
    with pytest.raises(ValueError):
        gs.coneqp(*args, kktsolver='qr', **kw)              # not a coneqp solver in the reference either (coneprog.py:1810)
    sp_ = synth.socp(16, 5, 4, seed=2, ml=3)
    cargs = (matrix(sp_['c']), matrix(sp_['G']), matrix(sp_['h']), sp_['dims'])
    ref = solvers.conelp(*cargs)
    for ks in (None, 'qr', 'chol'):
        sol = gs.conelp(*cargs, kktsolver=ks)
        assert sol['status'] == 'optimal' and sol['iterations'] == ref['iterations']
        assert abs(sol['primal objective'] - ref['primal objective']) <= 1e-8 * max(1.0, abs(ref['primal objective']))


def test_solvers_kktreg_reaches_the_ldl_engine(ref_cvxopt):
    """options['kktreg'] with kktsolver='ldl' (coneprog.py:575, :1973) was dropped on the way to the GPU factory in round 1.
    A duplicated equality constraint (Rank(A) < p) makes the unregularised KKT matrix singular: only with the regularisation
    does the problem solve (the reference: optimal in 8 iterations with kktreg = 1e-8, 100 useless iterations without)."""
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    pr = synth.dense_qp(30, 50, seed=8, p=3)
    A = np.vstack([pr['A'], pr['A'][:1]])
    b = np.concatenate([pr['b'], pr['b'][:1]])
    args = (matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']))
    kw = dict(A=matrix(A), b=matrix(b))
    opts = {'show_progress': False, 'kktreg': 1e-8}
    ref = solvers.coneqp(*args, kktsolver='ldl', options=opts, **kw)
    assert ref['status'] == 'optimal'
    for loop in (True, False):
        sol = gs.coneqp(*args, kktsolver='ldl', options=opts, device_loop=loop, **kw)
        assert sol['status'] == 'optimal' and sol['iterations'] == ref['iterations'], loop
        assert abs(sol['primal objective'] - ref['primal objective']) <= 1e-8 * max(1.0, abs(ref['primal objective']))
        assert relerr(np.array(sol['x']).ravel(), np.array(ref['x']).ravel()) < 1e-6
    with pytest.raises((ValueError, ArithmeticError)):          # without it the static-order factorisation hits the singular K
        gs.coneqp(*args, kktsolver='ldl', options={'show_progress': False}, **kw)
    with pytest.raises(ValueError):
        gs.coneqp(*args, kktsolver='ldl', options={'kktreg': -1.0}, **kw)


def _table(text):
    rows = []
    for ln in text.splitlines():
        mm = LINE.match(ln)
        if mm:
            rows.append([float(v) if v is not None else np.nan for v in mm.groups()[1:]])
    return np.array(rows)


def test_show_progress_lines_of_the_device_loops_match_the_reference(ref_cvxopt):
    """options['show_progress'] (the reference's default) is honoured by the device-resident loops: same header, one line per
    iteration with the reference's format, same numbers to the printed precision, same closing line"""
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    pr = synth.dense_qp(50, 90, seed=4)
    args = (matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']))
    bufr, bufg = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(bufr):
        solvers.coneqp(*args, options={'show_progress': True})
    with contextlib.redirect_stdout(bufg):
        gs.coneqp(*args, options={'show_progress': True})
    tr, tg = _table(bufr.getvalue()), _table(bufg.getvalue())
    assert tr.shape == tg.shape and tr.shape[0] > 3
    assert bufr.getvalue().splitlines()[0] == bufg.getvalue().splitlines()[0]                # header
    assert bufr.getvalue().strip().splitlines()[-1] == bufg.getvalue().strip().splitlines()[-1] == "Optimal solution found."
    assert np.allclose(tr[:, :2], tg[:, :2], rtol=2e-4, atol=1e-9)
    sq = synth.socp(20, 4, 5, seed=6, ml=4)
    cargs = (matrix(sq['c']), matrix(sq['G']), matrix(sq['h']), sq['dims'])
    bufr, bufg = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(bufr):
        solvers.conelp(*cargs, options={'show_progress': True})
    with contextlib.redirect_stdout(bufg):
        gs.conelp(*cargs, options={'show_progress': True})
    tr, tg = _table(bufr.getvalue()), _table(bufg.getvalue())
    assert tr.shape == tg.shape and tr.shape[1] == 6
    assert np.allclose(tr[:, :2], tg[:, :2], rtol=2e-4, atol=1e-9)
    assert bufr.getvalue().splitlines()[0] == bufg.getvalue().splitlines()[0]
    # silent when asked
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        gs.coneqp(*args, options={'show_progress': False})
    assert buf.getvalue() == ""

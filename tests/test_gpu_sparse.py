"""Sparse path (BASELINE configs[3] class: sparse coneqp via cvxopt.spmatrix, supernodal HIP Cholesky).
ssget id 1288 cannot be fetched offline; the stand-in of the same class is a 2-D / 3-D Laplacian P with box
constraints G = [I; -I].  Parity: the sparse device engine against (i) the dense device engine
and the dense oracle on the same problem, (ii) the unmodified reference running its own sparse kkt_chol2 branch
(sp_dgemm / sp_dsyrk assembly + the SciPy-backed cholmod shim)."""
import numpy as np
import pytest
import scipy.sparse as sp

from cvxopt_amd import kkt, synth
from oracle import kkt_oracle as ko
from helpers import record, relerr

pytestmark = pytest.mark.gpu


class FakeSp(object):
    """minimal stand-in for cvxopt.spmatrix (size + CCS) so the hook can be driven without cvxopt"""

    def __init__(self, A):
        A = sp.csc_matrix(A)
        A.sort_indices()
        self.size = A.shape
        self.CCS = (A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(float))
        self._A = A

    def __len__(self):
        return self._A.nnz


def laplace2d(nx, ny, shift=1e-2):
    ex, ey = np.ones(nx), np.ones(ny)
    Tx = sp.diags([-ex[:-1], 2 * ex, -ex[:-1]], [-1, 0, 1])
    Ty = sp.diags([-ey[:-1], 2 * ey, -ey[:-1]], [-1, 0, 1])
    return (sp.kron(sp.eye(ny), Tx) + sp.kron(Ty, sp.eye(nx)) + shift * sp.eye(nx * ny)).tocsc()


def laplace3d(k, shift=1e-2):
    e = np.ones(k)
    T = sp.diags([-e[:-1], 2 * e, -e[:-1]], [-1, 0, 1])
    I = sp.eye(k)
    return (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T) + shift * sp.eye(k ** 3)).tocsc()


def box(n):
    return sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()


@pytest.mark.parametrize("maker,arg", [(laplace2d, (7, 5)), (laplace2d, (40, 33)), (laplace3d, (9,)), (laplace2d, (150, 120))])
def test_sparse_factor_solve_matches_dense_oracle(maker, arg):
    P = maker(*arg)
    n = P.shape[0]
    G = box(n)
    dims = {'l': 2 * n, 'q': [], 's': []}
    A = np.zeros((0, n))
    f = kkt.kkt_chol2(FakeSp(G), dims, A)
    rng = np.random.default_rng(n)
    for it in range(2):
        W = synth.random_scaling(dims, seed=it, spread=1.5)
        bx, bz = rng.standard_normal(n), rng.standard_normal(2 * n)
        x, y, z = bx.copy(), np.zeros(0), bz.copy()
        f(W, FakeSp(sp.tril(P)))(x, y, z)
        assert f.engine._mode == "sparse"
        if n <= 3000:
            xo, yo, zo = bx.copy(), np.zeros(0), bz.copy()
            ko.KktChol2(G.toarray(), dims, A).factor(W, P.toarray())(xo, yo, zo)
            assert relerr(x, xo) < 1e-9 and relerr(z, zo) < 1e-9
        # residual of the reduced system (size independent)
        S = (P + G.T @ sp.diags(W['di'] ** 2) @ G).tocsc()
        rhs = bx + G.T @ (W['di'] ** 2 * bz)
        assert np.linalg.norm(S @ x - rhs) / np.linalg.norm(rhs) < 1e-11
        assert relerr(z, W['di'] * (G @ x) - W['di'] * bz) < 1e-12
    st = f.engine.sparse_stats()
    assert st['nnzL'] >= n and st['supernodes'] >= 1
    f.engine.close()


@pytest.mark.parametrize("maker,arg", [(laplace3d, (20,)), (laplace2d, (150, 120))])
def test_sparse_factor_reads_nothing_it_has_not_written(maker, arg, knobs):
    """round 4: a factorisation clears the supernodes' panels and, tile by tile, the update matrices of the big fronts -- not the
    whole store.  With the store poisoned (every byte 0xff = NaN, test knob MI355KKT_SPARSE_POISON) before each factorisation the
    solutions must be bit for bit those of the unpoisoned run: nothing outside the cleared regions is ever read."""
    P = maker(*arg)
    n = P.shape[0]
    G = box(n)
    dims = {'l': 2 * n, 'q': [], 's': []}
    A = np.zeros((0, n))
    rng = np.random.default_rng(n)
    bx, bz = rng.standard_normal(n), rng.standard_normal(2 * n)
    out = []
    for poison in (False, True, True):
        if poison:
            knobs.setenv("MI355KKT_SPARSE_POISON", "1")
        f = kkt.kkt_chol2(FakeSp(G), dims, A)
        sols = []
        for it in range(2):                       # the second factorisation runs over the first one's left-overs
            W = synth.random_scaling(dims, seed=it, spread=1.5)
            x, y, z = bx.copy(), np.zeros(0), bz.copy()
            f(W, FakeSp(sp.tril(P)))(x, y, z)
            assert np.all(np.isfinite(x)) and np.all(np.isfinite(z))
            sols.append((x, z))
        S = (P + G.T @ sp.diags(W['di'] ** 2) @ G).tocsc()
        rhs = bx + G.T @ (W['di'] ** 2 * bz)
        assert np.linalg.norm(S @ sols[-1][0] - rhs) / np.linalg.norm(rhs) < 1e-11
        out.append(sols)
        f.engine.close()
    for a, b in ((out[0], out[1]), (out[1], out[2])):
        for (xa, za), (xb, zb) in zip(a, b):
            assert np.array_equal(xa, xb) and np.array_equal(za, zb)


def test_sparse_general_G_pattern_and_dense_engine_agree():
    """G with off-diagonal couplings (S gets fill from G'D^2G) -- sparse engine vs the dense device engine."""
    rng = np.random.default_rng(3)
    n, m = 300, 500
    G = sp.random(m, n, density=0.01, random_state=4, format='csc') + sp.vstack([sp.eye(n), sp.csc_matrix((m - n, n))])
    P = laplace2d(20, 15)
    dims = {'l': m, 'q': [], 's': []}
    A = np.zeros((0, n))
    W = synth.random_scaling(dims, seed=1, spread=1.0)
    bx, bz = rng.standard_normal(n), rng.standard_normal(m)
    fs = kkt.kkt_chol2(FakeSp(G), dims, A)
    xs, ys, zs = bx.copy(), np.zeros(0), bz.copy()
    fs(W, FakeSp(sp.tril(P)))(xs, ys, zs)
    assert fs.engine._mode == "sparse"
    fd = kkt.kkt_chol2(np.asfortranarray(G.toarray()), dims, A)
    xd, yd, zd = bx.copy(), np.zeros(0), bz.copy()
    fd(W, np.asfortranarray(P.toarray()))(xd, yd, zd)
    assert relerr(xs, xd) < 1e-9 and relerr(zs, zd) < 1e-9
    fs.engine.close()
    fd.engine.close()


def test_sparse_not_positive_definite_raises():
    n = 50
    P = sp.csc_matrix((n, n))
    G = sp.vstack([sp.eye(n, format="csr")[:20], -sp.eye(n, format="csr")[:20]]).tocsc()       # only 20 of 50 variables constrained, P = 0
    dims = {'l': 40, 'q': [], 's': []}
    f = kkt.kkt_chol2(FakeSp(G), dims, np.zeros((0, n)))
    with pytest.raises(ArithmeticError):
        f(synth.random_scaling(dims, seed=0), None)
    f.engine.close()


def test_sparse_coneqp_drop_in_matches_reference_sparse_branch(ref_cvxopt):
    """The unmodified reference runs its own sparse kkt_chol2 branch (cholmod shim); same iterates expected.
    Reference probe: 2-D Laplacian box-QP n=400: optimal, 7 iterations, pobj -4.881627878645e+02."""
    from cvxopt import matrix, spmatrix, solvers, sparse
    nx = 20
    P = laplace2d(nx, nx)
    n = nx * nx
    Pc = sp.tril(P).tocoo()
    Pcv = spmatrix(list(Pc.data), list(map(int, Pc.row)), list(map(int, Pc.col)), (n, n))
    Gc = box(n).tocoo()
    Gcv = spmatrix(list(Gc.data), list(map(int, Gc.row)), list(map(int, Gc.col)), (2 * n, n))
    q = matrix(-np.ones(n))
    h = matrix(np.ones(2 * n))
    ref = solvers.coneqp(Pcv, q, Gcv, h, kktsolver='chol2')
    A = spmatrix([], [], [], (0, n))
    ks = kkt.kktsolver_qp(Gcv, {'l': 2 * n, 'q': [], 's': []}, A, Pcv)
    got = solvers.coneqp(Pcv, q, Gcv, h, kktsolver=ks)
    assert ks.engine._mode == "sparse"
    assert got['status'] == ref['status'] == 'optimal'
    assert got['iterations'] == ref['iterations']
    assert abs(got['primal objective'] - ref['primal objective']) <= 1e-9 * abs(ref['primal objective'])
    assert relerr(np.array(got['x']).ravel(), np.array(ref['x']).ravel()) < 1e-7
    ks.engine.close()


def test_sparse_resident_coneqp_matches_reference_sparse_branch(ref_cvxopt):
    """Device-resident loop in sparse mode (sparse residual products + supernodal factor/solve) vs the unmodified
    reference's sparse kkt_chol2 run of the same box-QP."""
    import cvxopt_amd
    from cvxopt import matrix, spmatrix, solvers
    nx = 20
    P = laplace2d(nx, nx)
    n = nx * nx
    Pc = sp.tril(P).tocoo()
    Pcv = spmatrix(list(Pc.data), list(map(int, Pc.row)), list(map(int, Pc.col)), (n, n))
    Gc = box(n).tocoo()
    Gcv = spmatrix(list(Gc.data), list(map(int, Gc.row)), list(map(int, Gc.col)), (2 * n, n))
    q, h = -np.ones(n), np.ones(2 * n)
    ref = solvers.coneqp(Pcv, matrix(q), Gcv, matrix(h), kktsolver='chol2')
    got = cvxopt_amd.coneqp_lp(Pcv, q, Gcv, h)
    assert got['status'] == ref['status'] == 'optimal'
    assert got['iterations'] == ref['iterations']
    assert abs(got['primal objective'] - ref['primal objective']) <= 1e-9 * abs(ref['primal objective'])
    assert abs(got['dual objective'] - ref['dual objective']) <= 1e-9 * abs(ref['dual objective'])
    assert relerr(got['x'], np.array(ref['x']).ravel()) < 1e-7
    assert relerr(got['z'], np.array(ref['z']).ravel()) < 1e-6


@pytest.mark.parametrize("maker,arg", [(laplace2d, (60, 45)), (laplace3d, (14,))])
def test_sparse_resident_coneqp_matches_dense_resident(maker, arg):
    """Same problem through the sparse engine and (densified) through the dense engine: same iterates."""
    import time
    import cvxopt_amd
    P = maker(*arg)
    n = P.shape[0]
    G = box(n)
    rng = np.random.default_rng(n)
    q, h = rng.standard_normal(n), 0.5 + rng.random(2 * n)
    t = time.perf_counter()
    a = cvxopt_amd.coneqp_lp(FakeSp(sp.tril(P)), q, FakeSp(G), h)
    ta = time.perf_counter() - t
    b = cvxopt_amd.coneqp_lp(np.asfortranarray(P.toarray()), q, np.asfortranarray(G.toarray()), h)
    print("sparse resident coneqp n=%d: %.3f s, %d iterations" % (n, ta, a['iterations']))
    assert a['status'] == b['status'] == 'optimal' and a['iterations'] == b['iterations']
    assert abs(a['primal objective'] - b['primal objective']) <= 1e-9 * max(1.0, abs(b['primal objective']))
    assert relerr(a['x'], b['x']) < 1e-7 and relerr(a['z'], b['z']) < 1e-6


@pytest.mark.parametrize("maker,arg,p", [(laplace2d, (30, 24), 5), (laplace3d, (10,), 12), (laplace3d, (16,), 70)])
def test_sparse_factor_solve_with_equalities_matches_dense_oracle(maker, arg, p):
    """p > 0 on the sparse engine: Asct = L^-1 P A' through the supernodal forward solve, K = Asct'Asct dense
    (reference misc.py:1464-1487 / :1528-1558, sparse branch) vs the dense NumPy oracle of the same KKT system."""
    P = maker(*arg)
    n = P.shape[0]
    G = box(n)
    rng = np.random.default_rng(p)
    A = rng.standard_normal((p, n))
    dims = {'l': 2 * n, 'q': [], 's': []}
    W = synth.random_scaling(dims, seed=1, spread=1.0)
    f = kkt.kkt_chol2(FakeSp(G), dims, A)
    solve = f(W, FakeSp(sp.tril(P)))
    assert f.engine._mode == "sparse"
    bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(2 * n)
    x, y, z = bx.copy(), by.copy(), bz.copy()
    solve(x, y, z)
    xo, yo, zo = bx.copy(), by.copy(), bz.copy()
    ko.KktChol2(G.toarray(), dims, A).factor(W, P.toarray())(xo, yo, zo)
    assert relerr(x, xo) < 1e-9 and relerr(y, yo) < 1e-8 and relerr(z, zo) < 1e-9
    f.engine.close()


def test_sparse_coneqp_with_equalities_drop_in(ref_cvxopt):
    from cvxopt import matrix, spmatrix, solvers
    nx = 16
    P = laplace2d(nx, nx)
    n = nx * nx
    Pc = sp.tril(P).tocoo()
    Pcv = spmatrix(list(Pc.data), list(map(int, Pc.row)), list(map(int, Pc.col)), (n, n))
    Gc = box(n).tocoo()
    Gcv = spmatrix(list(Gc.data), list(map(int, Gc.row)), list(map(int, Gc.col)), (2 * n, n))
    rng = np.random.default_rng(0)
    A = matrix(rng.standard_normal((3, n)) / np.sqrt(n))
    b = matrix(np.zeros(3))
    q, h = matrix(-np.ones(n)), matrix(np.ones(2 * n))
    ref = solvers.coneqp(Pcv, q, Gcv, h, A=A, b=b, kktsolver='chol2')
    ks = kkt.kktsolver_qp(Gcv, {'l': 2 * n, 'q': [], 's': []}, A, Pcv)
    got = solvers.coneqp(Pcv, q, Gcv, h, A=A, b=b, kktsolver=ks)
    assert ks.engine._mode == "sparse"
    assert got['status'] == ref['status'] == 'optimal' and got['iterations'] == ref['iterations']
    assert abs(got['primal objective'] - ref['primal objective']) <= 1e-9 * abs(ref['primal objective'])
    assert relerr(np.array(got['x']).ravel(), np.array(ref['x']).ravel()) < 1e-7
    assert relerr(np.array(got['y']).ravel(), np.array(ref['y']).ravel()) < 1e-6
    ks.engine.close()


def test_sparse_resident_loops_with_equalities(ref_cvxopt):
    """Device-resident coneqp / conelp on the sparse engine with a few equality constraints vs the dense engine."""
    import cvxopt_amd
    P = laplace2d(18, 14)
    n = P.shape[0]
    G = box(n)
    rng = np.random.default_rng(5)
    A = rng.standard_normal((4, n)) / np.sqrt(n)
    b = np.zeros(4)
    q, h = rng.standard_normal(n), 0.5 + rng.random(2 * n)
    a = cvxopt_amd.coneqp_lp(FakeSp(sp.tril(P)), q, FakeSp(G), h, A=A, b=b)
    d = cvxopt_amd.coneqp_lp(np.asfortranarray(P.toarray()), q, np.asfortranarray(G.toarray()), h, A=A, b=b)
    assert a['status'] == d['status'] == 'optimal' and a['iterations'] == d['iterations']
    assert relerr(a['x'], d['x']) < 1e-7 and relerr(a['y'], d['y']) < 1e-6
    c = rng.standard_normal(n)
    a = cvxopt_amd.conelp_device(c, FakeSp(G), np.ones(2 * n), A=A, b=b)
    d = cvxopt_amd.conelp_device(c, np.asfortranarray(G.toarray()), np.ones(2 * n), A=A, b=b)
    assert a['status'] == d['status'] == 'optimal' and a['iterations'] == d['iterations']
    assert relerr(a['x'], d['x']) < 1e-7 and abs(a['primal objective'] - d['primal objective']) <= 1e-9 * max(1, abs(d['primal objective']))


def _mesh_laplacian(n, seed=0):
    from scipy.spatial import Delaunay
    tri = Delaunay(np.random.default_rng(seed).random((n, 2))).simplices
    r = np.concatenate([tri[:, 0], tri[:, 1], tri[:, 2]])
    c = np.concatenate([tri[:, 1], tri[:, 2], tri[:, 0]])
    A = sp.coo_matrix((np.ones(len(r)), (r, c)), shape=(n, n))
    A = ((A + A.T) > 0).astype(float)
    return (sp.diags(np.asarray(A.sum(1)).ravel() + 0.05) - A).tocsc()


def _hub_laplacian(n, seed=0):
    rng = np.random.default_rng(seed)
    r = np.arange(1, n)
    c = (rng.random(n - 1) ** 3 * r).astype(int)          # attaches preferentially to early nodes: a few large hubs
    A = sp.coo_matrix((np.ones(n - 1), (r, c)), shape=(n, n))
    r2, c2 = rng.integers(0, n, n // 2), rng.integers(0, n, n // 2)
    A = A + sp.coo_matrix((np.ones(n // 2), (r2, c2)), shape=(n, n))
    A = ((A + A.T) > 0).astype(float).tolil()
    A.setdiag(0)
    A = A.tocsc()
    return (sp.diags(np.asarray(A.sum(1)).ravel() + 0.05) - A).tocsc()


@pytest.mark.parametrize("pattern", ["mesh", "hubs"])
@pytest.mark.parametrize("ordering", ["auto", "amd", "nd", "nd+amd-leaves", "nd-multilevel"])
def test_sparse_unstructured_patterns_with_every_ordering(pattern, ordering, knobs):
    """the orderings of csrc/ordering.cpp (approximate minimum degree, nested dissection with level-set / multilevel
    separators, minimum-degree leaves) only change the elimination order: same solution as the dense oracle"""
    env = {"auto": {}, "amd": {"MI355KKT_ORDERING": "amd"}, "nd": {"MI355KKT_ORDERING": "nd"},
           "nd+amd-leaves": {"MI355KKT_ORDERING": "nd", "MI355KKT_ND_LEAF_AMD": "1"},
           "nd-multilevel": {"MI355KKT_ORDERING": "nd", "MI355KKT_ND_MODE": "2"}}[ordering]
    for k, v in env.items():
        knobs.setenv(k, v)
    P = _mesh_laplacian(1800, seed=2) if pattern == "mesh" else _hub_laplacian(1500, seed=2)
    n = P.shape[0]
    G = box(n)
    dims = {'l': 2 * n, 'q': [], 's': []}
    A = np.zeros((0, n))
    f = kkt.kkt_chol2(FakeSp(G), dims, A)
    rng = np.random.default_rng(n)
    W = synth.random_scaling(dims, seed=3, spread=1.5)
    bx, bz = rng.standard_normal(n), rng.standard_normal(2 * n)
    x, y, z = bx.copy(), np.zeros(0), bz.copy()
    f(W, FakeSp(sp.tril(P)))(x, y, z)
    assert f.engine._mode == "sparse"
    xo, yo, zo = bx.copy(), np.zeros(0), bz.copy()
    ko.KktChol2(G.toarray(), dims, A).factor(W, P.toarray())(xo, yo, zo)
    assert relerr(x, xo) < 1e-9 and relerr(z, zo) < 1e-9
    S = (P + G.T @ sp.diags(W['di'] ** 2) @ G).tocsc()
    rhs = bx + G.T @ (W['di'] ** 2 * bz)
    assert np.linalg.norm(S @ x - rhs) / np.linalg.norm(rhs) < 1e-11
    f.engine.close()


@pytest.mark.parametrize("nodes", [1500, 20000])
def test_sparse_elasticity_stand_in(nodes):
    """The irregular stand-in for BASELINE configs[3] (VERDICT r4 missing 3): 3 degrees of freedom per node of a random tetrahedral
    mesh (synth.tet_mesh_elasticity: 3 x 3 blocks, ~48 entries per row, nothing grid-like) as the P of a box QP.  1500 nodes
    (n = 4500): every entry against the dense oracle (pinned to the reference's kkt_chol2, tests/test_oracle.py); 20000 nodes
    (n = 60000): against SuperLU's solution of the same reduced system and through its residual."""
    P = synth.tet_mesh_elasticity(nodes, seed=5)
    n = P.shape[0]
    G = box(n)
    dims = {'l': 2 * n, 'q': [], 's': []}
    A = np.zeros((0, n))
    f = kkt.kkt_chol2(FakeSp(G), dims, A)
    rng = np.random.default_rng(n)
    W = synth.random_scaling(dims, seed=4, spread=1.5)
    bx, bz = rng.standard_normal(n), rng.standard_normal(2 * n)
    x, y, z = bx.copy(), np.zeros(0), bz.copy()
    try:
        f(W, FakeSp(sp.tril(P)))(x, y, z)
        assert f.engine._mode == "sparse"
        st = f.engine.sparse_stats()
    finally:
        f.engine.close()
    S = (P + G.T @ sp.diags(W['di'] ** 2) @ G).tocsc()
    rhs = bx + G.T @ (W['di'] ** 2 * bz)
    res = float(np.linalg.norm(S @ x - rhs) / np.linalg.norm(rhs))
    if nodes <= 2000:
        xo, yo, zo = bx.copy(), np.zeros(0), bz.copy()
        ko.KktChol2(G.toarray(), dims, A).factor(W, P.toarray())(xo, yo, zo)
        ex, ez = relerr(x, xo), relerr(z, zo)
    else:
        import scipy.sparse.linalg as sla
        xo = sla.splu(S).solve(rhs)
        ex = relerr(x, xo)
        ez = relerr(z, W['di'] * (G @ xo - bz))
    record("config4_elasticity_%d" % nodes, n=n, nnzL=st['nnzL'], supernodes=st['supernodes'], levels=st['levels'], residual=res,
           x_relerr=ex, z_relerr=ez)
    assert res < 1e-11, res
    assert ex < 1e-9 and ez < 1e-9, (ex, ez)


def test_sparse_engine_reproduces_the_reference_doc_cholmod_example():
    """The only sparse-Cholesky known answer the reference holds (doc/source/spsolvers.rst:281-302 and :440-448,
    cholmod.linsolve / symbolic + numeric + solve on the 4x4 matrix e-A-pd).  Here S = H + G'D^2G with H = A - e0 e0',
    G = e0', d = 1, so the sparse engine factors exactly that A."""
    A = sp.csc_matrix(np.array([[10., 0, 3, 0], [0, 5, 0, -2], [3, 0, 5, 0], [0, -2, 0, 2]]))
    H = A.copy().tolil(); H[0, 0] = 9.0; H = sp.csc_matrix(H)
    G = sp.csc_matrix(np.array([[1., 0, 0, 0]]))
    dims = {'l': 1, 'q': [], 's': []}
    f = kkt.kkt_chol2(FakeSp(G), dims, np.zeros((0, 4)))
    W = {'d': np.ones(1), 'di': np.ones(1), 'v': [], 'beta': [], 'r': [], 'rti': []}
    solve = f(W, FakeSp(sp.tril(H)))
    assert f.engine._mode == "sparse"
    X = np.arange(8.0).reshape(4, 2, order='F')
    doc = np.array([[-1.46e-01, 4.88e-02], [1.33e+00, 4.00e+00], [4.88e-01, 1.17e+00], [2.83e+00, 7.50e+00]])
    for k in range(2):
        x, y, z = X[:, k].copy(), np.zeros(0), np.zeros(1)
        solve(x, y, z)
        assert np.allclose(x, doc[:, k], rtol=5e-3)                      # the printed transcript (3 digits)
        assert np.allclose(A @ x, X[:, k], rtol=0, atol=1e-13)
    f.engine.close()

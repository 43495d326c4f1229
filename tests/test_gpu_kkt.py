"""Hook-level parity: cvxopt_amd factories (HIP path through the C ABI) vs the oracle
(oracle/kkt_oracle.py, pinned to the reference) and vs the real reference kktsolvers / coneqp."""
import numpy as np
import pytest

from cvxopt_amd import kkt, synth
from oracle import kkt_oracle as ko

pytestmark = pytest.mark.gpu

# stated FP64 tolerances (SURVEY.md 8(d) "Parity tolerance to state")
SOLVE_RTOL = 1e-9          # ||u_gpu - u_oracle||_inf / ||u_oracle||_inf per solve at cond(S) <~ 1e8
RESID_TOL = 1e-12          # relative KKT residual of the GPU solution (or <= 10x the CPU oracle's)


def rand_rhs(rng, n, p, cdim):
    return rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(cdim)


def run_pair(factory, oracle, W, H, rhs):
    bx, by, bz = rhs
    x, y, z = bx.copy(), by.copy(), bz.copy()
    f = factory(W, H)
    f(x, y, z)
    xo, yo, zo = bx.copy(), by.copy(), bz.copy()
    oracle.factor(W, H)(xo, yo, zo)
    return (x, y, z), (xo, yo, zo)


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b)))) if a.size else 0.0


@pytest.mark.parametrize("n,m,p", [(64, 128, 0), (256, 512, 0), (200, 333, 7), (300, 700, 150), (129, 64, 129),
                                   (1000, 2048, 0), (5, 3, 2)])
@pytest.mark.parametrize("kind", ["chol2", "chol", "ldl", "ldl2"])
def test_factor_solve_matches_oracle_lp_cone(n, m, p, kind):
    pr = synth.dense_qp(n, m, seed=n + m + p, p=p)
    G, P = pr['G'], pr['P']
    A = pr.get('A', np.zeros((0, n)))
    dims = pr['dims']
    factory = {"chol2": kkt.kkt_chol2, "chol": kkt.kkt_chol, "ldl": kkt.kkt_ldl, "ldl2": kkt.kkt_ldl2}[kind](G, dims, A)
    oracle = ko.KktChol2(G, dims, A)
    rng = np.random.default_rng(11)
    for it in range(3):                              # W changes between factor calls, same handle
        W = synth.random_scaling(dims, seed=it, spread=1.0 + it)
        rhs = rand_rhs(rng, n, p, m)
        got, ref = run_pair(factory, oracle, W, P, rhs)
        # forward error of two backward-stable solvers differs by O(eps * cond(S)): scale the bound
        K2 = np.block([[oracle.S, A.T], [A, np.zeros((p, p))]])       # reduced 2x2 KKT matrix
        rtol = max(SOLVE_RTOL, 50 * np.finfo(float).eps * np.linalg.cond(K2))
        for g, r in zip(got, ref):
            assert relerr(g, r) < rtol, (relerr(g, r), rtol)
        res = ko.kkt_residual(P, A, G, W, dims, rhs[0], rhs[1], rhs[2], got[0], got[1], got[2])
        res_ref = ko.kkt_residual(P, A, G, W, dims, rhs[0], rhs[1], rhs[2], ref[0], ref[1], ref[2])
        assert res < max(RESID_TOL, 10.0 * res_ref), (res, res_ref)   # as accurate as LAPACK on the CPU
    factory.engine.close()


def test_h_none_is_zero_and_lower_triangle_only():
    n, m = 96, 300
    pr = synth.dense_qp(n, m, seed=3)
    G, dims, A = pr['G'], pr['dims'], np.zeros((0, n))
    W = synth.random_scaling(dims, seed=1)
    rng = np.random.default_rng(0)
    rhs = rand_rhs(rng, n, 0, m)
    f = kkt.kkt_chol2(G, dims, A)
    got, ref = run_pair(f, ko.KktChol2(G, dims, A), W, None, rhs)        # conelp: factor(W)
    assert relerr(got[0], ref[0]) < SOLVE_RTOL
    Hpoison = np.tril(pr['P']) + np.triu(np.full((n, n), 1e200), 1)       # strict upper must be ignored
    got2, ref2 = run_pair(f, ko.KktChol2(G, dims, A), W, np.asfortranarray(Hpoison), rhs)
    assert relerr(got2[0], ref2[0]) < SOLVE_RTOL and np.all(np.isfinite(got2[0]))
    f.engine.close()


def test_singular_first_call_switches_to_S_plus_AtA():
    # H = 0 and G with a null space => S singular; A restores rank (reference misc.py:1433-1447)
    n, m, p = 40, 30, 15
    rng = np.random.default_rng(2)
    G = np.asfortranarray(rng.standard_normal((m, n)))
    A = np.asfortranarray(rng.standard_normal((p, n)))
    dims = {'l': m, 'q': [], 's': []}
    W = synth.random_scaling(dims, seed=4, spread=0.5)
    f = kkt.kkt_chol2(G, dims, A)
    o = ko.KktChol2(G, dims, A)
    rhs = rand_rhs(rng, n, p, m)
    got, ref = run_pair(f, o, W, None, rhs)
    assert o.singular
    assert f.engine.L.mi355kkt_is_singular_mode(f.engine.h) == 1
    for g, r in zip(got, ref):
        assert relerr(g, r) < 1e-7
    res = ko.kkt_residual(None, A, G, W, dims, rhs[0], rhs[1], rhs[2], *got)
    assert res < 1e-9
    f.engine.close()


def test_not_positive_definite_raises_arithmetic_error():
    n, m = 50, 20
    rng = np.random.default_rng(9)
    G = np.asfortranarray(rng.standard_normal((m, n)))          # rank 20 < 50, no A, H = 0
    dims = {'l': m, 'q': [], 's': []}
    f = kkt.kkt_chol2(G, dims, np.zeros((0, n)))
    with pytest.raises(ArithmeticError):
        f(synth.random_scaling(dims, seed=0), None)
    o = ko.KktChol2(G, dims, np.zeros((0, n)))
    with pytest.raises(ArithmeticError):
        o.factor(synth.random_scaling(dims, seed=0), None)
    f.engine.close()


@pytest.mark.parametrize("reg", [1e-6, 1e-2])
def test_ldl_kktreg_matches_reference_formulation(reg):
    n, m, p = 120, 260, 11
    pr = synth.dense_qp(n, m, seed=1, p=p)
    G, P, A, dims = pr['G'], pr['P'], pr['A'], pr['dims']
    W = synth.random_scaling(dims, seed=2)
    rng = np.random.default_rng(1)
    rhs = rand_rhs(rng, n, p, m)
    f = kkt.kkt_ldl(G, dims, A, kktreg=reg)
    got, ref = run_pair(f, ko.KktLdl(G, dims, A, kktreg=reg), W, P, rhs)
    for g, r in zip(got, ref):
        assert relerr(g, r) < 1e-9
    f.engine.close()


def test_chol2_rejects_soc_like_reference():
    with pytest.raises(ValueError):
        kkt.kkt_chol2(np.zeros((5, 3), order='F'), {'l': 2, 'q': [3], 's': []}, np.zeros((0, 3)))


def test_vs_real_reference_kktsolver(ref_cvxopt):
    """Same W (taken from a real coneqp run), same rhs: GPU closure vs misc.kkt_chol2 closure."""
    cvx = ref_cvxopt
    from cvxopt import matrix, misc
    n, m = 200, 450
    pr = synth.dense_qp(n, m, seed=8)
    P, G = matrix(pr['P']), matrix(pr['G'])
    A = cvx.spmatrix([], [], [], (0, n))
    dims = pr['dims']
    Wnp = synth.random_scaling(dims, seed=3, spread=3.0)
    W = {'d': matrix(Wnp['d']), 'di': matrix(Wnp['di']), 'v': [], 'beta': [], 'r': [], 'rti': []}
    rng = np.random.default_rng(5)
    bx, bz = rng.standard_normal(n), rng.standard_normal(m)
    xr, yr, zr = matrix(bx), matrix(0.0, (0, 1)), matrix(bz)
    misc.kkt_chol2(G, dims, A)(W, P)(xr, yr, zr)
    xg, yg, zg = matrix(bx), matrix(0.0, (0, 1)), matrix(bz)
    fg = kkt.kkt_chol2(G, dims, A)
    fg(W, P)(xg, yg, zg)
    assert relerr(np.array(xg).ravel(), np.array(xr).ravel()) < 1e-8
    assert relerr(np.array(zg).ravel(), np.array(zr).ravel()) < 1e-8
    fg.engine.close()


@pytest.mark.parametrize("n,m,p", [(256, 512, 0), (120, 300, 20)])
def test_coneqp_drop_in_same_iterates(ref_cvxopt, n, m, p):
    """BASELINE config 1: solvers.coneqp with the GPU kktsolver vs kktsolver='chol2' on the CPU:
    same status, same iteration count, objectives within 1e-9 relative, x within 1e-7."""
    cvx = ref_cvxopt
    from cvxopt import matrix, solvers
    pr = synth.dense_qp(n, m, seed=0, p=p)
    P, q, G, h = matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h'])
    kw = {}
    if p:
        kw = dict(A=matrix(pr['A']), b=matrix(pr['b']))
    ref = solvers.coneqp(P, q, G, h, kktsolver='chol2', **kw)
    A = kw.get('A', cvx.spmatrix([], [], [], (0, n)))
    ks = kkt.kktsolver_qp(G, pr['dims'], A, P)
    got = solvers.coneqp(P, q, G, h, kktsolver=ks, **kw)
    assert got['status'] == ref['status'] == 'optimal'
    assert got['iterations'] == ref['iterations']
    for key in ('primal objective', 'dual objective'):
        assert abs(got[key] - ref[key]) <= 1e-9 * max(1.0, abs(ref[key]))
    dx = np.max(np.abs(np.array(got['x']) - np.array(ref['x'])))
    assert dx <= 1e-7 * max(1.0, np.max(np.abs(np.array(ref['x']))))
    if n == 256 and p == 0:
        assert abs(ref['primal objective'] - 5.032917338763e+01) < 1e-9     # SURVEY.md 8(c) probe value
    ks.engine.close()


def test_install_routes_string_names_to_gpu(ref_cvxopt):
    from cvxopt import matrix, solvers, misc
    import cvxopt_amd
    pr = synth.dense_qp(64, 160, seed=2)
    P, q, G, h = matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h'])
    ref = solvers.coneqp(P, q, G, h, kktsolver='chol2')
    cvxopt_amd.install(misc)
    try:
        assert misc.kkt_chol2 is cvxopt_amd.kkt_chol2
        got = solvers.coneqp(P, q, G, h, kktsolver='chol2')
        got_ldl = solvers.coneqp(P, q, G, h, kktsolver='ldl')
    finally:
        cvxopt_amd.uninstall()
    assert misc.kkt_chol2 is not cvxopt_amd.kkt_chol2
    for g in (got, got_ldl):
        assert g['iterations'] == ref['iterations']
        assert abs(g['primal objective'] - ref['primal objective']) <= 1e-9 * abs(ref['primal objective'])

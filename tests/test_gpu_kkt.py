"""Hook-level parity: cvxopt_amd factories (HIP path through the C ABI) vs the oracle
(oracle/kkt_oracle.py, pinned to the reference) and vs the real reference kktsolvers / coneqp."""
import numpy as np
import pytest

from cvxopt_amd import kkt, synth
from helpers import relerr as _relerr  # noqa: F401
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


@pytest.mark.parametrize("flavour", ["ldl", "ldl2", "chol"])
def test_singular_S_at_a_later_call_ldl_and_chol_flavours(flavour):
    # First factorisation: H = I, S positive definite.  Second: H = 0 and G with a null space, so S is singular while the
    # KKT matrix is not (A restores the rank).  The reference's kkt_ldl / kkt_ldl2 (pivoted LDL', lapack.c:2282) and kkt_chol
    # (QR elimination of A, misc.py:1250-1282) solve both; only kkt_chol2 restricts the S + A'A switch to its first call.
    n, m, p = 40, 30, 15
    rng = np.random.default_rng(5)
    G = np.asfortranarray(rng.standard_normal((m, n)))
    A = np.asfortranarray(rng.standard_normal((p, n)))
    dims = {'l': m, 'q': [], 's': []}
    fac = {"ldl": kkt.kkt_ldl, "ldl2": kkt.kkt_ldl2, "chol": kkt.kkt_chol}[flavour]
    f = fac(G, dims, A)
    o = {"ldl": ko.KktLdl, "ldl2": ko.KktLdl, "chol": ko.KktChol}[flavour](G, dims, A)
    for H, seed in ((np.asfortranarray(np.eye(n)), 4), (None, 6)):
        W = synth.random_scaling(dims, seed=seed, spread=0.5)
        rhs = rand_rhs(rng, n, p, m)
        got, ref = run_pair(f, o, W, H, rhs)
        for g, r in zip(got, ref):
            assert relerr(g, r) < 1e-7
        assert ko.kkt_residual(H, A, G, W, dims, rhs[0], rhs[1], rhs[2], *got) < 1e-9
    assert f.engine.L.mi355kkt_is_singular_mode(f.engine.h) == 1
    f.engine.close()
    # kkt_chol2: the same sequence raises at the second call, like misc.py:1440-1447 (the switch is first-call only)
    f2 = kkt.kkt_chol2(G, dims, A)
    f2(synth.random_scaling(dims, seed=4, spread=0.5), np.asfortranarray(np.eye(n)))
    with pytest.raises(ArithmeticError):
        f2(synth.random_scaling(dims, seed=6, spread=0.5), None)
    f2.engine.close()


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


# ---- second-order cones (BASELINE config 3) ------------------------------------------------------------
from helpers import load_golden, dims_of, w_of   # noqa: E402


@pytest.mark.parametrize("case", ["scale1", "scale3"])
def test_cone_scale_op_matches_reference_golden(capi, case):
    import ctypes as C
    rec = load_golden(case)
    dims = dims_of(rec)
    W = w_of(rec, dims)
    x = np.asfortranarray(rec['x'])
    dX = capi.DeviceBuffer.from_array(x)
    q = (C.c_int * len(dims['q']))(*dims['q'])
    ddi = capi.DeviceBuffer.from_array(W['di'] if dims['l'] else np.zeros(1))
    dv = capi.DeviceBuffer.from_array(np.concatenate(W['v']))
    db = capi.DeviceBuffer.from_array(np.array(W['beta']))
    capi.check(capi.lib().mi355kkt_op_cone_scale(dims['l'], len(dims['q']), q, dX.ptr, x.shape[0], x.shape[1],
                                                ddi.ptr, dv.ptr, db.ptr, None), "op_cone_scale")
    got = dX.to_array(x.shape)
    assert relerr(got, rec['out_TI']) < 1e-13


@pytest.mark.parametrize("case,kinds", [("kkt_soc", ["chol", "ldl", "ldl2"]), ("kkt_soc_many", ["chol", "ldl"]),
                                        ("kkt_lp_p5", ["chol2", "chol", "ldl", "ldl2"])])
def test_hook_matches_reference_golden(case, kinds):
    rec = load_golden(case)
    dims = dims_of(rec)
    W = w_of(rec, dims)
    G, A, H = np.asfortranarray(rec['G']), np.asfortranarray(rec['A']), np.asfortranarray(rec['H'])
    for kind in kinds:
        f = getattr(kkt, 'kkt_' + kind)(G, dims, A)
        x, y, z = rec['bx'].copy(), rec['by'].copy(), rec['bz'].copy()
        f(W, H)(x, y, z)
        assert relerr(x, rec['x_' + kind]) < 1e-9 and relerr(y, rec['y_' + kind]) < 1e-9
        assert relerr(z, rec['z_' + kind]) < 1e-9
        f.engine.close()
    f = kkt.kkt_ldl(G, dims, A, kktreg=1e-3)
    x, y, z = rec['bx'].copy(), rec['by'].copy(), rec['bz'].copy()
    f(W, H)(x, y, z)
    assert relerr(x, rec['x_ldlreg']) < 1e-9 and relerr(z, rec['z_ldlreg']) < 1e-9
    f.engine.close()


@pytest.mark.parametrize("n,ml,q,p", [(40, 0, [8] * 6, 0), (64, 10, [3, 40, 5, 100], 4), (200, 0, [4] * 100, 0),
                                      (128, 7, [1, 2, 33], 3)])
def test_soc_factor_solve_matches_oracle(n, ml, q, p):
    dims = {'l': ml, 'q': q, 's': []}
    m = ml + sum(q)
    rng = np.random.default_rng(n + m)
    G = np.asfortranarray(rng.standard_normal((m, n)))
    A = np.asfortranarray(rng.standard_normal((p, n)))
    B = rng.standard_normal((n, n))
    H = np.asfortranarray(B @ B.T / n + 0.05 * np.eye(n))
    f = kkt.kkt_chol(G, dims, A)
    for it in range(2):
        W = synth.random_scaling(dims, seed=it, spread=1.0)
        rhs = rand_rhs(rng, n, p, m)
        x, y, z = (u.copy() for u in rhs)
        f(W, H)(x, y, z)
        xo, yo, zo = (u.copy() for u in rhs)
        ko.KktLdl(G, dims, A).factor(W, H)(xo, yo, zo)
        for g, r in zip((x, y, z), (xo, yo, zo)):
            assert relerr(g, r) < 1e-8, relerr(g, r)
        res = ko.kkt_residual(H, A, G, W, dims, rhs[0], rhs[1], rhs[2], x, y, z)
        # accuracy bar = the reference's own reduced (normal-equations) formulation, misc.kkt_chol
        xc, yc, zc = (u.copy() for u in rhs)
        ko.KktChol(G, dims, A).factor(W, H)(xc, yc, zc)
        res_ref = ko.kkt_residual(H, A, G, W, dims, rhs[0], rhs[1], rhs[2], xc, yc, zc)
        assert res < max(RESID_TOL, 10.0 * res_ref), (res, res_ref)
    f.engine.close()


@pytest.mark.parametrize("name", ["conelp_socp_small", "conelp_socp_mid"])
def test_conelp_socp_drop_in(ref_cvxopt, name):
    """BASELINE config 3 (scaled down): solvers.conelp on a SOCP with the GPU kktsolver vs the golden
    reference run (kktsolver='chol' on the CPU): same status / iterations / objectives."""
    from cvxopt import matrix, solvers
    g = load_golden(name)
    pr = synth.socp(int(g['n']), int(g['N']), int(g['r']), seed=int(g['seed']), ml=int(g['ml']))
    c, G, h = matrix(pr['c']), matrix(pr['G']), matrix(pr['h'])
    A = ref_cvxopt.spmatrix([], [], [], (0, int(g['n'])))
    for kind in ("chol", "qr"):
        # 'qr' has no kktsolver_lp name of its own: it is the kkt_qr factory (misc.py:1570) behind the same call shape
        if kind == "qr":
            fac = kkt.kkt_qr(G, pr['dims'], A)
            ks = lambda W, fac=fac: fac(W)
            ks.engine = fac.engine
        else:
            ks = kkt.kktsolver_lp(G, pr['dims'], A, kind=kind)
        sol = solvers.conelp(c, G, h, pr['dims'], kktsolver=ks)
        assert sol['status'] == 'optimal' and int(g['status_optimal']) == 1
        assert sol['iterations'] == int(g['iterations'])
        assert abs(sol['primal objective'] - float(g['pobj'])) <= 1e-8 * max(1.0, abs(float(g['pobj'])))
        assert relerr(np.array(sol['x']).ravel(), g['x']) < 1e-6
        ks.engine.close()


# ---- edge cases of the boundary ----------------------------------------------------------------------
def test_no_inequalities_and_tiny_problems():
    """cdim = 0 (G is 0 x n): the KKT system is [H A'; A 0]; also n = 1."""
    rng = np.random.default_rng(1)
    for n, p in [(6, 2), (1, 0), (130, 5)]:
        B = rng.standard_normal((n, n))
        H = np.asfortranarray(B @ B.T + np.eye(n))
        A = np.asfortranarray(rng.standard_normal((p, n)))
        G = np.zeros((0, n), order='F')
        dims = {'l': 0, 'q': [], 's': []}
        W = {'d': np.zeros(0), 'di': np.zeros(0), 'v': [], 'beta': [], 'r': [], 'rti': []}
        f = kkt.kkt_chol2(G, dims, A)
        bx, by = rng.standard_normal(n), rng.standard_normal(p)
        x, y, z = bx.copy(), by.copy(), np.zeros(0)
        f(W, H)(x, y, z)
        K = np.block([[H, A.T], [A, np.zeros((p, p))]])
        ref = np.linalg.solve(K, np.concatenate([bx, by]))
        assert relerr(x, ref[:n]) < 1e-10 and relerr(y, ref[n:]) < 1e-9
        f.engine.close()


def test_non_contiguous_and_wrong_dtype_inputs_are_rejected_or_copied():
    n, m = 20, 30
    pr = synth.dense_qp(n, m, seed=0)
    Gc = np.ascontiguousarray(pr['G'])                   # C-ordered: must be accepted (copied to column-major)
    f = kkt.kkt_chol2(Gc, pr['dims'], np.zeros((0, n)))
    W = synth.random_scaling(pr['dims'], seed=0)
    x, y, z = np.ones(n), np.zeros(0), np.ones(m)
    f(W, pr['P'])(x, y, z)
    xo, yo, zo = np.ones(n), np.zeros(0), np.ones(m)
    ko.KktChol2(pr['G'], pr['dims'], np.zeros((0, n))).factor(W, pr['P'])(xo, yo, zo)
    assert relerr(x, xo) < 1e-9
    with pytest.raises(TypeError):
        f(W, pr['P'])(np.ones(n, dtype=np.float32), y, z)
    with pytest.raises(ValueError):
        f(W, pr['P'])(np.ones(n + 1), y, z)
    with pytest.raises(TypeError):
        kkt.kkt_chol2(pr['G'][:, :-1], pr['dims'], np.zeros((0, n)))
    f.engine.close()


def test_w_is_reread_on_every_factor_call():
    """misc.update_scaling mutates W in place (misc.py:450-464): the hook must never cache W by identity."""
    n, m = 40, 90
    pr = synth.dense_qp(n, m, seed=3)
    W = synth.random_scaling(pr['dims'], seed=1)
    f = kkt.kkt_chol2(pr['G'], pr['dims'], np.zeros((0, n)))
    rng = np.random.default_rng(0)
    for it in range(3):
        W['d'] *= rng.uniform(0.5, 2.0, m)               # same dict, same arrays, new values
        W['di'][:] = 1.0 / W['d']
        bx, bz = rng.standard_normal(n), rng.standard_normal(m)
        x, y, z = bx.copy(), np.zeros(0), bz.copy()
        f(W, pr['P'])(x, y, z)
        xo, yo, zo = bx.copy(), np.zeros(0), bz.copy()
        ko.KktChol2(pr['G'], pr['dims'], np.zeros((0, n))).factor(W, pr['P'])(xo, yo, zo)
        assert relerr(x, xo) < 1e-9 and relerr(z, zo) < 1e-9
    f.engine.close()


@pytest.mark.parametrize("n,m", [(2000, 2500), (127, 300), (129, 10), (3333, 100)])
def test_persistent_triangular_solves_stress(n, m):
    """The single-launch trsv hands x blocks between workgroups (agent-scope release/acquire): hammer it with
    repeated solves on the same factor and check every one against the oracle solution of the same system."""
    rng = np.random.default_rng(n)
    G = np.asfortranarray(rng.standard_normal((m, n)))
    H = np.asfortranarray(np.diag(rng.uniform(1.0, 2.0, n)))
    dims = {'l': m, 'q': [], 's': []}
    A = np.zeros((0, n))
    W = synth.random_scaling(dims, seed=1, spread=0.5)
    f = kkt.kkt_chol2(G, dims, A)
    s = f(W, H)
    o = ko.KktChol2(G, dims, A).factor(W, H)
    worst = 0.0
    for it in range(60):
        bx, bz = rng.standard_normal(n), rng.standard_normal(m)
        x, y, z = bx.copy(), np.zeros(0), bz.copy()
        s(x, y, z)
        xo, yo, zo = bx.copy(), np.zeros(0), bz.copy()
        o(xo, yo, zo)
        worst = max(worst, relerr(x, xo), relerr(z, zo))
    assert worst < 1e-9, worst
    f.engine.close()


def test_multi_kernel_trsm_path_still_matches():
    """the blocked multi-kernel triangular solve (orders beyond the persistent kernel's co-residency limit, multiple right-hand
    sides: Asct = L^-1 A') against NumPy, through the stand-alone operator"""
    import ctypes as C
    from cvxopt_amd import _capi
    rng = np.random.default_rng(0)
    n, nrhs = 700, 3
    Lm = np.tril(rng.standard_normal((n, n))) / np.sqrt(n) + 2.0 * np.eye(n)
    X = rng.standard_normal((n, nrhs))
    dL, ms = _capi.DeviceBuffer.from_array(np.asfortranarray(Lm)), C.c_float(0)
    for trans in (0, 1):
        dX = _capi.DeviceBuffer.from_array(np.asfortranarray(X))
        _capi.check(_capi.lib().mi355kkt_op_trsm_lower(C.c_void_p(dL.ptr), n, n, C.c_void_p(dX.ptr), n, nrhs, trans, C.byref(ms)),
                    "op_trsm_lower")
        got = dX.to_array((n, nrhs))
        want = np.linalg.solve(Lm.T if trans else Lm, X)
        assert relerr(got, want) < 1e-11
        dX.free()
    dL.free()


def test_full_size_config2_coneqp_matches_reference_probe(ref_cvxopt):
    """BASELINE configs[1] at full size (n=8192, m=16384) through the reference driver with the GPU kktsolver.
    Reference values: tests/golden/full_qp8192.npz, the unmodified reference with kktsolver='chol2' on the same seeded
    problem (tests/golden/make_golden_full.py).
    Plus the size-independent property: every KKT solve leaves a small residual (checked on the reduced system
    with a matrix-free product on the host for the last iteration's scaling)."""
    from cvxopt import matrix, solvers, spmatrix
    n, m = 8192, 16384
    pr = synth.dense_qp(n, m, seed=0)
    P, q, G, h = matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h'])
    A = spmatrix([], [], [], (0, n))
    ks = kkt.kktsolver_qp(G, pr['dims'], A, P)
    resid = []

    def wrapped(W):
        f = ks(W)

        def solve(x, y, z):
            bx, bz = np.array(x).ravel().copy(), np.array(z).ravel().copy()
            f(x, y, z)
            if len(resid) < 4:            # a few matrix-free residual checks of S ux = bx + G' di^2 bz (host GEMVs)
                di = np.array(W['di']).ravel()
                ux = np.array(x).ravel()
                lhs = pr['P'] @ ux + pr['G'].T @ (di * di * (pr['G'] @ ux))
                rhs = bx + pr['G'].T @ (di * di * bz)
                resid.append(np.linalg.norm(lhs - rhs) / np.linalg.norm(rhs))
        return solve
    sol = solvers.coneqp(P, q, G, h, kktsolver=wrapped)
    assert sol['status'] == 'optimal'
    g = load_golden("full_qp8192")
    assert sol['iterations'] == int(g['iterations'])
    assert abs(sol['primal objective'] - float(g['pobj'])) <= 1e-9 * abs(float(g['pobj']))
    assert abs(sol['dual objective'] - float(g['dobj'])) <= 1e-9 * abs(float(g['dobj']))
    assert relerr(np.array(sol['x']).ravel(), g['x']) < 1e-6
    assert max(resid) < 1e-11, resid
    ks.engine.close()


# ---- semidefinite ('s') cones ----------------------------------------------------------------------------
def _pack_sym(z, dims):
    return ko.pack(ko._symmetrize_s(z.copy(), dims), dims)


def test_sdp_hook_matches_reference_golden():
    rec = load_golden("kkt_sdp")
    dims = dims_of(rec)
    W = w_of(rec, dims)
    G, A, H = np.asfortranarray(rec['G']), np.asfortranarray(rec['A']), np.asfortranarray(rec['H'])
    for kind in ("chol", "ldl", "ldl2"):
        f = getattr(kkt, 'kkt_' + kind)(G, dims, A)
        x, y, z = rec['bx'].copy(), rec['by'].copy(), rec['bz'].copy()
        f(W, H)(x, y, z)
        assert relerr(x, rec['x_' + kind]) < 1e-9 and relerr(y, rec['y_' + kind]) < 1e-9
        assert relerr(_pack_sym(z, dims), _pack_sym(rec['z_' + kind], dims)) < 1e-9
        f.engine.close()
    f = kkt.kkt_ldl(G, dims, A, kktreg=1e-3)
    x, y, z = rec['bx'].copy(), rec['by'].copy(), rec['bz'].copy()
    f(W, H)(x, y, z)
    assert relerr(x, rec['x_ldlreg']) < 1e-9
    assert relerr(_pack_sym(z, dims), _pack_sym(rec['z_ldlreg'], dims)) < 1e-9
    f.engine.close()


@pytest.mark.parametrize("dims,n,p", [({'l': 0, 'q': [], 's': [5]}, 8, 0), ({'l': 4, 'q': [3, 6], 's': [2, 7, 12]}, 30, 3),
                                      ({'l': 0, 'q': [], 's': [40, 1]}, 100, 0),
                                      ({'l': 3, 'q': [4], 's': [96, 7, 130]}, 24, 2)])     # blocks beyond the LDS-resident kernel
def test_sdp_factor_solve_matches_oracle(dims, n, p):
    m = ko.cdim(dims)
    rng = np.random.default_rng(n + m)
    G = np.asfortranarray(rng.standard_normal((m, n)))
    for c in range(n):
        G[:, c] = ko._symmetrize_s(G[:, c], dims)
    A = np.asfortranarray(rng.standard_normal((p, n)))
    B = rng.standard_normal((n, n))
    H = np.asfortranarray(B @ B.T / n + 0.1 * np.eye(n))
    f = kkt.kkt_chol(G, dims, A)
    for it in range(2):
        W = synth.random_scaling(dims, seed=it, spread=0.7)
        bx, by = rng.standard_normal(n), rng.standard_normal(p)
        bz = ko._symmetrize_s(rng.standard_normal(m), dims)
        x, y, z = bx.copy(), by.copy(), bz.copy()
        f(W, H)(x, y, z)
        xo, yo, zo = bx.copy(), by.copy(), bz.copy()
        ko.KktChol(G, dims, A).factor(W, H)(xo, yo, zo)
        # random r / rti are not well conditioned: forward error bound loose, backward error (KKT residual) tight
        assert relerr(x, xo) < 1e-6 and relerr(y, yo) < 1e-6
        assert relerr(_pack_sym(z, dims), _pack_sym(zo, dims)) < 1e-6
        zs, zos = ko._symmetrize_s(z, dims), ko._symmetrize_s(zo, dims)
        res = ko.kkt_residual(H, A, G, W, dims, bx, by, bz, x, y, zs)
        res_ref = ko.kkt_residual(H, A, G, W, dims, bx, by, bz, xo, yo, zos)
        assert res < max(RESID_TOL, 10.0 * res_ref), (res, res_ref)
    f.engine.close()


def test_sdp_drop_in_matches_reference(ref_cvxopt):
    """solvers.sdp (reference doc example, doc/source/coneprog.rst sdp section / examples/doc/chap8/sdp.py data) with
    the GPU kktsolver routed through install(): same status / iterations / objective as the CPU default."""
    from cvxopt import matrix, solvers, misc
    import cvxopt_amd
    c = matrix([1., -1., 1.])
    G = [matrix([[-7., -11., -11., 3.], [7., -18., -18., 8.], [-2., -8., -8., 1.]])]
    G += [matrix([[-21., -11., 0., -11., 10., 8., 0., 8., 5.], [0., 10., 16., 10., -10., -10., 16., -10., 3.],
                  [-5., 2., -17., 2., -6., 8., -17., 8., 6.]])]
    h = [matrix([[33., -9.], [-9., 26.]]), matrix([[14., 9., 40.], [9., 91., 10.], [40., 10., 15.]])]
    ref = solvers.sdp(c, Gs=G, hs=h, kktsolver='chol')
    cvxopt_amd.install(misc)
    try:
        got = solvers.sdp(c, Gs=G, hs=h, kktsolver='chol')
        got_default = solvers.sdp(c, Gs=G, hs=h)             # default 'qr' -> kkt_qr mirror
    finally:
        cvxopt_amd.uninstall()
    for g in (got, got_default):
        assert g['status'] == ref['status'] == 'optimal'
        assert abs(g['primal objective'] - ref['primal objective']) <= 1e-7 * max(1, abs(ref['primal objective']))
        assert relerr(np.array(g['x']).ravel(), np.array(ref['x']).ravel()) < 1e-6
    assert got['iterations'] == ref['iterations']


def test_large_sdp_block_drop_in_matches_reference(ref_cvxopt):
    """mcsdp-style problem (reference examples/doc/chap8/mcsdp.py: maximise 1'x s.t. w + diag(x) <= 0) with one 100 x 100
    block: beyond the 80 x 80 LDS kernel, through the panel kernel."""
    from cvxopt import matrix, solvers, spmatrix
    n = 100
    rng = np.random.default_rng(0)
    Wm = rng.standard_normal((n, n)); Wm = (Wm + Wm.T) / 2
    c = matrix(-1.0, (n, 1))
    G = spmatrix(1.0, [i * (n + 1) for i in range(n)], list(range(n)), (n * n, n))
    h = matrix(-Wm.reshape(-1, order='F'))
    dims = {'l': 0, 'q': [], 's': [n]}
    Gd = matrix(G)
    ref = solvers.conelp(c, Gd, h, dims)
    ks = kkt.kktsolver_lp(Gd, dims, spmatrix([], [], [], (0, n)))
    got = solvers.conelp(c, Gd, h, dims, kktsolver=ks)
    ks.engine.close()
    assert got['status'] == ref['status'] == 'optimal' and got['iterations'] == ref['iterations']
    assert abs(got['primal objective'] - ref['primal objective']) <= 1e-8 * max(1.0, abs(ref['primal objective']))
    assert relerr(np.array(got['x']).ravel(), np.array(ref['x']).ravel()) < 1e-6

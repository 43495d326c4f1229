import os
"""Device-resident coneqp loop (SURVEY.md 8(f) row 1) for a single LP-cone QP: mi355kkt_coneqp_lp vs the reference
driver (same iterates: status, iteration count, objectives, x/s/z), the committed golden run, and -- at BASELINE
configs[1]'s full size -- the surveyor's probe values of the unmodified reference."""
import time

import numpy as np
import pytest

import cvxopt_amd
from cvxopt_amd import synth
from helpers import load_golden, relerr

pytestmark = pytest.mark.gpu


def test_resident_coneqp_matches_golden_reference_run():
    g = load_golden("coneqp_qp256")
    pr = synth.dense_qp(int(g['n']), int(g['m']), seed=int(g['seed']))
    sol = cvxopt_amd.coneqp_lp(pr['P'], pr['q'], pr['G'], pr['h'])
    assert sol['status'] == 'optimal' and sol['iterations'] == int(g['iterations'])
    assert abs(sol['primal objective'] - float(g['pobj'])) <= 1e-9 * abs(float(g['pobj']))
    assert relerr(sol['x'], g['x']) < 1e-7


@pytest.mark.parametrize("n,m,kind", [(48, 100, 'chol2'), (200, 333, 'chol'), (130, 61, 'ldl'), (300, 700, 'ldl2')])
def test_resident_coneqp_matches_reference_driver(ref_cvxopt, n, m, kind):
    from cvxopt import matrix, solvers
    pr = synth.dense_qp(n, m, seed=n + m)
    ref = solvers.coneqp(matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']), kktsolver='chol2')
    sol = cvxopt_amd.coneqp_lp(matrix(pr['P']), pr['q'], matrix(pr['G']), pr['h'], kktsolver=kind)
    assert sol['status'] == ref['status'] == 'optimal'
    assert sol['iterations'] == ref['iterations']
    for k in ('primal objective', 'dual objective'):
        assert abs(sol[k] - ref[k]) <= 1e-9 * max(1.0, abs(ref[k])), k
    assert abs(sol['gap'] - ref['gap']) <= 1e-6 * ref['gap'] + 1e-14
    assert relerr(sol['x'], np.array(ref['x']).ravel()) < 1e-7
    assert relerr(sol['s'], np.array(ref['s']).ravel()) < 1e-6
    assert relerr(sol['z'], np.array(ref['z']).ravel()) < 1e-6


def test_resident_coneqp_only_reads_tril_of_P_and_handles_lp(ref_cvxopt):
    from cvxopt import matrix, solvers
    pr = synth.dense_qp(60, 140, seed=5)
    Pl = np.tril(pr['P']) + np.triu(np.full((60, 60), 7.0), 1)          # garbage above the diagonal
    a = cvxopt_amd.coneqp_lp(Pl, pr['q'], pr['G'], pr['h'])
    b = cvxopt_amd.coneqp_lp(pr['P'], pr['q'], pr['G'], pr['h'])
    assert a['iterations'] == b['iterations'] and relerr(a['x'], b['x']) < 1e-12
    # P = None: an LP in coneqp form, bounded because G has a box part
    rng = np.random.default_rng(3)
    n = 30
    G = np.vstack([np.eye(n), -np.eye(n), rng.standard_normal((20, n))])
    h = np.concatenate([np.ones(2 * n), 5.0 + rng.random(20)])
    q = rng.standard_normal(n)
    sol = cvxopt_amd.coneqp_lp(None, q, G, h)
    ref = solvers.coneqp(matrix(np.zeros((n, n))), matrix(q), matrix(G), matrix(h), kktsolver='chol2')
    assert sol['status'] == ref['status'] and sol['iterations'] == ref['iterations']
    assert abs(sol['primal objective'] - ref['primal objective']) <= 1e-8 * max(1, abs(ref['primal objective']))


def test_resident_coneqp_errors():
    pr = synth.dense_qp(40, 20, seed=1)
    with pytest.raises(ValueError):                                     # Rank([P; G]) < n
        cvxopt_amd.coneqp_lp(None, pr['q'], pr['G'], pr['h'])
    pr = synth.dense_qp(32, 64, seed=2)
    sol = cvxopt_amd.coneqp_lp(pr['P'], pr['q'], pr['G'], pr['h'], maxiters=2)
    assert sol['status'] == 'unknown' and sol['iterations'] == 2


def test_resident_coneqp_full_size_config2():
    """n = 8192, m = 16384 against tests/golden/full_qp8192.npz (the unmodified reference, tests/golden/make_golden_full.py)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_qp8192.npz"))
    n, m = 8192, 16384
    pr = synth.dense_qp(n, m, seed=0)
    t = time.perf_counter()
    sol = cvxopt_amd.coneqp_lp(pr['P'], pr['q'], pr['G'], pr['h'])
    t = time.perf_counter() - t
    print("resident coneqp n=8192: %.2f s wall incl. upload, %d iterations" % (t, sol['iterations']))
    assert sol['status'] == 'optimal' and sol['iterations'] == int(g['iterations'])
    assert abs(sol['primal objective'] - float(g['pobj'])) <= 1e-9 * abs(float(g['pobj']))
    assert abs(sol['dual objective'] - float(g['dobj'])) <= 1e-9 * abs(float(g['dobj']))
    assert relerr(sol['x'], g['x']) < 1e-6
    # size-independent properties: primal/dual feasibility and complementarity of the returned point
    x, s, z = sol['x'], sol['s'], sol['z']
    assert np.all(s > 0) and np.all(z > 0)
    assert np.linalg.norm(pr['G'] @ x + s - pr['h']) <= 1e-7 * max(1.0, np.linalg.norm(pr['h']))
    assert np.linalg.norm(pr['P'] @ x + pr['q'] + pr['G'].T @ z) <= 1e-7 * max(1.0, np.linalg.norm(pr['q']))
    assert abs(s @ z - sol['gap']) <= 1e-9 * max(1.0, sol['gap'])


@pytest.mark.parametrize("n,m,p,kind", [(64, 150, 9, 'chol2'), (200, 310, 40, 'chol'), (120, 100, 33, 'ldl')])
def test_resident_coneqp_with_equality_constraints(ref_cvxopt, n, m, p, kind):
    """A x = b in the device-resident loop: y, ry, dy bookkeeping of coneprog.py:2170-2190, :2459-2463."""
    from cvxopt import matrix, solvers
    pr = synth.dense_qp(n, m, seed=7 + n, p=p)
    ref = solvers.coneqp(matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']), A=matrix(pr['A']),
                         b=matrix(pr['b']), kktsolver='chol2')
    sol = cvxopt_amd.coneqp_lp(pr['P'], pr['q'], pr['G'], pr['h'], A=pr['A'], b=pr['b'], kktsolver=kind)
    assert sol['status'] == ref['status'] == 'optimal'
    assert sol['iterations'] == ref['iterations']
    for k in ('primal objective', 'dual objective'):
        assert abs(sol[k] - ref[k]) <= 1e-9 * max(1.0, abs(ref[k])), k
    assert relerr(sol['x'], np.array(ref['x']).ravel()) < 1e-7
    assert relerr(sol['y'], np.array(ref['y']).ravel()) < 1e-6
    assert relerr(sol['z'], np.array(ref['z']).ravel()) < 1e-6
    assert np.linalg.norm(pr['A'] @ sol['x'] - pr['b']) <= 1e-8 * max(1.0, np.linalg.norm(pr['b']))


def test_resident_coneqp_equalities_singular_P_uses_S_plus_AtA(ref_cvxopt):
    """P + G'G singular on the first call (few inequalities, P = 0 on half of the space; the objective is still
    strictly convex on {A x = b}): the engine's S += A'A fallback
    (misc.py:1433-1447) inside the resident loop."""
    from cvxopt import matrix, solvers
    rng = np.random.default_rng(11)
    n, m, p = 30, 12, 20
    G = rng.standard_normal((m, n))
    A = rng.standard_normal((p, n))
    x0 = rng.standard_normal(n)
    h, b = G @ x0 + 0.5 + rng.random(m), A @ x0
    q = rng.standard_normal(n)
    P = np.diag(np.concatenate([np.ones(15), np.zeros(15)]))
    ref = solvers.coneqp(matrix(P), matrix(q), matrix(G), matrix(h), A=matrix(A), b=matrix(b), kktsolver='chol2')
    sol = cvxopt_amd.coneqp_lp(P, q, G, h, A=A, b=b)
    assert sol['status'] == ref['status'] == 'optimal' and sol['iterations'] == ref['iterations']
    assert abs(sol['primal objective'] - ref['primal objective']) <= 1e-8 * max(1.0, abs(ref['primal objective']))
    assert relerr(sol['x'], np.array(ref['x']).ravel()) < 1e-6


# ---- conelp (self-dual loop) resident on the device, LP cone ------------------------------------------------
def _lp(n, m, p=0, seed=0):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((m, n))
    x0 = rng.standard_normal(n)
    h = G @ x0 + rng.uniform(0.1, 1.0, m)
    z0 = rng.uniform(0.1, 1.0, m)
    A = rng.standard_normal((p, n))
    b = A @ x0
    c = -G.T @ z0 - A.T @ rng.standard_normal(p)          # dual feasible => bounded
    return c, G, h, A, b


@pytest.mark.parametrize("n,m,p,kind", [(40, 100, 0, 'chol'), (150, 400, 0, 'chol2'), (80, 200, 20, 'chol'),
                                        (60, 61, 30, 'ldl'), (300, 900, 0, 'qr')])
def test_resident_conelp_matches_reference_driver(ref_cvxopt, n, m, p, kind):
    from cvxopt import matrix, solvers
    c, G, h, A, b = _lp(n, m, p, seed=n + m + p)
    kw = dict(A=matrix(A), b=matrix(b)) if p else {}
    ref = solvers.conelp(matrix(c), matrix(G), matrix(h), **kw)
    sol = cvxopt_amd.conelp_lp(c, G, h, A=A if p else None, b=b if p else None, kktsolver=kind)
    assert sol['status'] == ref['status'] == 'optimal'
    assert sol['iterations'] == ref['iterations']
    for k in ('primal objective', 'dual objective'):
        assert abs(sol[k] - ref[k]) <= 1e-9 * max(1.0, abs(ref[k])), k
    assert abs(sol['gap'] - ref['gap']) <= 1e-4 * ref['gap'] + 1e-14      # the final gap is itself O(1e-6): rounding shows
    assert relerr(sol['x'], np.array(ref['x']).ravel()) < 1e-7
    assert relerr(sol['s'], np.array(ref['s']).ravel()) < 1e-6
    assert relerr(sol['z'], np.array(ref['z']).ravel()) < 1e-6
    if p:
        assert relerr(sol['y'], np.array(ref['y']).ravel()) < 1e-6
    for k in ('primal infeasibility', 'dual infeasibility'):
        assert abs(sol[k] - ref[k]) <= 1e-4 * abs(ref[k]) + 1e-12, k
    assert abs(sol['primal slack'] - ref['primal slack']) <= 1e-6 * abs(ref['primal slack']) + 1e-12


def test_resident_conelp_reference_doc_example(ref_cvxopt):
    """reference tests/test_examples.py:31-34 / doc/source/coneprog.rst LP: x = [1, 1]."""
    from cvxopt import matrix, solvers
    c = np.array([-4.0, -5.0])
    G = np.array([[2.0, 1.0], [1.0, 2.0], [-1.0, 0.0], [0.0, -1.0]])
    h = np.array([3.0, 3.0, 0.0, 0.0])
    sol = cvxopt_amd.conelp_lp(c, G, h)
    ref = solvers.lp(matrix(c), matrix(G), matrix(h))
    assert sol['status'] == 'optimal' and sol['iterations'] == ref['iterations']
    assert np.allclose(sol['x'], [1.0, 1.0], atol=1e-6)
    assert relerr(sol['x'], np.array(ref['x']).ravel()) < 1e-8


def test_resident_conelp_infeasibility_certificates(ref_cvxopt):
    from cvxopt import matrix, solvers
    rng = np.random.default_rng(4)
    n, m = 12, 30
    # primal infeasible: x_0 <= -1 and -x_0 <= -1 among random rows
    G = rng.standard_normal((m, n))
    G[0, :] = 0.0; G[0, 0] = 1.0
    G[1, :] = 0.0; G[1, 0] = -1.0
    h = rng.uniform(0.5, 1.0, m); h[0] = -1.0; h[1] = -1.0
    c = rng.standard_normal(n)
    ref = solvers.conelp(matrix(c), matrix(G), matrix(h))
    sol = cvxopt_amd.conelp_lp(c, G, h)
    assert sol['status'] == ref['status'] == 'primal infeasible' and sol['iterations'] == ref['iterations']
    assert sol['x'] is None and sol['s'] is None
    assert relerr(sol['z'], np.array(ref['z']).ravel()) < 1e-6
    assert abs(sol['residual as primal infeasibility certificate'] - ref['residual as primal infeasibility certificate']) \
        <= 1e-6 * ref['residual as primal infeasibility certificate'] + 1e-14
    # dual infeasible (unbounded): minimise -x_0 with only x_0 >= 0 and box constraints on the other variables
    G2 = np.vstack([-np.eye(n), np.eye(n)[1:]])
    h2 = np.concatenate([np.zeros(n), np.ones(n - 1)])
    c2 = np.zeros(n); c2[0] = -1.0
    ref = solvers.conelp(matrix(c2), matrix(G2), matrix(h2))
    sol = cvxopt_amd.conelp_lp(c2, G2, h2)
    assert sol['status'] == ref['status'] == 'dual infeasible' and sol['iterations'] == ref['iterations']
    assert sol['z'] is None and relerr(sol['x'], np.array(ref['x']).ravel()) < 1e-6


def test_resident_conelp_limits_and_errors():
    c, G, h, A, b = _lp(30, 70, 0, seed=1)
    sol = cvxopt_amd.conelp_lp(c, G, h, maxiters=2)
    assert sol['status'] == 'unknown' and sol['iterations'] == 2
    with pytest.raises(ValueError):                        # p + m < n: Rank([G; A]) < n
        cvxopt_amd.conelp_lp(np.ones(10), np.ones((4, 10)), np.ones(4))
    with pytest.raises(ValueError):                        # rank deficient G at the start
        Gd = np.ones((20, 10))
        cvxopt_amd.conelp_lp(np.ones(10), Gd, np.ones(20))


def test_resident_conelp_large_lp_optimality_conditions():
    """n = 4096, m = 12288 (no CPU reference at this size in the suite): size-independent properties of the returned
    point -- primal / dual feasibility, complementarity, zero duality gap."""
    n, m = 4096, 12288
    c, G, h, A, b = _lp(n, m, 0, seed=3)
    t = time.perf_counter()
    sol = cvxopt_amd.conelp_lp(c, G, h)
    t = time.perf_counter() - t
    print("resident conelp n=%d m=%d: %.2f s wall incl. upload, %d iterations" % (n, m, t, sol['iterations']))
    assert sol['status'] == 'optimal'
    x, s, z = sol['x'], sol['s'], sol['z']
    assert np.all(s > 0) and np.all(z > 0)
    assert np.linalg.norm(G @ x + s - h) <= 1e-7 * max(1.0, np.linalg.norm(h))
    assert np.linalg.norm(G.T @ z + c) <= 1e-7 * max(1.0, np.linalg.norm(c))
    assert abs(s @ z - sol['gap']) <= 1e-6 * max(1e-6, sol['gap'])
    assert abs(c @ x - sol['primal objective']) <= 1e-9 * max(1.0, abs(c @ x))
    assert abs(-(h @ z) - sol['dual objective']) <= 1e-9 * max(1.0, abs(h @ z))
    assert abs(sol['primal objective'] - sol['dual objective']) <= 1e-6 * max(1.0, abs(sol['primal objective']))


# ---- conelp resident on the device with second-order cones (refinement 1, hyperbolic Householder scalings) ------
@pytest.mark.parametrize("n,ncones,r,ml,p", [(24, 6, 5, 0, 0), (60, 20, 4, 15, 0), (96, 48, 8, 10, 7), (50, 3, 30, 0, 0)])
def test_resident_conelp_socp_matches_reference_driver(ref_cvxopt, n, ncones, r, ml, p):
    from cvxopt import matrix, solvers
    pr = synth.socp(n=n, ncones=ncones, r=r, seed=n + ncones, ml=ml)
    rng = np.random.default_rng(p + 1)
    kw, kwd = {}, {}
    if p:
        A = rng.standard_normal((p, n))
        xf = np.linalg.lstsq(pr['G'], pr['h'], rcond=None)[0] * 0.0
        b = A @ xf                                  # x = 0 direction: h = G x0 + s0 with interior s0, keep it simple
        kw, kwd = dict(A=matrix(A), b=matrix(b)), dict(A=A, b=b)
    ref = solvers.conelp(matrix(pr['c']), matrix(pr['G']), matrix(pr['h']), pr['dims'], **kw)
    sol = cvxopt_amd.conelp_device(pr['c'], pr['G'], pr['h'], pr['dims'], **kwd)
    assert sol['status'] == ref['status']
    assert sol['iterations'] == ref['iterations']
    if ref['status'] == 'optimal':
        for k in ('primal objective', 'dual objective'):
            assert abs(sol[k] - ref[k]) <= 1e-8 * max(1.0, abs(ref[k])), k
        assert relerr(sol['x'], np.array(ref['x']).ravel()) < 1e-6
        assert relerr(sol['s'], np.array(ref['s']).ravel()) < 1e-5
        assert relerr(sol['z'], np.array(ref['z']).ravel()) < 1e-5
        assert abs(sol['primal slack'] - ref['primal slack']) <= 1e-4 * abs(ref['primal slack']) + 1e-9


def test_resident_conelp_socp_config3_full_size(ref_cvxopt):
    """BASELINE configs[2]: n=2048, 1024 cones of dimension 8; vs the reference driver with the GPU kktsolver + device
    operators (same iterates as the CPU reference, tests/test_gpu_solvers.py)."""
    from cvxopt import matrix
    import cvxopt_amd.solvers as gs
    pr = synth.socp(n=2048, ncones=1024, r=8, seed=0)
    t = time.perf_counter()
    sol = cvxopt_amd.conelp_device(pr['c'], pr['G'], pr['h'], pr['dims'])
    t = time.perf_counter() - t
    print("resident conelp, config 3: %.3f s wall incl. upload, %d iterations" % (t, sol['iterations']))
    # the CPU reference's run of the same problem (tests/golden/full_socp2048.npz), then the hook-level GPU run as well
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_socp2048.npz"))
    assert sol['status'] == 'optimal' and sol['iterations'] == int(g['iterations'])
    assert abs(sol['primal objective'] - float(g['pobj'])) <= 1e-8 * max(1.0, abs(float(g['pobj'])))
    assert relerr(sol['x'], g['x']) < 1e-6
    ref = gs.conelp(matrix(pr['c']), matrix(pr['G']), matrix(pr['h']), pr['dims'], device_loop=False)
    assert ref['status'] == 'optimal' and ref['iterations'] == int(g['iterations'])
    assert abs(ref['primal objective'] - float(g['pobj'])) <= 1e-8 * max(1.0, abs(float(g['pobj'])))
    assert relerr(np.array(ref['x']).ravel(), g['x']) < 1e-6


# ---- coneqp resident on the device with second-order cones -------------------------------------------------------
@pytest.mark.parametrize("n,ncones,r,ml,p", [(30, 6, 5, 0, 0), (64, 16, 4, 20, 0), (80, 10, 9, 12, 6), (40, 0, 0, 90, 5)])
def test_resident_coneqp_with_cones_matches_reference_driver(ref_cvxopt, n, ncones, r, ml, p):
    from cvxopt import matrix, solvers
    rng = np.random.default_rng(n + ncones)
    if ncones:
        pr = synth.socp(n=n, ncones=ncones, r=r, seed=n, ml=ml)
        G, h, dims = pr['G'], pr['h'], pr['dims']
    else:
        pr = synth.dense_qp(n, ml, seed=n)
        G, h, dims = pr['G'], pr['h'], pr['dims']
    B = rng.standard_normal((n, n)) / np.sqrt(n)
    P = B.T @ B + 1e-2 * np.eye(n)
    q = rng.standard_normal(n)
    kw, kwd = {}, {}
    if p:
        A = rng.standard_normal((p, n))
        b = np.zeros(p)                               # x = 0 satisfies A x = b; the cone part is strictly feasible nearby
        kw, kwd = dict(A=matrix(A), b=matrix(b)), dict(A=A, b=b)
    ref = solvers.coneqp(matrix(P), matrix(q), matrix(G), matrix(h), dims, **kw)
    sol = cvxopt_amd.coneqp_device(P, q, G, h, dims, **kwd)
    assert sol['status'] == ref['status']
    assert sol['iterations'] == ref['iterations']
    if ref['status'] == 'optimal':
        for k in ('primal objective', 'dual objective'):
            assert abs(sol[k] - ref[k]) <= 1e-8 * max(1.0, abs(ref[k])), k
        assert relerr(sol['x'], np.array(ref['x']).ravel()) < 1e-6
        assert relerr(sol['s'], np.array(ref['s']).ravel()) < 1e-5
        assert relerr(sol['z'], np.array(ref['z']).ravel()) < 1e-5
        if p:
            assert relerr(sol['y'], np.array(ref['y']).ravel()) < 1e-5

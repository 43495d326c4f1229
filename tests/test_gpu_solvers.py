"""cvxopt_amd.solvers: the unmodified reference drivers fed with device operators (G, A, P as callables around
mi355kkt_product) and the GPU kktsolver -- same iterates as the reference on its own data path, for every cone type."""
import time

import numpy as np
import pytest

from cvxopt_amd import synth
from helpers import relerr

pytestmark = pytest.mark.gpu


def _same(sol, ref, xt=1e-7):
    assert sol['status'] == ref['status']
    assert sol['iterations'] == ref['iterations']
    for k in ('primal objective', 'dual objective'):
        assert abs(sol[k] - ref[k]) <= 1e-9 * max(1.0, abs(ref[k])), k
    assert relerr(np.array(sol['x']).ravel(), np.array(ref['x']).ravel()) < xt


def test_products_match_numpy():
    from cvxopt_amd import kkt
    rng = np.random.default_rng(0)
    n, m, p = 50, 120, 7
    G, A = np.asfortranarray(rng.standard_normal((m, n))), np.asfortranarray(rng.standard_normal((p, n)))
    P = rng.standard_normal((n, n)); P = P @ P.T
    f = kkt.kkt_chol2(G, {'l': m, 'q': [], 's': []}, A)
    e = f.engine
    e._set_H(np.asfortranarray(np.tril(P) + np.triu(np.full((n, n), 3.0), 1)))      # garbage above the diagonal
    x, z, y = rng.standard_normal(n), rng.standard_normal(m), rng.standard_normal(p)
    assert relerr(e.product(0, False, x), G @ x) < 1e-13 and relerr(e.product(0, True, z), G.T @ z) < 1e-13
    assert relerr(e.product(1, False, x), A @ x) < 1e-13 and relerr(e.product(1, True, y), A.T @ y) < 1e-13
    assert relerr(e.product(2, False, x), P @ x) < 1e-13
    e.close()


def test_default_dispatch_uses_device_loop_and_returns_reference_types(ref_cvxopt):
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    pr = synth.socp(n=40, ncones=10, r=5, seed=4, ml=6)
    c, G, h, dims = matrix(pr['c']), matrix(pr['G']), matrix(pr['h']), pr['dims']
    ref = solvers.conelp(c, G, h, dims)
    dev = gs.conelp(c, G, h, dims)                       # 'l' + 'q' cones, default start: whole loop on the device
    host = gs.conelp(c, G, h, dims, device_loop=False)   # reference driver + device operators
    for sol in (dev, host):
        _same(sol, ref, xt=1e-6)
        assert isinstance(sol['x'], type(ref['x'])) and sol['x'].size == ref['x'].size
        assert abs(sol['primal slack'] - ref['primal slack']) <= 1e-5 * abs(ref['primal slack']) + 1e-9
    qp = synth.dense_qp(30, 70, seed=1, p=3)
    P, q, Gq, hq = matrix(qp['P']), matrix(qp['q']), matrix(qp['G']), matrix(qp['h'])
    A, b = matrix(qp['A']), matrix(qp['b'])
    ref = solvers.coneqp(P, q, Gq, hq, A=A, b=b)
    for sol in (gs.coneqp(P, q, Gq, hq, A=A, b=b), gs.coneqp(P, q, Gq, hq, A=A, b=b, device_loop=False)):
        _same(sol, ref)
        assert abs(sol['dual slack'] - ref['dual slack']) <= 1e-5 * abs(ref['dual slack']) + 1e-9
    old = dict(solvers.options)
    try:
        solvers.options['maxiters'] = 3                  # options are honoured by the device loop like by the reference
        assert gs.conelp(c, G, h, dims)['iterations'] == 3
    finally:
        solvers.options.clear()
        solvers.options.update(old)


def test_lp_qp_wrappers(ref_cvxopt):
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    c = matrix([-4., -5.])
    G = matrix([[2., 1., -1., 0.], [1., 2., 0., -1.]])
    h = matrix([3., 3., 0., 0.])
    sol = gs.lp(c, G, h)                                  # reference tests/test_examples.py:31-34: x = [1, 1]
    ref = solvers.lp(c, G, h)
    _same(sol, ref)
    assert np.allclose(np.array(sol['x']).ravel(), [1.0, 1.0], atol=1e-6)
    qp = synth.dense_qp(20, 50, seed=9)
    ref = solvers.qp(matrix(qp['P']), matrix(qp['q']), matrix(qp['G']), matrix(qp['h']))
    _same(gs.qp(matrix(qp['P']), matrix(qp['q']), matrix(qp['G']), matrix(qp['h'])), ref)


def test_socp_config3_class_matches_reference(ref_cvxopt):
    """BASELINE configs[2] class scaled down: many small second-order cones through solvers.socp's path (conelp)."""
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    pr = synth.socp(n=96, ncones=48, r=8, seed=3, ml=10)
    c, G, h, dims = matrix(pr['c']), matrix(pr['G']), matrix(pr['h']), pr['dims']
    ref = solvers.conelp(c, G, h, dims)
    sol = gs.conelp(c, G, h, dims, device_loop=False)
    _same(sol, ref)
    assert relerr(np.array(sol['z']).ravel(), np.array(ref['z']).ravel()) < 1e-6


def test_socp_wrapper_returns_reference_layout(ref_cvxopt):
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    pr = synth.socp(n=30, ncones=5, r=6, seed=1, ml=4)
    G, h = pr['G'], pr['h']
    Gl, hl = matrix(G[:4]), matrix(h[:4])
    Gq = [matrix(G[4 + 6 * k:4 + 6 * (k + 1)]) for k in range(5)]
    hq = [matrix(h[4 + 6 * k:4 + 6 * (k + 1)]) for k in range(5)]
    ref = solvers.socp(matrix(pr['c']), Gl, hl, Gq, hq)
    sol = gs.socp(matrix(pr['c']), Gl, hl, Gq, hq)
    _same(sol, ref)
    assert len(sol['sq']) == 5 and relerr(np.array(sol['sq'][2]).ravel(), np.array(ref['sq'][2]).ravel()) < 1e-6
    assert relerr(np.array(sol['zl']).ravel(), np.array(ref['zl']).ravel()) < 1e-6


def test_sdp_and_mixed_cones_through_operators(ref_cvxopt):
    """'s' cones: the operator applies misc.sgemv's trisc convention before G'."""
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    # reference doc example (doc/source/coneprog.rst, conelp with l, q and s blocks)
    c = matrix([-6., -4., -5.])
    G = matrix([[16., 7., 24., -8., 8., -1., 0., -1., 0., 0., 7., -5., 1., -5., 1., -7., 1., -7., -4.],
                [-14., 2., 7., -13., -18., 3., 0., 0., -1., 0., 3., 13., -6., 13., 12., -10., -6., -10., -28.],
                [5., 0., -15., 12., -6., 17., 0., 0., 0., -1., 9., 6., -6., 6., -7., -7., -6., -7., -11.]])
    h = matrix([-3., 5., 12., -2., -14., -13., 10., 0., 0., 0., 68., -30., -19., -30., 99., 23., -19., 23., 10.])
    dims = {'l': 2, 'q': [4, 4], 's': [3]}
    ref = solvers.conelp(c, G, h, dims)
    sol = gs.conelp(c, G, h, dims)
    _same(sol, ref, xt=1e-6)
    assert np.allclose(np.array(sol['x']).ravel(), [-1.22, 9.66e-02, 3.58], rtol=5e-3)     # doc transcript (coneprog.rst:332-335)


def test_sdp_wrapper_matches_reference(ref_cvxopt):
    """solvers.sdp's argument convention (examples/doc/chap8/sdp.py; known answer coneprog.rst:957-969)"""
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    c = matrix([1., -1., 1.])
    G = [matrix([[-7., -11., -11., 3.], [7., -18., -18., 8.], [-2., -8., -8., 1.]])]
    G += [matrix([[-21., -11., 0., -11., 10., 8., 0., 8., 5.], [0., 10., 16., 10., -10., -10., 16., -10., 3.],
                  [-5., 2., -17., 2., -6., 8., -17., -7., 6.]])]
    h = [matrix([[33., -9.], [-9., 26.]]), matrix([[14., 9., 40.], [9., 91., 10.], [40., 10., 15.]])]
    ref = solvers.sdp(c, Gs=G, hs=h)
    sol = gs.sdp(c, Gs=G, hs=h)
    assert sol['status'] == ref['status'] == 'optimal' and sol['iterations'] == ref['iterations']
    assert abs(sol['primal objective'] - ref['primal objective']) < 1e-8 * max(1, abs(ref['primal objective']))
    assert relerr(np.array(sol['x']).ravel(), np.array(ref['x']).ravel()) < 1e-6
    assert np.allclose(np.array(sol['x']).ravel(), [-0.367, 1.898, -0.887], atol=1e-3)
    for k in range(2):
        assert sol['zs'][k].size == ref['zs'][k].size
        assert relerr(np.array(sol['zs'][k]), np.array(ref['zs'][k])) < 1e-6
        assert relerr(np.array(sol['ss'][k]), np.array(ref['ss'][k])) < 1e-5


def test_coneqp_with_soc_and_equalities_through_operators(ref_cvxopt):
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    rng = np.random.default_rng(5)
    n, p = 40, 6
    pr = synth.socp(n=n, ncones=6, r=5, seed=2, ml=12)
    B = rng.standard_normal((n, n)) / np.sqrt(n)
    P = matrix(B.T @ B + 1e-2 * np.eye(n))
    A = matrix(rng.standard_normal((p, n)))
    b = matrix(np.array(A) @ pr['x0']) if 'x0' in pr else matrix(np.zeros(p))
    q, G, h, dims = matrix(pr['c']), matrix(pr['G']), matrix(pr['h']), pr['dims']
    ref = solvers.coneqp(P, q, G, h, dims, A, b)
    sol = gs.coneqp(P, q, G, h, dims, A, b, device_loop=False)
    _same(sol, ref, xt=1e-6)


def test_sparse_lp_through_operators(ref_cvxopt):
    from cvxopt import matrix, spmatrix, solvers
    import cvxopt_amd.solvers as gs
    import scipy.sparse as sp
    rng = np.random.default_rng(9)
    n = 120
    Gs = sp.vstack([sp.eye(n), -sp.eye(n), sp.random(60, n, density=0.05, random_state=3)]).tocoo()
    G = spmatrix(list(Gs.data), list(map(int, Gs.row)), list(map(int, Gs.col)), Gs.shape)
    h = matrix(np.concatenate([np.ones(2 * n), 5.0 + rng.random(60)]))
    c = matrix(rng.standard_normal(n))
    ref = solvers.conelp(c, G, h)
    sol = gs.conelp(c, G, h)
    _same(sol, ref)


def test_socp_config3_full_size_timing(ref_cvxopt):
    """BASELINE configs[2]: n=2048, 1024 cones of dimension 8 -- operators + GPU kktsolver vs the host products with the
    GPU kktsolver hook (both through the unmodified reference driver): same iterates, timing printed."""
    from cvxopt import matrix, solvers, spmatrix
    import cvxopt_amd.solvers as gs
    from cvxopt_amd import kkt
    pr = synth.socp(n=2048, ncones=1024, r=8, seed=0)
    c, G, h, dims = matrix(pr['c']), matrix(pr['G']), matrix(pr['h']), pr['dims']
    t = time.perf_counter(); sol = gs.conelp(c, G, h, dims, device_loop=False); t_ops = time.perf_counter() - t
    ks = kkt.kktsolver_lp(G, dims, spmatrix([], [], [], (0, 2048)))
    t = time.perf_counter(); hook = solvers.conelp(c, G, h, dims, kktsolver=ks); t_hook = time.perf_counter() - t
    ks.engine.close()
    print("config 3 (n=2048, cdim=8192): operators+kktsolver %.2f s, kktsolver hook only %.2f s, %d iterations"
          % (t_ops, t_hook, sol['iterations']))
    _same(sol, hook)
    assert sol['status'] == 'optimal'


@pytest.mark.parametrize("case", ["lp_cone_all", "mixed_partial", "sdp_block"])
def test_coneqp_initvals_run_on_the_device_loop_like_the_reference(ref_cvxopt, case):
    """initvals of solvers.coneqp (coneprog.py:2109-2149): the device-resident loop starts from the caller's point (round 3;
    before, the host driver with device operators took over): same iteration count, objectives and iterates as the reference
    started from the same point; a non-interior s raises the reference's ValueError."""
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    rng = np.random.default_rng(5)
    if case == "lp_cone_all":
        pr = synth.dense_qp(24, 60, seed=12, p=3)
        dims = pr['dims']
        iv = {'x': matrix(rng.standard_normal(24)), 'y': matrix(rng.standard_normal(3)),
              's': matrix(rng.uniform(0.5, 2.0, 60)), 'z': matrix(rng.uniform(0.5, 2.0, 60))}
        args = dict(A=matrix(pr['A']), b=matrix(pr['b']))
        G, h = matrix(pr['G']), matrix(pr['h'])
    elif case == "mixed_partial":
        base = synth.socp(n=24, ncones=4, r=5, seed=2, ml=6)
        dims = base['dims']
        pr = {'P': synth.dense_qp(24, 8, seed=3)['P'], 'q': base['c']}
        m = dims['l'] + sum(dims['q'])
        s0 = np.zeros(m)
        s0[:dims['l']] = rng.uniform(0.5, 1.5, dims['l'])
        o = dims['l']
        for mk in dims['q']:
            s0[o + 1:o + mk] = rng.standard_normal(mk - 1)
            s0[o] = np.linalg.norm(s0[o + 1:o + mk]) + 0.7
            o += mk
        iv = {'s': matrix(s0)}                       # only s is given: x = 0, z = e by default
        args = {}
        G, h = matrix(base['G']), matrix(base['h'])
    else:
        dims = {'l': 2, 'q': [], 's': [3]}
        n = 4
        Gm = rng.standard_normal((2 + 9, n))
        for j in range(n):                           # symmetric 's' columns
            X = Gm[2:, j].reshape(3, 3)
            Gm[2:, j] = (X + X.T).ravel() / 2
        x0 = rng.standard_normal(n)
        S0 = np.eye(3) * 2.0
        h = Gm @ x0 + np.concatenate([[1.0, 1.5], S0.ravel()])
        B = rng.standard_normal((n, n))
        pr = {'P': B.T @ B + 0.1 * np.eye(n), 'q': rng.standard_normal(n)}
        Z0 = np.array([[2.0, 0.3, 0.0], [0.3, 1.5, -0.2], [0.0, -0.2, 1.0]])
        iv = {'x': matrix(x0), 'z': matrix(np.concatenate([[0.7, 0.9], Z0.ravel(order='F')]))}
        args = {}
        G, h = matrix(Gm), matrix(h)
    P, q = matrix(pr['P']), matrix(pr['q'])
    ref = solvers.coneqp(P, q, G, h, dims, initvals=iv, **args)
    sol = gs.coneqp(P, q, G, h, dims, initvals=iv, **args)
    _same(sol, ref)
    assert relerr(np.array(sol['s']).ravel(), np.array(ref['s']).ravel()) < 1e-6
    bad = dict(iv)
    sbad = np.array(iv['s'] if 's' in iv else matrix(1.0, (h.size[0], 1))).ravel().copy()
    sbad[0] = -1.0
    bad['s'] = matrix(sbad)
    with pytest.raises(ValueError):
        gs.coneqp(P, q, G, h, dims, initvals=bad, **args)


@pytest.mark.parametrize("which", ["both", "primal", "dual"])
def test_conelp_primalstart_dualstart_run_on_the_device_loop_like_the_reference(ref_cvxopt, which):
    """primalstart / dualstart of solvers.conelp (coneprog.py:696-739) in the device-resident loop (round 3): the given point is
    taken as it is, only a CONSTRUCTED s or z is shifted into the interior, the 'starting point is optimal' exit belongs to the
    fully constructed start -- same iteration count, objectives and iterates as the reference from the same starts."""
    from cvxopt import matrix, solvers
    import cvxopt_amd.solvers as gs
    pr = synth.socp(n=30, ncones=5, r=6, seed=4, ml=8)
    dims = pr['dims']
    c, G, h = matrix(pr['c']), matrix(pr['G']), matrix(pr['h'])
    rng = np.random.default_rng(7)
    m = dims['l'] + sum(dims['q'])

    def interior_point():
        u = np.zeros(m)
        u[:dims['l']] = rng.uniform(0.5, 1.5, dims['l'])
        o = dims['l']
        for mk in dims['q']:
            u[o + 1:o + mk] = rng.standard_normal(mk - 1)
            u[o] = np.linalg.norm(u[o + 1:o + mk]) + rng.uniform(0.3, 1.0)
            o += mk
        return u
    ps = {'x': matrix(rng.standard_normal(30)), 's': matrix(interior_point())} if which in ("both", "primal") else None
    ds = {'z': matrix(interior_point())} if which in ("both", "dual") else None
    ref = solvers.conelp(c, G, h, dims, primalstart=ps, dualstart=ds)
    sol = gs.conelp(c, G, h, dims, primalstart=ps, dualstart=ds)
    _same(sol, ref)
    assert relerr(np.array(sol['z']).ravel(), np.array(ref['z']).ravel()) < 1e-6
    if ps is not None:
        bad = np.array(ps['s']).ravel().copy()
        bad[dims['l']] = -5.0                              # the first second-order cone leaves its interior
        with pytest.raises(ValueError):
            gs.conelp(c, G, h, dims, primalstart={'x': ps['x'], 's': matrix(bad)}, dualstart=ds)

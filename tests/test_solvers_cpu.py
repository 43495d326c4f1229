"""CPU tests of the host-side pieces of cvxopt_amd.solvers (no GPU): the trisc convention of misc.sgemv and the
alpha/beta combination of the operator closures, against the real reference where it is available."""
import numpy as np
import pytest

from cvxopt_amd import solvers as gs


class _FakeEngine(object):
    """stands in for the device engine: products on the host with the same (which, trans) convention"""
    def __init__(self, G, A, P):
        self.G, self.A, self.P = G, A, P

    def product(self, which, trans, x):
        M = (self.G, self.A, self.P)[which]
        return (M.T if (trans and which != 2) else M) @ x


def test_trisc_matches_reference(ref_cvxopt):
    from cvxopt import matrix, misc
    dims = {'l': 2, 'q': [3], 's': [3, 2]}
    rng = np.random.default_rng(0)
    x = rng.standard_normal(2 + 3 + 9 + 4)
    xr = matrix(x.copy())
    misc.trisc(xr, dims)
    assert np.allclose(gs._trisc(x, dims), np.array(xr).ravel(), rtol=0, atol=0)
    assert gs._trisc(x, {'l': 18, 'q': [], 's': []}) is x


def test_operator_closures_match_sgemv(ref_cvxopt):
    from cvxopt import matrix, misc, base
    dims = {'l': 2, 'q': [3], 's': [3]}
    cdim, n, p = 14, 5, 2
    rng = np.random.default_rng(1)
    G, A = rng.standard_normal((cdim, n)), rng.standard_normal((p, n))
    P = rng.standard_normal((n, n)); P = P + P.T
    Gop, Aop, Pop = gs._operators(_FakeEngine(G, A, P), dims)
    Gm, Am, Pm = matrix(G), matrix(A), matrix(P)
    for trans, nx, ny in (('N', n, cdim), ('T', cdim, n)):
        x = matrix(rng.standard_normal(nx))
        y1, y2 = matrix(rng.standard_normal(ny)), None
        y2 = matrix(np.array(y1).copy())
        Gop(x, y1, alpha=-0.7, beta=1.3, trans=trans)
        misc.sgemv(Gm, x, y2, dims, trans=trans, alpha=-0.7, beta=1.3)
        assert np.allclose(np.array(y1), np.array(y2), rtol=1e-13, atol=1e-13)
    x, y1 = matrix(rng.standard_normal(n)), matrix(rng.standard_normal(n))
    y2 = matrix(np.array(y1).copy())
    Pop(x, y1, alpha=2.0, beta=-1.0)
    base.symv(Pm, x, y2, alpha=2.0, beta=-1.0)
    assert np.allclose(np.array(y1), np.array(y2), rtol=1e-13, atol=1e-13)
    x, y1 = matrix(rng.standard_normal(p)), matrix(np.zeros(n))
    y2 = matrix(np.zeros(n))
    Aop(x, y1, trans='T')
    base.gemv(Am, x, y2, trans='T')
    assert np.allclose(np.array(y1), np.array(y2), rtol=1e-13, atol=1e-13)


def test_dims_default():
    class H(object):
        size = (7, 1)
    assert gs._dims_of(H(), None) == {'l': 7, 'q': [], 's': []}
    assert gs._dims_of(H(), {'l': 1, 'q': (2, 4), 's': [0]}) == {'l': 1, 'q': [2, 4], 's': [0]}


def _host_conelp(monkeypatch):
    """route the wrappers' conelp to the reference's (CPU) so that only the argument packing / result splitting of
    cvxopt_amd.solvers.socp / sdp is under test"""
    from cvxopt import solvers

    def conelp(c, G, h, dims=None, A=None, b=None, primalstart=None, dualstart=None, **kw):
        kw.pop('device_loop', None)
        return solvers.conelp(c, G, h, dims, A, b, primalstart, dualstart, options={'show_progress': False}, **kw)
    monkeypatch.setattr(gs, 'conelp', conelp)


def _same(a, b):
    if a is None or b is None:
        return a is None and b is None
    if isinstance(a, list):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, (float, int, str)):
        return a == b
    return a.size == b.size and np.array_equal(np.array(a), np.array(b))


def test_sdp_wrapper_packs_and_splits_like_the_reference(ref_cvxopt, monkeypatch):
    """the SDP of the reference's doc section 8.6 (examples/doc/chap8/sdp.py; coneprog.rst known answer for x) + an 'l'
    block, sparse blocks, and user starting points"""
    from cvxopt import matrix, solvers, sparse
    _host_conelp(monkeypatch)
    c = matrix([1., -1., 1.])
    G = [matrix([[-7., -11., -11., 3.], [7., -18., -18., 8.], [-2., -8., -8., 1.]])]
    G += [matrix([[-21., -11., 0., -11., 10., 8., 0., 8., 5.], [0., 10., 16., 10., -10., -10., 16., -10., 3.],
                  [-5., 2., -17., 2., -6., 8., -17., -7., 6.]])]
    h = [matrix([[33., -9.], [-9., 26.]]), matrix([[14., 9., 40.], [9., 91., 10.], [40., 10., 15.]])]
    opts = {'show_progress': False}
    ref = solvers.sdp(c, Gs=G, hs=h, options=opts)
    got = gs.sdp(c, Gs=G, hs=h)
    assert set(ref) == set(got)
    assert all(_same(ref[k], got[k]) for k in ref), [k for k in ref if not _same(ref[k], got[k])]
    assert np.allclose(np.array(got['x']).ravel(), [-0.367, 1.898, -0.887], atol=1e-3)      # coneprog.rst:957-969
    assert got['zs'][1].size == (3, 3) and got['ss'][0].size == (2, 2)
    # with an 'l' block, sparse data, and both starting points
    Gl, hl = matrix([[1., 0.], [0., 1.], [1., 1.]]), matrix([10., 10.])
    ps = {'x': matrix(0.0, (3, 1)), 'sl': matrix([1., 1.]), 'ss': [matrix([[2., 0.], [0., 2.]]), matrix(np.eye(3) * 3)]}
    ds = {'zl': matrix([1., 2.]), 'zs': [matrix([[1., 0.], [0., 1.]]), matrix(np.eye(3))]}
    for conv in (lambda M: M, sparse):
        ref = solvers.sdp(c, conv(Gl), hl, [conv(g) for g in G], h, primalstart=ps, dualstart=ds, options=opts)
        got = gs.sdp(c, conv(Gl), hl, [conv(g) for g in G], h, primalstart=ps, dualstart=ds)
        assert ref['status'] == 'optimal'
        assert all(_same(ref[k], got[k]) for k in ref), [k for k in ref if not _same(ref[k], got[k])]
    with pytest.raises(TypeError):
        gs.sdp(c, Gs=[G[0][:3, :]], hs=[h[0]])
    with pytest.raises(TypeError):
        gs.sdp(c, Gs=G, hs=[h[1], h[0]])


def test_socp_wrapper_packs_and_splits_like_the_reference(ref_cvxopt, monkeypatch):
    """the SOCP of the reference's doc section 8.5 (examples/doc/chap8/socp.py) + an 'l' block and starting points"""
    from cvxopt import matrix, solvers
    _host_conelp(monkeypatch)
    c = matrix([-2., 1., 5.])
    G = [matrix([[12., 13., 12.], [6., -3., -12.], [-5., -5., 6.]]),
         matrix([[3., 3., -1., 1.], [-6., -6., -9., 19.], [10., -2., -2., -3.]])]
    h = [matrix([-12., -3., -2.]), matrix([27., 0., 3., -42.])]
    opts = {'show_progress': False}
    ref = solvers.socp(c, Gq=G, hq=h, options=opts)
    got = gs.socp(c, Gq=G, hq=h)
    assert ref['status'] == 'optimal' and set(ref) == set(got)
    assert all(_same(ref[k], got[k]) for k in ref), [k for k in ref if not _same(ref[k], got[k])]
    Gl, hl = matrix([[1., 0.], [0., 1.], [1., -1.]]), matrix([50., 50.])
    ps = {'x': matrix(0.0, (3, 1)), 'sl': matrix([1., 1.]), 'sq': [matrix([2., 0., 0.]), matrix([3., 0., 0., 0.])]}
    ds = {'zl': matrix([1., 1.]), 'zq': [matrix([1., 0., 0.]), matrix([1., 0., 0., 0.])]}
    ref = solvers.socp(c, Gl, hl, G, h, primalstart=ps, dualstart=ds, options=opts)
    got = gs.socp(c, Gl, hl, G, h, primalstart=ps, dualstart=ds)
    assert ref['status'] == 'optimal'
    assert all(_same(ref[k], got[k]) for k in ref), [k for k in ref if not _same(ref[k], got[k])]


def test_external_solver_bridges_are_refused(ref_cvxopt, monkeypatch):
    from cvxopt import matrix
    seen = {}
    monkeypatch.setattr(gs, 'conelp', lambda *a, **k: seen.setdefault('conelp', k) or {})
    monkeypatch.setattr(gs, 'coneqp', lambda *a, **k: seen.setdefault('coneqp', k) or {})
    c, G, h = matrix([1.0, 1.0]), matrix([[-1.0, 0.0], [0.0, -1.0]]), matrix([0.0, 0.0])
    gs.lp(c, G, h, solver=None, kktsolver='ldl')
    assert seen['conelp'] == {'kktsolver': 'ldl'}
    gs.qp(matrix([[1.0, 0.0], [0.0, 1.0]]), c, G, h, solver=None)
    assert seen['coneqp'] == {}
    for call in (lambda: gs.lp(c, G, h, solver='glpk'), lambda: gs.qp(G, c, G, h, solver='mosek'),
                 lambda: gs.socp(c, G, h, solver='mosek'), lambda: gs.sdp(c, G, h, solver='dsdp')):
        with pytest.raises(ValueError):
            call()
    # anything else -- 'default' is what cvxopt.modeling.op.solve passes (modeling.py:2627; found by running the reference's own
    # test_modeling / chap10 examples through cvxopt_amd.solvers in round 4) -- means conelp / coneqp in the reference
    # (coneprog.py:2807, :2877: only the names of the bridges are compared)
    seen.clear()
    gs.lp(c, G, h, solver='default')
    gs.lp(c, G, h, solver='dsdp')                    # not a bridge of lp: falls through to conelp there too
    gs.qp(matrix([[1.0, 0.0], [0.0, 1.0]]), c, G, h, solver='default')
    assert seen['conelp'] == {} and seen['coneqp'] == {}


def test_cvxprog_wrappers_bind_the_gpu_factories_for_the_call_only(ref_cvxopt, monkeypatch):
    """cp / cpl / gp run the reference drivers with cvxopt.misc.kkt_* pointing at the GPU factories during the call."""
    import cvxopt.misc as misc
    from cvxopt import cvxprog as solvers      # the wrappers take the drivers from their defining module (a patched cvxopt.solvers
    from cvxopt_amd import kkt                 # attribute -- e.g. cvxopt.solvers.cp = cvxopt_amd.solvers.cp -- must not recurse)
    before = {n: getattr(misc, n) for n in ('kkt_chol', 'kkt_chol2', 'kkt_ldl', 'kkt_ldl2', 'kkt_qr')}
    seen = {}

    def fake(name):
        def f(*a, **k):
            seen[name] = dict(bound={n: getattr(misc, n) for n in before}, args=a, kw=k)
            return {'status': 'optimal'}
        return f
    for name in ('cp', 'cpl', 'gp'):
        monkeypatch.setattr(solvers, name, fake(name))
    assert gs.cp('F', 'G', 'h', kktsolver='ldl', options={'maxiters': 3}) == {'status': 'optimal'}
    assert seen['cp']['args'] == ('F', 'G', 'h', None, None, None) and seen['cp']['kw'] == {'kktsolver': 'ldl', 'options': {'maxiters': 3}}
    assert seen['cp']['bound'] == {'kkt_chol': kkt.kkt_chol, 'kkt_chol2': kkt.kkt_chol2, 'kkt_ldl': kkt.kkt_ldl,
                                   'kkt_ldl2': kkt.kkt_ldl2, 'kkt_qr': kkt.kkt_qr}
    assert {n: getattr(misc, n) for n in before} == before            # restored afterwards
    gs.cpl('c', 'F')
    assert seen['cpl']['args'][:2] == ('c', 'F') and seen['cpl']['kw'] == {'kktsolver': None}
    gs.gp('K', 'F', 'g')
    assert seen['gp']['args'][:3] == ('K', 'F', 'g')
    assert {n: getattr(misc, n) for n in before} == before
    # an install() made by the user stays in force
    kkt.install()
    try:
        gs.cp('F')
        assert misc.kkt_chol is kkt.kkt_chol
    finally:
        kkt.uninstall()
    assert {n: getattr(misc, n) for n in before} == before

    # and a failing driver does not leave the factories bound
    def boom(*a, **k):
        raise ArithmeticError(3)
    monkeypatch.setattr(solvers, 'cp', boom)
    with pytest.raises(ArithmeticError):
        gs.cp('F')
    assert {n: getattr(misc, n) for n in before} == before


def test_explicit_chol2_with_second_order_or_semidefinite_cones_raises_like_the_reference():
    """misc.kkt_chol2 raises ValueError for q / s cones (misc.py:1381-1384); only the DEFAULT resolves to qr / chol"""
    from cvxopt_amd import solvers as S
    for dims in ({'l': 2, 'q': [3], 's': []}, {'l': 0, 'q': [], 's': [2]}):
        for lp in (True, False):
            with pytest.raises(ValueError):
                S._resolve_kktsolver('chol2', dims, lp)
            assert S._resolve_kktsolver(None, dims, lp) == ('qr' if lp else 'chol')
    assert S._resolve_kktsolver('chol2', {'l': 3, 'q': [], 's': []}, True) == 'chol2'
    with pytest.raises(ValueError):
        S._resolve_kktsolver('qr', {'l': 3, 'q': [], 's': []}, False)

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

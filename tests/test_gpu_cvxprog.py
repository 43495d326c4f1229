"""Nonlinear convex programming drivers (cvxprog.cp / cpl: mnl > 0, H and Df change at every iteration;
SURVEY.md 8(f) row 4) through the GPU factories: factor(W, H, Df) with the Jacobian rows stacked on top of G."""
import numpy as np
import pytest

import cvxopt_amd
from cvxopt_amd import kkt
from helpers import relerr

pytestmark = pytest.mark.gpu


def _acent(ref_matrix, A, b, xstart):
    """analytic centering: minimise -sum log(b - A x)  (reference doc/source/solvers.rst acent example, as F for cp)"""
    from cvxopt import matrix, spdiag, log, div
    m, n = A.size

    def F(x=None, z=None):
        if x is None:
            return 0, matrix(xstart)                     # a point of the domain {x : A x < b}
        y = b - A * x
        if min(y) <= 0:
            return None
        f = -sum(log(y))
        Df = (div(1.0, y)).T * A
        if z is None:
            return f, Df
        H = A.T * spdiag(z[0] * div(1.0, y ** 2)) * A
        return f, Df, H
    return F


def test_cp_with_nonlinear_constraint_matches_reference(ref_cvxopt):
    """cp with one nonlinear constraint and linear inequalities: minimise c'x s.t. sum exp(x_i) <= t-like constraint."""
    from cvxopt import matrix, solvers, exp, spdiag
    rng = np.random.default_rng(2)
    n, m = 12, 30
    c = matrix(rng.standard_normal(n))
    G = matrix(rng.standard_normal((m, n)))
    h = matrix(rng.uniform(1.0, 2.0, m))

    def F(x=None, z=None):
        if x is None:
            return 1, matrix(0.0, (n, 1))
        e = exp(x)
        f = matrix([c.T * x, sum(e) - 2.0 * n])          # objective, and f1(x) = sum exp(x_i) - 2n <= 0
        Df = matrix(0.0, (2, n))
        Df[0, :] = c.T
        Df[1, :] = e.T
        if z is None:
            return f, Df
        H = spdiag(z[1] * e)
        return f, Df, matrix(H)
    solvers.options['show_progress'] = False
    ref = solvers.cp(F, G, h, kktsolver='chol')
    kkt.install()
    try:
        got = solvers.cp(F, G, h, kktsolver='chol')
        got2 = solvers.cp(F, G, h, kktsolver='ldl')
    finally:
        kkt.uninstall()
    for sol in (got, got2):
        assert sol['status'] == ref['status'] == 'optimal'
        assert abs(sol['primal objective'] - ref['primal objective']) <= 1e-7 * max(1.0, abs(ref['primal objective']))
        assert relerr(np.array(sol['x']).ravel(), np.array(ref['x']).ravel()) < 1e-5
        assert relerr(np.array(sol['znl']).ravel(), np.array(ref['znl']).ravel()) < 1e-4
    assert got['iterations'] == ref['iterations'] if 'iterations' in ref else True


def test_cp_analytic_centering_with_equalities(ref_cvxopt):
    from cvxopt import matrix, solvers
    rng = np.random.default_rng(3)
    m, n, p = 60, 20, 4
    A = matrix(rng.standard_normal((m, n)))
    x0 = rng.standard_normal(n)
    b = matrix(np.array(A) @ x0 + rng.uniform(0.5, 1.5, m))
    Ae = matrix(rng.standard_normal((p, n)))
    be = matrix(np.array(Ae) @ x0)
    F = _acent(matrix, A, b, x0)
    solvers.options['show_progress'] = False
    ref = solvers.cp(F, A=Ae, b=be, kktsolver='chol')
    kkt.install()
    try:
        got = solvers.cp(F, A=Ae, b=be, kktsolver='chol')
    finally:
        kkt.uninstall()
    assert got['status'] == ref['status']
    assert abs(got['primal objective'] - ref['primal objective']) <= 1e-7 * max(1.0, abs(ref['primal objective']))
    assert relerr(np.array(got['x']).ravel(), np.array(ref['x']).ravel()) < 1e-5

"""CPU parity of the second-order-cone operations the device-resident loops use.  cone_ops.h is compiled for the host as well
(`mi355kkt_test_cone_op_host` runs the very same functions the kernels call); here they are compared with the reference's
misc / misc_solvers on random cones -- sprod, sinv, ssqr, scale2 (both ways), scale (W and W^-1), jnrm2, max_step,
compute_scaling and a chain of update_scaling steps."""
import ctypes as C

import numpy as np
import pytest

from cvxopt_amd import _capi


def _op(op, mk, x, y=None, w=None, arg=0):
    L = _capi.lib()
    p = lambda a: a.ctypes.data if a is not None else None
    rc = L.mi355kkt_test_cone_op_host(op, mk, arg, p(x), p(y), p(w))
    assert rc == 0


def _cone_point(rng, mk, margin=0.5):
    v = rng.standard_normal(mk)
    v[0] = np.linalg.norm(v[1:]) + margin + rng.random()
    return v


@pytest.mark.parametrize("mk", [1, 2, 3, 8, 33])
def test_cone_vector_ops_match_reference(ref_cvxopt, mk):
    from cvxopt import matrix, misc
    rng = np.random.default_rng(mk)
    dims = {'l': 0, 'q': [mk], 's': []}
    x, y = rng.standard_normal(mk), _cone_point(rng, mk)
    # sprod
    a, r = x.copy(), matrix(x)
    _op(0, mk, a, y); misc.sprod(r, matrix(y), dims)
    assert np.allclose(a, np.array(r).ravel(), rtol=1e-13, atol=1e-13)
    # sinv
    a, r = x.copy(), matrix(x)
    _op(1, mk, a, y); misc.sinv(r, matrix(y), dims)
    assert np.allclose(a, np.array(r).ravel(), rtol=1e-12, atol=1e-12)
    # ssqr
    a, r = np.zeros(mk), matrix(0.0, (mk, 1))
    _op(2, mk, a, y); misc.ssqr(r, matrix(y), dims)
    assert np.allclose(a, np.array(r).ravel(), rtol=1e-13, atol=1e-13)
    # scale2, both directions
    for inv, flag in ((0, 'N'), (1, 'I')):
        a, r = x.copy(), matrix(x)
        _op(3, mk, a, y, arg=inv); misc.scale2(matrix(y), r, dims, inverse=flag)
        assert np.allclose(a, np.array(r).ravel(), rtol=1e-12, atol=1e-12)
    # jnrm2, max_step
    w = np.zeros(1)
    _op(5, mk, y.copy(), w=w)
    assert abs(w[0] - misc.jnrm2(matrix(y))) <= 1e-13 * abs(w[0])
    _op(8, mk, x.copy(), w=w)
    assert abs(w[0] - misc.max_step(matrix(x), dims)) <= 1e-13 * max(1.0, abs(w[0]))


@pytest.mark.parametrize("mk", [2, 5, 16])
def test_nesterov_todd_scaling_of_a_cone_matches_reference(ref_cvxopt, mk):
    from cvxopt import matrix, misc
    rng = np.random.default_rng(100 + mk)
    dims = {'l': 0, 'q': [mk], 's': []}
    s, z = _cone_point(rng, mk), _cone_point(rng, mk)
    lm = matrix(0.0, (mk, 1))
    W = misc.compute_scaling(matrix(s), matrix(z), lm, dims)
    w = np.zeros(2 * mk + 1)
    _op(6, mk, s.copy(), z.copy(), w)
    v, lam, beta = w[:mk], w[mk:2 * mk], w[2 * mk]
    assert np.allclose(v, np.array(W['v'][0]).ravel(), rtol=1e-12, atol=1e-13)
    assert np.allclose(lam, np.array(lm).ravel(), rtol=1e-12, atol=1e-13)
    assert abs(beta - W['beta'][0]) <= 1e-13 * beta
    # scale: W x and W^-1 x with that scaling
    x = rng.standard_normal(mk)
    for inv, kw in ((0, {}), (1, {'inverse': 'I'})):
        a, r = x.copy(), matrix(x)
        _op(4, mk, a, v.copy(), np.array([beta]), arg=inv)
        misc.scale(r, W, **kw)
        assert np.allclose(a, np.array(r).ravel(), rtol=1e-12, atol=1e-12)
    # three update_scaling steps in a row (misc.py:503-573), state carried on both sides
    for step in range(3):
        ds, dz = _cone_point(rng, mk, 0.2), _cone_point(rng, mk, 0.2)
        rs, rz = matrix(ds), matrix(dz)
        misc.update_scaling(W, lm, rs, rz)
        a, b = ds.copy(), dz.copy()
        _op(7, mk, a, b, w)
        assert np.allclose(w[:mk], np.array(W['v'][0]).ravel(), rtol=1e-11, atol=1e-12), step
        assert np.allclose(w[mk:2 * mk], np.array(lm).ravel(), rtol=1e-11, atol=1e-12), step
        assert abs(w[2 * mk] - W['beta'][0]) <= 1e-12 * abs(W['beta'][0]), step
        assert np.allclose(a, np.array(rs).ravel(), rtol=1e-12, atol=1e-13)      # normalised in place like the reference

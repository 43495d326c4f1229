"""CPU parity of the 's'-block operations of the device-resident loops.  cone_ops_s.h is written once over a "team" of
threads; `mi355kkt_test_sdp_op_host` runs the very same functions with a team of one.  They are compared here with the
reference's misc / misc_solvers (compute_scaling, update_scaling, scale, scale2, sprod, sinv, max_step with and without
sigma) on random blocks.  LAPACK's eigen / singular vectors are unique up to signs (and order inside clusters), so scaling
matrices are compared through the quantities that do not depend on that choice: r r', rti rti', the sorted lmbda, and the
defining identities r' z r = diag(lmbda), r' s^-1 r = diag(lmbda)^-1, rti = r^-T."""
import numpy as np
import pytest

from cvxopt_amd import _capi


def _op(op, m, x, y=None, r=None, rti=None, lam=None, arg=0):
    L = _capi.lib()
    p = lambda a: a.ctypes.data if a is not None else None
    return L.mi355kkt_test_sdp_op_host(op, m, arg, p(x), p(y), p(r), p(rti), p(lam))


def _F(a):
    return np.asfortranarray(a, dtype=float)


def _spd(rng, m, cond=1e3):
    q, _ = np.linalg.qr(rng.standard_normal((m, m)))
    ev = np.logspace(0, -np.log10(cond), m) if m > 1 else np.ones(1)
    a = (q * ev) @ q.T
    return _F(0.5 * (a + a.T))


def _sym(rng, m):
    a = rng.standard_normal((m, m))
    return _F(0.5 * (a + a.T))


def _ref_mat(a):
    from cvxopt import matrix
    return matrix(np.asarray(a).ravel(order='F'), (a.size, 1))


def _lower(a):
    return np.tril(a)


SIZES = [1, 2, 3, 4, 7, 16, 33]


@pytest.mark.parametrize("m", SIZES)
def test_elementwise_block_ops_match_reference(ref_cvxopt, m):
    from cvxopt import matrix, misc
    rng = np.random.default_rng(m)
    dims = {'l': 0, 'q': [], 's': [m]}
    x, y = _sym(rng, m), _sym(rng, m)
    lam = rng.random(m) + 0.1
    # sprod, diag = 'N'
    a = x.copy(order='F'); xr = _ref_mat(x); yr = _ref_mat(y)
    assert _op(1, m, a, y.copy(order='F')) == 0
    misc.sprod(xr, yr, dims)
    ref = np.array(xr).reshape(m, m, order='F')
    assert np.allclose(_lower(a), _lower(ref), rtol=1e-13, atol=1e-13)
    assert np.array_equal(a, a.T)                                 # both triangles kept
    # sprod diag = 'D' and sinv
    a = x.copy(order='F'); xr = _ref_mat(x)
    assert _op(2, m, a, lam=lam) == 0
    misc.sprod(xr, matrix(lam), dims, diag='D')
    assert np.allclose(_lower(a), _lower(np.array(xr).reshape(m, m, order='F')), rtol=1e-14, atol=0)
    a = x.copy(order='F'); xr = _ref_mat(x)
    assert _op(2, m, a, lam=lam, arg=1) == 0
    misc.sinv(xr, matrix(lam), dims)
    assert np.allclose(_lower(a), _lower(np.array(xr).reshape(m, m, order='F')), rtol=1e-14, atol=0)
    # scale2 both ways (full matrix: the inverse is applied to nonsymmetric blocks too)
    g = _F(rng.standard_normal((m, m)))
    for inv, flag in ((0, 'N'), (1, 'I')):
        a = g.copy(order='F'); xr = _ref_mat(g)
        assert _op(3, m, a, lam=lam, arg=inv) == 0
        misc.scale2(matrix(lam), xr, dims, inverse=flag)
        assert np.allclose(a, np.array(xr).reshape(m, m, order='F'), rtol=1e-14, atol=0)


@pytest.mark.parametrize("m", SIZES)
def test_max_step_and_eigendecomposition_match_reference(ref_cvxopt, m):
    from cvxopt import matrix, misc
    rng = np.random.default_rng(10 + m)
    dims = {'l': 0, 'q': [], 's': [m]}
    x = _sym(rng, m)
    nrm = np.linalg.norm(x)
    out = np.zeros(m)
    assert _op(4, m, x.copy(order='F'), lam=out) == 0
    t_ref = misc.max_step(_ref_mat(x), dims)
    assert abs(-out[0] - t_ref) <= 1e-13 * max(1.0, nrm)
    # with sigma: eigenvectors in place, eigenvalues ascending
    a = x.copy(order='F'); sig = np.zeros(m)
    assert _op(5, m, a, lam=sig) == 0
    xr = _ref_mat(x); sr = matrix(0.0, (m, 1))
    t2 = misc.max_step(xr, dims, sigma=sr)
    assert np.allclose(sig, np.array(sr).ravel(), rtol=0, atol=1e-13 * max(1.0, nrm))
    assert abs(-sig[0] - t2) <= 1e-13 * max(1.0, nrm)
    assert np.allclose(a.T @ a, np.eye(m), atol=1e-13)
    assert np.allclose(a @ np.diag(sig) @ a.T, x, atol=1e-12 * max(1.0, nrm))


def test_eigendecomposition_with_repeated_and_opposite_eigenvalues():
    # +-1 pairs and a triple eigenvalue: the shifted one-sided Jacobi must still return an orthonormal eigenbasis
    rng = np.random.default_rng(3)
    m = 8
    q, _ = np.linalg.qr(rng.standard_normal((m, m)))
    ev = np.array([-1.0, -1.0, 1.0, 1.0, 1.0, 0.0, 2.0, -2.0])
    x = _F((q * ev) @ q.T)
    x = _F(0.5 * (x + x.T))
    a = x.copy(order='F'); sig = np.zeros(m)
    assert _op(5, m, a, lam=sig) == 0
    assert np.allclose(sig, np.sort(ev), atol=1e-13)
    assert np.allclose(a.T @ a, np.eye(m), atol=1e-13)
    assert np.allclose(a @ np.diag(sig) @ a.T, x, atol=1e-13)


@pytest.mark.parametrize("m", SIZES)
@pytest.mark.parametrize("cond", [1e1, 1e8])
def test_compute_scaling_matches_reference(ref_cvxopt, m, cond):
    from cvxopt import matrix, misc
    rng = np.random.default_rng(20 + m)
    dims = {'l': 0, 'q': [], 's': [m]}
    s, z = _spd(rng, m, cond), _spd(rng, m, cond)
    r, rti, lam = _F(np.zeros((m, m))), _F(np.zeros((m, m))), np.zeros(m)
    assert _op(6, m, s.copy(order='F'), z.copy(order='F'), r, rti, lam) == 0
    lr = matrix(0.0, (m, 1))
    W = misc.compute_scaling(_ref_mat(s), _ref_mat(z), lr, dims)
    rr, rtir, lref = np.array(W['r'][0]), np.array(W['rti'][0]), np.array(lr).ravel()
    # singular values, descending like LAPACK's (dgesvd is accurate to eps * sigma_max, the Jacobi iteration to eps * sigma_i)
    assert np.allclose(lam, lref, rtol=1e-11, atol=1e-14 * lref.max())
    scale = np.linalg.norm(rr) ** 2
    assert np.allclose(r @ r.T, rr @ rr.T, rtol=0, atol=1e-10 * scale)    # invariant under r -> r D
    assert np.allclose(rti @ rti.T, rtir @ rtir.T, rtol=0, atol=1e-10 * np.linalg.norm(rtir) ** 2)
    # defining identities (misc.py:358-370)
    assert np.allclose(r.T @ z @ r, np.diag(lam), atol=1e-11 * lam.max() * max(1.0, cond ** 0.5))
    assert np.allclose(rti.T @ r, np.eye(m), atol=1e-9)
    assert np.allclose(rti.T @ s @ rti, np.diag(lam), atol=1e-11 * lam.max() * max(1.0, cond ** 0.5))


@pytest.mark.parametrize("m", [2, 5, 16])
def test_scale_all_four_modes_match_reference(ref_cvxopt, m):
    from cvxopt import matrix, misc
    rng = np.random.default_rng(30 + m)
    dims = {'l': 0, 'q': [], 's': [m]}
    s, z = _spd(rng, m, 1e2), _spd(rng, m, 1e2)
    lr = matrix(0.0, (m, 1))
    W = misc.compute_scaling(_ref_mat(s), _ref_mat(z), lr, dims)
    r, rti = _F(np.array(W['r'][0])), _F(np.array(W['rti'][0]))
    x = _sym(rng, m)
    for trans in ('N', 'T'):
        for inverse in ('N', 'I'):
            a = x.copy(order='F'); xr = _ref_mat(x)
            arg = (1 if inverse == 'I' else 0) | (2 if trans == 'T' else 0)
            assert _op(0, m, a, r=r, rti=rti, arg=arg) == 0
            misc.scale(xr, W, trans=trans, inverse=inverse)
            ref = np.array(xr).reshape(m, m, order='F')
            assert np.allclose(_lower(a), _lower(ref), rtol=1e-12, atol=1e-12 * np.abs(ref).max()), (trans, inverse)
            assert np.array_equal(a, a.T)


@pytest.mark.parametrize("m", [1, 2, 5, 16, 33])
def test_chain_of_update_scaling_steps_matches_reference(ref_cvxopt, m):
    """A sequence of interior-point-like updates: the reference's update_scaling (misc.py:592-634) against the device code's,
    each starting from ITS OWN previous scaling; compared through r r', rti rti' and the sorted lmbda."""
    from cvxopt import matrix, misc
    rng = np.random.default_rng(40 + m)
    dims = {'l': 0, 'q': [], 's': [m]}
    s, z = _spd(rng, m, 1e2), _spd(rng, m, 1e2)
    r, rti, lam = _F(np.zeros((m, m))), _F(np.zeros((m, m))), np.zeros(m)
    assert _op(6, m, s.copy(order='F'), z.copy(order='F'), r, rti, lam) == 0
    lr = matrix(0.0, (m + 1, 1))
    W = misc.compute_scaling(_ref_mat(s), _ref_mat(z), lr, dims)
    for step in range(6):
        # factors Ls, Lz of "updated variables in the current scaling": any nonsingular matrices will do for the algebra;
        # the loops produce lmbda^(1/2) Q (1 + step sigma)^(1/2), here random well-conditioned ones with a signed permutation
        # difference between the two sides (what the Jacobi / LAPACK eigenvector ambiguity amounts to)
        base_s = np.linalg.cholesky(_spd(rng, m, 30.0))
        base_z = np.linalg.cholesky(_spd(rng, m, 30.0))
        # express the same updated variables in each side's own scaled coordinates: r_ours = r_ref D
        D = np.linalg.solve(np.array(W['r'][0]), r)            # signed permutation (up to rounding)
        Ls_ref, Lz_ref = base_s, base_z
        Ls, Lz = _F(D.T @ base_s), _F(D.T @ base_z)             # st = D' st_ref D  =>  factor D' Ls_ref
        sref = matrix(0.0, (m * m, 1)); zref = matrix(0.0, (m * m, 1))
        sref[:] = Ls_ref.ravel(order='F'); zref[:] = Lz_ref.ravel(order='F')
        misc.update_scaling(W, lr, sref, zref)
        assert _op(7, m, Ls, Lz, r, rti, lam) == 0
        rr, rtir = np.array(W['r'][0]), np.array(W['rti'][0])
        lref = np.array(lr).ravel()[:m]
        assert np.allclose(lam, lref, rtol=1e-10, atol=0), step
        assert np.allclose(r @ r.T, rr @ rr.T, rtol=0, atol=1e-9 * np.linalg.norm(rr) ** 2), step
        assert np.allclose(rti @ rti.T, rtir @ rtir.T, rtol=0, atol=1e-9 * np.linalg.norm(rtir) ** 2), step
        assert np.allclose(rti.T @ r, np.eye(m), atol=1e-8), step


def test_potrf_block_reports_the_failing_pivot():
    rng = np.random.default_rng(5)
    m = 9
    a = _spd(rng, m, 10.0)
    c = a.copy(order='F')
    assert _op(8, m, c) == 0
    assert np.allclose(c, np.linalg.cholesky(a), atol=1e-13)
    a[4, 4] = -1.0
    assert _op(8, m, a.copy(order='F')) == 5


@pytest.mark.parametrize("nt", [4, 64])
@pytest.mark.parametrize("m", [2, 5, 16, 33])
def test_operations_with_a_team_of_host_threads(m, nt):
    """the same SPMD source run by nt host threads with a pthread barrier as the team barrier (`mi355kkt_test_sdp_op_host_team`):
    real concurrency inside a team, as on the device; results must be those of the team of one"""
    L = _capi.lib()
    p = lambda a: a.ctypes.data if a is not None else None
    rng = np.random.default_rng(50 + m)
    s, z, x, y = _spd(rng, m, 1e2), _spd(rng, m, 1e2), _sym(rng, m), _sym(rng, m)
    r, rti, lam = _F(np.zeros((m, m))), _F(np.zeros((m, m))), np.zeros(m)
    r1, rti1, lam1 = r.copy(order='F'), rti.copy(order='F'), lam.copy()
    sc, zc = s.copy(order='F'), z.copy(order='F')
    assert _op(6, m, sc, zc, r, rti, lam) == 0
    sc, zc = s.copy(order='F'), z.copy(order='F')
    assert L.mi355kkt_test_sdp_op_host_team(6, m, 0, nt, p(sc), p(zc), p(r1), p(rti1), p(lam1)) == 0
    assert np.allclose(lam, lam1, rtol=1e-12, atol=0)
    assert np.allclose(r @ r.T, r1 @ r1.T, rtol=0, atol=1e-11 * np.linalg.norm(r) ** 2)
    for op, arg, second in ((0, 0, None), (0, 3, None), (1, 0, y), (3, 1, None), (2, 0, None)):
        a, b = x.copy(order='F'), x.copy(order='F')
        yc = None if second is None else second.copy(order='F')
        lam0 = rng.random(m) + 0.2
        assert _op(op, m, a, yc, r, rti, lam0, arg=arg) == 0
        yc = None if second is None else second.copy(order='F')
        assert L.mi355kkt_test_sdp_op_host_team(op, m, arg, nt, p(b), p(yc), p(r), p(rti), p(lam0)) == 0
        assert np.allclose(a, b, rtol=1e-13, atol=1e-13 * np.abs(a).max()), (op, arg)
    a, sig = x.copy(order='F'), np.zeros(m)
    assert L.mi355kkt_test_sdp_op_host_team(5, m, 0, nt, p(a), None, None, None, p(sig)) == 0
    assert np.allclose(sig, np.linalg.eigvalsh(x), atol=1e-12 * np.linalg.norm(x))
    assert np.allclose(a @ np.diag(sig) @ a.T, x, atol=1e-12 * np.linalg.norm(x))

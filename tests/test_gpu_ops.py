"""Per-kernel parity of the stand-alone device operators (include/mi355kkt.h, `mi355kkt_op_*`)
against NumPy/LAPACK on the same seeded inputs.  All calls go through the C ABI."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sla

pytestmark = pytest.mark.gpu

TOL = 2e-13   # FP64 relative tolerance for O(1e3)-term dot products (stated per north_star)


def dev(capi, a):
    return capi.DeviceBuffer.from_array(np.asfortranarray(a))


@pytest.mark.parametrize("n,m,use_di,use_H", [
    (128, 256, True, True), (200, 333, True, True), (129, 17, True, False), (384, 1000, False, True),
    (1000, 2048, True, True), (64, 5000, True, False), (1, 7, True, True), (300, 0, True, True),
    # contraction lengths around the 8-deep chunks of the register-staged kernel: a single row, one chunk exactly, one chunk
    # plus one row, even / odd chunk counts with and without a remainder
    (5, 1, True, True), (70, 8, True, False), (130, 9, False, True), (70, 16, True, True), (70, 24, True, False),
    (260, 31, True, True), (140, 2, False, False),
    # 33 x 33 tiles: 512 unsplit work items (diagonal tiles among them: block masks of the pipelined loop) + 49 split ones
    (4224, 512, True, True), (4224, 256, False, False),
])
def test_syrk_scaled_matches_numpy(capi, n, m, use_di, use_H):
    rng = np.random.default_rng(n * 7 + m)
    G = np.asfortranarray(rng.standard_normal((m, n)))
    di = 10.0 ** rng.uniform(-2, 2, m)
    B = rng.standard_normal((n, n))
    H = B @ B.T
    Hl = np.tril(H) + np.triu(np.full((n, n), 1e300), 1)     # poison the strict upper: must be ignored
    Gs = (di[:, None] * G) if use_di else G
    ref = Gs.T @ Gs + (H if use_H else 0.0)
    dG, ddi, dH = dev(capi, G if m else np.zeros((1, n))), dev(capi, di if m else np.zeros(1)), dev(capi, Hl)
    dS = dev(capi, np.full((n, n), np.nan))
    ms = C.c_float()
    rc = capi.lib().mi355kkt_op_syrk_scaled(dG.ptr, max(m, 1), m, n, ddi.ptr if use_di else None,
                                            dH.ptr if use_H else None, n, dS.ptr, n, C.byref(ms))
    capi.check(rc, "op_syrk_scaled")
    S = dS.to_array((n, n))
    il = np.tril_indices(n)
    scale = np.abs(Gs).T @ np.abs(Gs) + (np.abs(H) if use_H else 0.0) + 1e-300
    err = np.max(np.abs(S[il] - ref[il]) / scale[il])
    assert err < TOL, err
    # strict upper triangle must be untouched
    iu = np.triu_indices(n, 1)
    assert np.all(np.isnan(S[iu]))


def test_syrk_scaled_odd_leading_dimension(capi):
    rng = np.random.default_rng(5)
    m, n, ld = 333, 257, 335
    Gbig = np.asfortranarray(rng.standard_normal((ld, n)))
    di = rng.uniform(0.5, 2, m)
    dG, ddi, dS = dev(capi, Gbig), dev(capi, di), dev(capi, np.zeros((n, n)))
    capi.check(capi.lib().mi355kkt_op_syrk_scaled(dG.ptr, ld, m, n, ddi.ptr, None, n, dS.ptr, n, None), "syrk")
    S = dS.to_array((n, n))
    Gs = di[:, None] * Gbig[:m]
    ref = Gs.T @ Gs
    il = np.tril_indices(n)
    assert np.max(np.abs(S[il] - ref[il])) < 1e-10


@pytest.mark.parametrize("n", [1, 16, 64, 100, 128, 129, 200, 513, 1024, 1500])
def test_potrf_matches_lapack(capi, n):
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n + 8))
    A = B @ B.T + 0.1 * np.eye(n)
    dA = dev(capi, np.tril(A) + np.triu(np.full((n, n), np.nan), 1))
    info, ms = C.c_int(-7), C.c_float()
    capi.check(capi.lib().mi355kkt_op_potrf(dA.ptr, n, n, C.byref(info), C.byref(ms)), "op_potrf")
    assert info.value == 0
    L = np.tril(dA.to_array((n, n)))
    Lref = np.linalg.cholesky(A)
    assert np.max(np.abs(L - Lref)) / np.max(np.abs(Lref)) < 1e-11
    resid = np.linalg.norm(L @ L.T - A) / np.linalg.norm(A)
    assert resid < 1e-14 * max(10, n)


@pytest.mark.parametrize("n,bad", [(64, 10), (300, 0), (300, 150), (300, 299), (1000, 700)])
def test_potrf_reports_first_bad_pivot_like_lapack(capi, n, bad):
    rng = np.random.default_rng(n + bad)
    B = rng.standard_normal((n, n))
    A = B @ B.T + n * np.eye(n)
    A[bad, bad] = -1.0                                  # leading minor of order bad+1 is not PD
    _, linfo = sla.lapack.dpotrf(A, lower=1)
    assert linfo == bad + 1
    dA = dev(capi, A)
    info = C.c_int(0)
    capi.check(capi.lib().mi355kkt_op_potrf(dA.ptr, n, n, C.byref(info), None), "op_potrf")
    assert info.value == linfo


def test_potrf_nan_is_reported(capi):
    n = 200
    A = np.eye(n)
    A[77, 77] = np.nan
    dA = dev(capi, A)
    info = C.c_int(0)
    capi.check(capi.lib().mi355kkt_op_potrf(dA.ptr, n, n, C.byref(info), None), "op_potrf")
    assert info.value == 78


@pytest.mark.parametrize("n,nrhs", [(1, 1), (64, 1), (65, 2), (128, 1), (130, 3), (1000, 1), (1025, 2)])
@pytest.mark.parametrize("trans", [0, 1])
def test_trsm_lower_matches_lapack(capi, n, nrhs, trans):
    rng = np.random.default_rng(n + nrhs + trans)
    L = np.tril(rng.standard_normal((n, n))) / np.sqrt(n) + 2.0 * np.eye(n)
    X = rng.standard_normal((n, nrhs))
    dL, dX = dev(capi, L + np.triu(np.full((n, n), np.nan), 1)), dev(capi, X)
    capi.check(capi.lib().mi355kkt_op_trsm_lower(dL.ptr, n, n, dX.ptr, n, nrhs, trans, None), "trsm")
    got = dX.to_array((n, nrhs))
    ref = sla.solve_triangular(L, X, lower=True, trans='T' if trans else 'N')
    assert np.max(np.abs(got - ref)) / np.max(np.abs(ref)) < 1e-12


@pytest.mark.parametrize("m,n", [(1, 1), (512, 256), (1000, 333), (333, 1000), (4097, 130)])
def test_gemv_pair_matches_numpy(capi, m, n):
    rng = np.random.default_rng(m + n)
    G = np.asfortranarray(rng.standard_normal((m, n)))
    w = 10.0 ** rng.uniform(-1, 1, m)
    z, y0, x = rng.standard_normal(m), rng.standard_normal(n), rng.standard_normal(n)
    dG, dw, dz, dy, dx = dev(capi, G), dev(capi, w), dev(capi, z), dev(capi, y0), dev(capi, x)
    dzs, dout = dev(capi, np.zeros(m)), dev(capi, np.zeros(m))
    L = capi.lib()
    capi.check(L.mi355kkt_op_gemv_t_scaled(dG.ptr, m, m, n, dw.ptr, dz.ptr, dzs.ptr, dy.ptr, None), "gemv_t")
    zs = dzs.to_array((m,))
    assert np.allclose(zs, w * z, rtol=1e-15, atol=0)
    y = dy.to_array((n,))
    ref = y0 + (w[:, None] * G).T @ (w * z)
    assert np.max(np.abs(y - ref)) / np.max(np.abs(ref)) < 1e-13
    capi.check(L.mi355kkt_op_gemv_n_scaled(dG.ptr, m, m, n, dw.ptr, dx.ptr, dzs.ptr, dout.ptr, None), "gemv_n")
    out = dout.to_array((m,))
    ref2 = w * (G @ x) - w * z
    assert np.max(np.abs(out - ref2)) / np.max(np.abs(ref2)) < 1e-13

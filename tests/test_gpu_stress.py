"""Soak test of the inter-workgroup hand-off protocols on the hot path: the persistent tile Cholesky
(ticketed tiles, streamed 16-column panels), the persistent triangular solves (data-tagged granules) and the sparse fronts'
tile kernel run back to back many times on the SAME inputs with randomised unrelated work in front of every repetition; every
repetition must reproduce the first one BIT FOR BIT (the kernels are deterministic by construction: fixed-order sums, no
floating-point atomics) and no hand-off timeout may be flagged.

Repetitions: $MI355KKT_STRESS_ITERS (default 150 per size inside the suite; tools/calls/*stress* runs 2000)."""
import ctypes as C
import os

import numpy as np
import pytest

from cvxopt_amd import _capi, kkt, synth

pytestmark = pytest.mark.gpu

ITERS = int(os.environ.get("MI355KKT_STRESS_ITERS", "150"))


def _noise(L, rng, bufs):
    """unrelated device work of random size in front of a repetition: copies between scratch buffers on the legacy stream
    and a small product on a scratch handle (different kernels, different durations, different compute units busy)"""
    a, b = bufs
    nbytes = int(rng.integers(1, a.nbytes // 8)) * 8
    k = int(rng.integers(0, 4))
    if k == 0:
        return
    if k >= 1:
        _capi.check(L.mi355kkt_memcpy_d2d(b.ptr, a.ptr, nbytes), "d2d")
    if k >= 2:
        _capi.check(L.mi355kkt_memcpy_d2d(a.ptr, b.ptr, nbytes // 2 or 8), "d2d")
    if k == 3:
        ms = C.c_float(0)
        n = int(rng.integers(64, 512))
        _capi.check(L.mi355kkt_op_gemv_n_scaled(C.c_void_p(a.ptr), n, n, n, None, C.c_void_p(b.ptr), C.c_void_p(b.ptr),
                                                C.c_void_p(b.ptr + 8 * 1024), C.byref(ms)), "gemv")


@pytest.mark.parametrize("n,m", [(1024, 1536), (2048, 1024), (8192, 1024), (1500, 700)])
def test_dense_factor_and_solves_are_bit_reproducible_under_load(n, m):
    L = _capi.lib()
    rng = np.random.default_rng(n + m)
    pr = synth.dense_qp(n, m, seed=n)
    eng = kkt._Engine(_capi.CHOL2, pr['G'], pr['dims'], kkt._EmptyA(n))
    Hd = _capi.DeviceBuffer.from_array(np.asfortranarray(pr['P']))
    eng._mode = "dense"
    eng.set_H_device(Hd.ptr, n)
    di = rng.uniform(0.3, 3.0, m)
    did = _capi.DeviceBuffer.from_array(di)
    bx, bz = rng.standard_normal(n), rng.standard_normal(m)
    bxd, bzd = _capi.DeviceBuffer.from_array(bx), _capi.DeviceBuffer.from_array(bz)
    xd, zd = _capi.DeviceBuffer(8 * n), _capi.DeviceBuffer(8 * m)
    yd = _capi.DeviceBuffer(8)
    noise = (_capi.DeviceBuffer(8 << 20), _capi.DeviceBuffer(8 << 20))
    ref_x = ref_z = ref_L = None
    iters = ITERS if n < 8192 else max(20, ITERS // 3)
    try:
        for it in range(iters):
            _noise(L, rng, noise)
            eng.factor_device(di_ptr=did.ptr)
            _capi.check(L.mi355kkt_memcpy_d2d(xd.ptr, bxd.ptr, 8 * n), "d2d")
            _capi.check(L.mi355kkt_memcpy_d2d(zd.ptr, bzd.ptr, 8 * m), "d2d")
            eng.solve_device(xd.ptr, yd.ptr, zd.ptr)
            eng.sync()
            x, z = xd.to_array((n,)), zd.to_array((m,))
            assert np.all(np.isfinite(x)) and np.all(np.isfinite(z)), it
            if ref_x is None:
                ref_x, ref_z = x, z
            else:
                assert np.array_equal(x, ref_x) and np.array_equal(z, ref_z), \
                    "repetition %d differs from the first: max |dx| = %.3e" % (it, np.max(np.abs(x - ref_x)))
            if it in (0, iters // 2, iters - 1):
                Lf = np.zeros((n, n), order='F')
                _capi.check(L.mi355kkt_get_factor(eng.h, Lf.ctypes.data_as(C.c_void_p), n), "get_factor")
                Lf = np.tril(Lf)
                if ref_L is None:
                    ref_L = Lf
                    # ... and the first one is RIGHT: L L' = S to rounding, the solve satisfies the KKT system
                    S = pr['P'] + (pr['G'] * (di * di)[:, None]).T @ pr['G']
                    assert np.max(np.abs(Lf @ Lf.T - S)) <= 1e-12 * np.max(np.abs(S)) * n ** 0.5
                else:
                    assert np.array_equal(Lf, ref_L), "factor of repetition %d differs bitwise" % it
    finally:
        eng.close()
        for b in (Hd, did, bxd, bzd, xd, zd, yd) + noise:
            b.free()


class _Sp(object):
    """the minimal spmatrix surface cvxopt_amd reads (.size, .CCS)"""

    def __init__(self, A):
        import scipy.sparse as sp
        A = sp.csc_matrix(A)
        A.sort_indices()
        self.size = A.shape
        self.CCS = (A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64))


def test_sparse_factor_and_solves_are_bit_reproducible_under_load():
    """the fronts' tile kernel (potrf_tiles_vb_kernel) + level-scheduled solves of the sparse engine, 3-D Laplacian box-QP"""
    import scipy.sparse as sp
    L = _capi.lib()
    rng = np.random.default_rng(7)
    k = 26                                       # n = 26^3 = 17576: top separators of several hundred columns
    P = synth.grid_laplacian(k)
    n = k ** 3
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()
    dims = {'l': 2 * n, 'q': [], 's': []}
    f = kkt.kkt_chol2(_Sp(G), dims, np.zeros((0, n)))
    noise = (_capi.DeviceBuffer(8 << 20), _capi.DeviceBuffer(8 << 20))
    W = synth.random_scaling(dims, seed=3, spread=1.0)
    Pl = _Sp(sp.tril(P))
    bx, bz = rng.standard_normal(n), rng.standard_normal(2 * n)
    ref = None
    try:
        for it in range(max(20, ITERS // 3)):
            _noise(L, rng, noise)
            solve = f(W, Pl)
            x, y, z = bx.copy(), np.zeros(0), bz.copy()
            solve(x, y, z)
            assert np.all(np.isfinite(x)), it
            if ref is None:
                ref = (x, z)
                r = P @ x + G.T @ ((W['di'] ** 2) * (G @ x)) - (bx + G.T @ ((W['di'] ** 2) * bz))   # reduced system S x = bx + G'D^2 bz
                assert np.max(np.abs(r)) <= 1e-9 * max(1.0, np.max(np.abs(bx))) * 1e3
            else:
                assert np.array_equal(x, ref[0]) and np.array_equal(z, ref[1]), it
    finally:
        f.engine.close()
        for b in noise:
            b.free()

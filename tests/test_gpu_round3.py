"""Round-3 additions at the drop-in boundary: a sparse H alternating with no H on a dense-G factory, the pinned
host H released by the next set_H_*, `mi355kkt_set_A_csr` with unsorted / repeated entries and with an invalid rowptr.
Parity against the NumPy oracle (oracle/kkt_oracle.py) and the dense engine."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from cvxopt_amd import _capi, kkt, synth
from helpers import relerr
from oracle import kkt_oracle as ko

pytestmark = pytest.mark.gpu


class RawSp(object):
    """stand-in for cvxopt.spmatrix that hands over the CCS arrays exactly as given (unsorted rows, repeated entries)"""

    def __init__(self, shape, colptr, rowind, values):
        self.size = shape
        self.CCS = (np.asarray(colptr, dtype=np.int64), np.asarray(rowind, dtype=np.int64), np.asarray(values, dtype=float))


def _sorted_sp(A):
    A = sp.csc_matrix(A)
    A.sort_indices()
    return RawSp(A.shape, A.indptr, A.indices, A.data)


def test_sparse_H_then_no_H_then_the_same_sparse_H_on_a_dense_G_factory():
    """factor(W, Hsp), factor(W), factor(W, Hsp): the third call used to find the cached tag of the first and skip the
    upload although the second call had dropped H from the device (S = G'D^2 G without H, silently)."""
    n, m = 160, 240
    pr = synth.dense_qp(n, m, seed=2)
    Hd = np.tril(pr['P'])
    Hd[np.abs(Hd) < 0.05] = 0.0
    Hd[np.arange(n), np.arange(n)] = np.abs(Hd[np.arange(n), np.arange(n)]) + 1.0
    Hsp = _sorted_sp(sp.csc_matrix(Hd))
    Hfull = Hd + np.tril(Hd, -1).T
    W = synth.random_scaling(pr['dims'], seed=3, spread=1.0)
    rng = np.random.default_rng(1)
    bx, bz = rng.standard_normal(n), rng.standard_normal(m)
    f = kkt.kkt_chol2(pr['G'], pr['dims'], np.zeros((0, n)))
    oracle = ko.KktChol2(pr['G'], pr['dims'], np.zeros((0, n)))
    for step, H in enumerate((Hsp, None, Hsp, None, Hsp)):
        x, z = bx.copy(), bz.copy()
        f(W, H)(x, np.zeros(0), z)
        xo, zo = bx.copy(), bz.copy()
        oracle.factor(W, None if H is None else Hfull)(xo, np.zeros(0), zo)
        assert relerr(x, xo) < 1e-9 and relerr(z, zo) < 1e-9, step
    f.engine.close()


def test_pinned_host_H_is_released_by_the_next_set_H():
    """async dense H (pinned in place) -> H = None -> a NEW buffer of the same size: the old registration must be gone
    (header: 'alive until the next set_H_*'), the new contents must arrive.  (n = 768: an H below 4 MB is uploaded synchronously
    and never pinned.)"""
    n, m = 768, 800
    pr = synth.dense_qp(n, m, seed=4)
    W = synth.random_scaling(pr['dims'], seed=5, spread=1.0)
    rng = np.random.default_rng(2)
    bx, bz = rng.standard_normal(n), rng.standard_normal(m)
    f = kkt.kkt_chol2(pr['G'], pr['dims'], np.zeros((0, n)))
    oracle = ko.KktChol2(pr['G'], pr['dims'], np.zeros((0, n)))
    for step in range(4):
        P = np.asfortranarray(pr['P'] + step * np.eye(n))          # a fresh allocation every round
        x, z = bx.copy(), bz.copy()
        f(W, P)(x, np.zeros(0), z)
        xo, zo = bx.copy(), bz.copy()
        oracle.factor(W, P)(xo, np.zeros(0), zo)
        assert relerr(x, xo) < 1e-9, step
        del P
        x, z = bx.copy(), bz.copy()
        f(W)(x, np.zeros(0), z)                                     # drops H: the pin of the freed buffer goes with it
        xo, zo = bx.copy(), bz.copy()
        oracle.factor(W, None)(xo, np.zeros(0), zo)
        assert relerr(x, xo) < 1e-9, step
    f.engine.close()


def _box_problem(k=7, p=9, seed=0):
    n = k ** 3
    P = synth.grid_laplacian(k)
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()
    rng = np.random.default_rng(seed)
    return n, P, G, rng


def test_set_A_csr_sums_repeated_entries_and_accepts_unsorted_columns():
    """A given with repeated (row, column) triplets and columns in arbitrary order: the sparse engine must use the same
    matrix as the dense upload (np.add.at), not race on the repeated entries."""
    p = 9
    n, P, G, rng = _box_problem(p=p)
    # triplets with repeats: every row gets 4 entries, two of them on the same column
    rows = np.repeat(np.arange(p), 4)
    cols = rng.integers(0, n, size=4 * p)
    cols[1::4] = cols[0::4]                                         # a repeated column in every row
    vals = rng.standard_normal(4 * p)
    Adense = np.zeros((p, n))
    np.add.at(Adense, (rows, cols), vals)
    Adense[np.arange(p), np.arange(p)] += 1.0                       # full row rank
    rows = np.concatenate([rows, np.arange(p)])
    cols = np.concatenate([cols, np.arange(p)])
    vals = np.concatenate([vals, np.ones(p)])
    # CSC arrays by a stable sort on the column only: rows inside a column unsorted, repeats kept
    perm = rng.permutation(rows.size)
    rows, cols, vals = rows[perm], cols[perm], vals[perm]
    order = np.argsort(cols, kind='stable')
    colptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(colptr, cols + 1, 1)
    A_raw = RawSp((p, n), np.cumsum(colptr), rows[order], vals[order])
    dims = {'l': 2 * n, 'q': [], 's': []}
    W = synth.random_scaling(dims, seed=4, spread=1.0)
    bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(2 * n)
    fs = kkt.kkt_chol2(_sorted_sp(G), dims, A_raw)
    x, y, z = bx.copy(), by.copy(), bz.copy()
    fs(W, _sorted_sp(sp.tril(P)))(x, y, z)
    assert fs.engine._mode == "sparse"
    fd = kkt.kkt_chol2(np.asfortranarray(G.toarray()), dims, np.asfortranarray(Adense))
    xd, yd, zd = bx.copy(), by.copy(), bz.copy()
    fd(W, np.asfortranarray(P.toarray()))(xd, yd, zd)
    assert relerr(x, xd) < 1e-8 and relerr(y, yd) < 1e-7 and relerr(z, zd) < 1e-8
    v = rng.standard_normal(n)
    assert relerr(fs.engine.product(1, False, v), Adense @ v) < 1e-12
    u = rng.standard_normal(p)
    assert relerr(fs.engine.product(1, True, u), Adense.T @ u) < 1e-12
    fs.engine.close()
    fd.engine.close()


def test_set_A_csr_rejects_an_invalid_rowptr():
    L = _capi.lib()
    n, p = 12, 3
    h = C.c_void_p()
    _capi.check(L.mi355kkt_create(C.byref(h), 0, _capi.CHOL2, n, p, 4, 0, None, 0, None), "create")
    try:
        ci = np.array([0, 1, 2, 3], dtype=np.int64)
        v = np.ones(4)
        as_i = lambda a: a.ctypes.data_as(_capi.c_i64_p)
        as_d = lambda a: a.ctypes.data_as(_capi.c_double_p)
        for rp in ([1, 2, 3, 4], [0, 3, 2, 4]):                    # rowptr[0] != 0; not monotone
            rp = np.array(rp, dtype=np.int64)
            assert L.mi355kkt_set_A_csr(h, as_i(rp), as_i(ci), as_d(v)) == _capi.EINVAL
        rp = np.array([0, 1, 2, 4], dtype=np.int64)
        bad = np.array([0, 1, 2, n], dtype=np.int64)               # column index out of range
        assert L.mi355kkt_set_A_csr(h, as_i(rp), as_i(bad), as_d(v)) == _capi.EINVAL
        assert L.mi355kkt_set_A_csr(h, as_i(rp), as_i(ci), as_d(v)) == 0
    finally:
        L.mi355kkt_destroy(h)


@pytest.mark.parametrize("spread,p", [(3.0, 0), (5.0, 0), (5.0, 12)])
def test_ldl_flavour_at_late_iteration_scalings_against_the_reference_kkt_ldl(ref_cvxopt, spread, p):
    """kind='ldl' is what users pick for robustness on ill-conditioned W.  Late interior-point iterations: d spans
    10^-spread .. 10^+spread, cond(K) up to ~1e12 for the 3 x 3 system.  The device engine factors the reduced quasi-definite
    form and adds two steps of iterative refinement against the 3 x 3 system (without them the residual was 3e-9 at spread 3);
    the reference (misc.kkt_ldl, misc.py:1085-1121) runs a pivoted LDL' (sytrf) of the whole 3 x 3 matrix.  Both residuals of
    the ORIGINAL 3 x 3 system are recorded in the parity report; ours must stay within a factor of three of the reference's."""
    from cvxopt import matrix, misc
    from helpers import record
    n, m = 300, 500
    pr = synth.dense_qp(n, m, seed=21, p=p)
    G, P, dims = pr['G'], pr['P'], pr['dims']
    A = pr.get('A', np.zeros((0, n)))
    W = synth.random_scaling(dims, seed=8, spread=spread)
    rng = np.random.default_rng(3)
    bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    f = kkt.kkt_ldl(G, dims, A)
    x, y, z = bx.copy(), by.copy(), bz.copy()
    f(W, P)(x, y, z)
    f.engine.close()
    Wr = {'d': matrix(W['d']), 'di': matrix(W['di']), 'v': [], 'beta': [], 'r': [], 'rti': []}
    fr = misc.kkt_ldl(matrix(G), dims, matrix(A) if p else matrix(0.0, (0, n)))
    xr, yr, zr = matrix(bx), matrix(by) if p else matrix(0.0, (0, 1)), matrix(bz)
    fr(Wr, matrix(P))(xr, yr, zr)
    xr, yr, zr = np.array(xr).ravel(), np.array(yr).ravel(), np.array(zr).ravel()
    res = ko.kkt_residual(P, A, G, W, dims, bx, by, bz, x, y, z)
    res_ref = ko.kkt_residual(P, A, G, W, dims, bx, by, bz, xr, yr, zr)
    ex, ez = relerr(x, xr), relerr(z, zr)
    record("ldl_late_iteration_spread%g_p%d" % (spread, p), residual_device=res, residual_reference_kkt_ldl=res_ref,
           x_relerr_vs_reference=ex, z_relerr_vs_reference=ez, d_min=float(W['d'].min()), d_max=float(W['d'].max()))
    # achieved (profiles/r03_parity_report.json): 2e-15 .. 5e-14 against the reference's 2e-13 .. 3e-13; x agrees to 3e-14 .. 1e-13
    assert res <= max(3e-13, 3.0 * res_ref), (res, res_ref)
    assert ex < 1e-10, ex


@pytest.mark.parametrize("B,n,m", [(3, 64, 128), (2, 100, 1030), (4, 512, 1024), (2, 70, 2300), (1, 130, 61), (2, 33, 7), (1, 1, 1)])
def test_batch_residual_products_in_one_pass_over_G(B, n, m):
    """device-resident callers of mi355kkt_batch_products (the batched coneqp loop) get G x and G' z from ONE pass over G
    (gemv_nt_fused_kernel): ragged row blocks (m not a multiple of 2 / 128 / 1024, several row blocks), column counts that are
    not multiples of the 64 columns of a workgroup; against NumPy and against the two-pass host-pointer path of the same call."""
    from cvxopt_amd.batch import BatchKkt
    rng = np.random.default_rng(B * 1000 + n + m)
    Gt = rng.standard_normal((B, n, m))                                 # Gt[b] = G_b' (row-major n x m == column-major m x n)
    Bm = rng.standard_normal((B, n, n))
    P = np.einsum('bij,bkj->bik', Bm, Bm)
    k = BatchKkt(Gt, P)
    x, z = rng.standard_normal((B, n)), rng.standard_normal((B, m))
    dx, dz = _capi.DeviceBuffer.from_array(x), _capi.DeviceBuffer.from_array(z)
    dGx, dGTz, dPx = _capi.DeviceBuffer(8 * B * m), _capi.DeviceBuffer(8 * B * n), _capi.DeviceBuffer(8 * B * n)
    _capi.check(_capi.lib().mi355kkt_batch_products(k.h, dx.ptr, dz.ptr, dGx.ptr, dGTz.ptr, dPx.ptr, 1), "batch_products")
    _capi.lib().mi355kkt_device_synchronize()
    Gx = dGx.to_array((B, m), order="C")
    GTz = dGTz.to_array((B, n), order="C")
    Px = dPx.to_array((B, n), order="C")
    Gx_ref = np.einsum('bnm,bn->bm', Gt, x)
    GTz_ref = np.einsum('bnm,bm->bn', Gt, z)
    sc_n = np.einsum('bnm,bn->bm', np.abs(Gt), np.abs(x)) + 1e-300
    sc_t = np.einsum('bnm,bm->bn', np.abs(Gt), np.abs(z)) + 1e-300
    assert np.max(np.abs(Gx - Gx_ref) / sc_n) < 2e-15 * max(8, n)
    assert np.max(np.abs(GTz - GTz_ref) / sc_t) < 2e-15 * max(8, m)
    Gx2, GTz2, Px2 = k.products(x, z)                                   # host pointers: the two-pass path
    assert relerr(Gx, Gx2) < 1e-13 and relerr(GTz, GTz2) < 1e-13 and relerr(Px, Px2) < 1e-14
    k.close()


def test_ldl_refinement_with_many_equalities_and_few_cone_rows():
    """n > 256, p large, few inequality rows: the product by - A ux of the refinement step needs ceil(n / 256) * p doubles of GEMV
    workspace -- more than the products with G and with L^-1 A' the buffer was sized for (found by review: 400 > 300 here)."""
    n, m, p = 300, 10, 200
    pr = synth.dense_qp(n, m, seed=77, p=p)
    W = synth.random_scaling(pr['dims'], seed=9, spread=1.0)
    rng = np.random.default_rng(4)
    bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    oracle = ko.KktChol2(pr['G'], pr['dims'], pr['A'])
    xo, yo, zo = bx.copy(), by.copy(), bz.copy()
    oracle.factor(W, pr['P'])(xo, yo, zo)
    for kind in (kkt.kkt_ldl, kkt.kkt_ldl2, kkt.kkt_chol):
        f = kkt_f = kind(pr['G'], pr['dims'], pr['A'])
        for _ in range(2):
            x, y, z = bx.copy(), by.copy(), bz.copy()
            f(W, pr['P'])(x, y, z)
            assert relerr(x, xo) < 1e-9 and relerr(y, yo) < 1e-8 and relerr(z, zo) < 1e-9
        kkt_f.engine.close()

"""Round 6 parity corners (VERDICT r5 "next round" item 5).

(a) ArithmeticError(info): the reference's lapack.potrf raises ArithmeticError with LAPACK's `info` -- the order of the first
    leading minor that is not positive definite (/root/reference/src/C/lapack.c:32-34, :1508-1521) -- and kkt_chol2 lets it
    through (misc.py:1440-1447).  The device Cholesky promises the same 1-based column; here the VALUE is compared with the real
    reference's, for a first bad pivot at the start, on both sides of a 128-column tile boundary, deep inside a large matrix and
    in a ragged last tile, through the raw kernel entry point, through the hook, per problem of a batch, and (as the permuted
    column of the engine's own fill-reducing order) inside a sparse front.
(b) the 512-row all-CU triangular solves (trsv512.hip) against the round-4 pair kernel and the oracle."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from cvxopt_amd import _capi, kkt, synth
from helpers import record, relerr
from oracle import kkt_oracle as ko

pytestmark = pytest.mark.gpu


def indefinite_at(n, k, seed=0):
    """symmetric A = L D L' with D = +1 except D[k-1] = -1, L unit-ish lower triangular and well conditioned: the leading minors of
    order < k are positive definite with pivots ~1, the k-th pivot is ~ -1 (nothing near rounding)"""
    rng = np.random.default_rng(seed)
    L = np.tril(rng.standard_normal((n, n)) / np.sqrt(n), -1) + np.diag(1.0 + rng.random(n))
    d = np.ones(n)
    if k:
        d[k - 1] = -1.0                                         # (k = 0: positive definite)
    A = (L * d) @ L.T
    return np.asfortranarray(0.5 * (A + A.T))


def ref_info(cvx, A):
    """info of the reference's lapack.potrf (0: positive definite)"""
    M = cvx.matrix(np.asfortranarray(A))
    try:
        cvx.lapack.potrf(M)
    except ArithmeticError as e:
        return int(e.args[0])
    return 0


def dev_potrf_info(A):
    n = A.shape[0]
    L = _capi.lib()
    dA = _capi.DeviceBuffer.from_array(np.asfortranarray(A))
    ms, info = C.c_float(), C.c_int()
    rc = L.mi355kkt_op_potrf(dA.ptr, n, n, C.byref(info), C.byref(ms))
    assert rc == 0 or rc == info.value, (rc, info.value)
    return info.value


@pytest.mark.parametrize("n,k", [(200, 1), (200, 127), (200, 128), (200, 129), (200, 200), (1600, 1), (1600, 127), (1600, 128),
                                 (1600, 129), (1600, 1500), (1600, 1537), (1600, 1600), (2048, 1920), (2048, 2048)])
def test_potrf_info_is_the_references_first_bad_minor(ref_cvxopt, n, k):
    """n = 200: the launch chain (two panels, the second ragged); n = 1600: the persistent tile kernel, ragged last tile
    (1537 = its first column, 1600 = its last); n = 2048: whole tiles"""
    A = indefinite_at(n, k, seed=n + k)
    want = ref_info(ref_cvxopt, A)
    assert want == k                                            # the construction does what it says
    assert dev_potrf_info(A) == want


@pytest.mark.parametrize("n,k", [(300, 129), (1600, 128), (1600, 1500)])
def test_hook_raises_arithmetic_error_with_the_references_info(ref_cvxopt, n, k):
    """through kkt_chol2's factor(W, H): S = H + G'W^-2 G with G = 0 rows, so S = H exactly"""
    m = 4
    H = indefinite_at(n, k, seed=7 * n + k)
    G = np.asfortranarray(np.zeros((m, n)))
    dims = {'l': m, 'q': [], 's': []}
    W = synth.random_scaling(dims, seed=0)
    f = kkt.kkt_chol2(G, dims, np.zeros((0, n)))
    try:
        with pytest.raises(ArithmeticError) as ei:
            f(W, H)
    finally:
        f.engine.close()
    assert int(ei.value.args[0]) == ref_info(ref_cvxopt, H) == k
    # and the oracle (the restated kkt_chol2) raises the same
    with pytest.raises(ArithmeticError) as eo:
        ko.KktChol2(G, dims, np.zeros((0, n))).factor(W, H)
    assert int(eo.value.args[0]) == k


def test_batched_info_per_problem_is_the_references(ref_cvxopt):
    """mi355kkt_batch_factor: info[b] of every problem equals the reference's lapack.potrf on that problem's S"""
    from cvxopt_amd.batch import BatchKkt
    B, n, m = 6, 270, 3
    ks = [0, 1, 128, 129, 257, 270]                             # 0: positive definite
    P = np.zeros((B, n, n))
    for b, k in enumerate(ks):
        P[b] = indefinite_at(n, k, seed=b)
    Gt = np.zeros((B, n, m))
    g = BatchKkt(Gt, P)
    try:
        info = g.factor(np.ones((B, m)))
    finally:
        g.close()
    want = [ref_info(ref_cvxopt, P[b]) for b in range(B)]
    assert want == ks
    assert [int(v) for v in info] == want


def test_sparse_info_is_the_first_bad_column_of_the_engines_own_order():
    """The reference's cholmod.numeric raises ArithmeticError(k) with CHOLMOD's L->minor: a column of ITS fill-reducing order
    (/root/reference/src/C/cholmod.c:428-438).  The orders differ, so the value that can be pinned is: the first non-positive pivot
    of a sequential Cholesky of S in the engine's own order (csrc/ordering.cpp via mi355kkt_op_symbolic), 1-based -- inside a
    front of the multifrontal factorisation, fronts of independent subtrees notwithstanding."""
    from test_gpu_sparse import FakeSp
    from test_sparse_symbolic_cpu import analyse, grid
    import scipy.linalg as sla
    nx = 24
    n = nx * nx
    P = grid(nx, nx).tolil()
    bad = [5, 300, 431]                                         # three indefinite diagonal entries: the smallest permuted one counts
    for i in bad:
        P[i, i] = -50.0
    P = sp.csc_matrix(P)
    m = 2 * n
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc() * 1e-3
    dims = {'l': m, 'q': [], 's': []}
    W = {'di': np.ones(m), 'd': np.ones(m), 'v': [], 'beta': [], 'r': [], 'rti': [], 'dnl': np.zeros(0), 'dnli': np.zeros(0)}
    f = kkt.kkt_chol2(FakeSp(G), dims, np.zeros((0, n)))
    try:
        with pytest.raises(ArithmeticError) as ei:
            f(W, FakeSp(sp.tril(P)))
        assert f.engine._mode == "sparse"
    finally:
        f.engine.close()
    perm, _, _, _ = analyse(G, P)
    S = (P + G.T @ G).toarray()[np.ix_(perm, perm)]
    _, info = sla.lapack.dpotrf(np.asfortranarray(S), lower=1)
    assert info > 0 and int(ei.value.args[0]) == info

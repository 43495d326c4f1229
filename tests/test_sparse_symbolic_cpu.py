"""CPU test of the host-side symbolic analysis of the sparse engine (no GPU needed): the ordering is a
permutation, the supernodal structure covers the true fill, and nested dissection beats the natural ordering."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp


def analyse(G, H):
    from cvxopt_amd import _capi
    L = _capi.lib()
    G = sp.csc_matrix(G); G.sort_indices()
    H = sp.csc_matrix(sp.tril(H)); H.sort_indices()
    m, n = G.shape
    perm = np.zeros(n, dtype=np.int32)
    nnzL, ns, nl = C.c_int64(), C.c_int(), C.c_int()
    i64 = lambda a: np.ascontiguousarray(a, dtype=np.int64).ctypes.data_as(_capi.c_i64_p)
    keep = [np.ascontiguousarray(a, dtype=np.int64) for a in (G.indptr, G.indices, H.indptr, H.indices)]
    rc = L.mi355kkt_op_symbolic(n, m, keep[0].ctypes.data_as(_capi.c_i64_p), keep[1].ctypes.data_as(_capi.c_i64_p),
                                keep[2].ctypes.data_as(_capi.c_i64_p), keep[3].ctypes.data_as(_capi.c_i64_p),
                                perm.ctypes.data_as(_capi.c_int_p), C.byref(nnzL), C.byref(ns), C.byref(nl))
    _capi.check(rc, "op_symbolic")
    return perm, nnzL.value, ns.value, nl.value


def exact_fill(S, perm):
    """nnz(L) of the Cholesky factor of S[perm, perm] by symbolic elimination (structure only)."""
    n = S.shape[0]
    A = sp.csc_matrix(S)[perm][:, perm].tocsc()
    cols = [set(int(i) for i in A.indices[A.indptr[j]:A.indptr[j + 1]] if i > j) for j in range(n)]
    nnz = n
    for j in range(n):
        s = cols[j]
        nnz += len(s)
        if s:
            p = min(s)                       # parent in the elimination tree inherits the rest of the structure
            cols[p] |= (s - {p})
    return nnz


def grid(nx, ny):
    ex, ey = np.ones(nx), np.ones(ny)
    Tx = sp.diags([-ex[:-1], 2 * ex, -ex[:-1]], [-1, 0, 1])
    Ty = sp.diags([-ey[:-1], 2 * ey, -ey[:-1]], [-1, 0, 1])
    return (sp.kron(sp.eye(ny), Tx) + sp.kron(Ty, sp.eye(nx))).tocsc()


@pytest.mark.parametrize("nx,ny", [(1, 1), (6, 4), (25, 25), (40, 13)])
def test_ordering_is_a_permutation_and_structure_covers_fill(nx, ny):
    P = grid(nx, ny)
    n = nx * ny
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()
    perm, nnzL, ns, nl = analyse(G, P)
    assert sorted(perm.tolist()) == list(range(n))
    S = (P + G.T @ G).tocsc()
    fill = exact_fill(S, perm)
    assert nnzL >= fill                      # supernodal panels contain every structural nonzero of L
    assert nnzL <= 2.0 * fill + 64           # ... with bounded padding from amalgamation
    assert 1 <= ns <= n and 1 <= nl <= ns


def test_nested_dissection_beats_natural_order_on_a_grid():
    P = grid(30, 30)
    n = 900
    G = sp.eye(n, format='csc')
    perm, nnzL, ns, nl = analyse(G, P)
    S = (P + G.T @ G).tocsc()
    assert exact_fill(S, perm) < 0.7 * exact_fill(S, np.arange(n))


def test_general_G_couplings_enter_the_pattern():
    rng = np.random.default_rng(0)
    n, m = 60, 90
    G = sp.random(m, n, density=0.05, random_state=1, format='csc') + sp.vstack([sp.eye(n), sp.csc_matrix((m - n, n))])
    P = sp.csc_matrix((n, n))
    perm, nnzL, ns, nl = analyse(G, P)
    S = (G.T @ G).tocsc()
    assert nnzL >= exact_fill(S, perm)

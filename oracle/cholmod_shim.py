"""SciPy-backed stand-in for the reference's `cvxopt.cholmod` extension module.

TEST INFRASTRUCTURE ONLY (installed as oracle/_ref/cvxopt/cholmod.py by oracle/build_ref.sh).

Why: `cvxopt.misc` imports `cholmod` unconditionally (reference src/python/misc.py:21) and the
sparse branch of `misc.kkt_chol2` (misc.py:1405-1487, 1528-1558) calls
`cholmod.symbolic / numeric / solve / spsolve`.  The real module (reference src/C/cholmod.c) binds
SuiteSparse CHOLMOD, a third-party dependency that is NOT vendored under /root/reference (CI pins
SuiteSparse v7.11.0, .github/workflows/linux_build.yml:12) and is not installed in this image.

What is restated here is the *interface contract* of src/C/cholmod.c, not CHOLMOD's algorithm:
  symbolic(A, p=None, uplo='L') -> opaque factor handle        (cholmod.c:273-333)
  numeric(A, F)                 -> numeric LL^T of tril(A); ArithmeticError if not PD
                                                                 (cholmod.c:364-448, :426-429)
  solve(F, B, sys=0)            -> in place on dense B; sys codes (cholmod.c:491-493):
        0: A x = b   4: L x = b   5: L^T x = b   7: x = P b   8: x = P^T b   (1,2,3,6: LDL^T forms)
  spsolve(F, B, sys=0)          -> new spmatrix                 (cholmod.c:583-654)
  linsolve / splinsolve / diag  -> convenience wrappers         (cholmod.c:685, :837, :969)
Because a Cholesky factorisation of an SPD matrix is unique for a given ordering, every product of
these calls that the kktsolver consumes (solutions of S x = b, K = A S^-1 A^T) is independent of
the fill-reducing ordering; only factor *entries* are ordering dependent and those are never
compared.  Small systems use a dense LAPACK Cholesky with the identity ordering; large ones use
SuperLU in symmetric mode (P S P^T = L U, U = D L^T  =>  Cholesky factor C = L sqrt(D)).
"""
import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from cvxopt.base import matrix, spmatrix, sparse

options = {}            # mirrors cholmod.options (cholmod.c:86-152); accepted and ignored
_DENSE_LIMIT = 3000


class _Factor(object):
    def __init__(self, n, uplo):
        self.n, self.uplo = n, uplo
        self.kind = None        # 'dense' | 'slu'
        self.C = None           # dense lower Cholesky factor
        self.L = None           # csr unit-lower factor (slu)
        self.sd = None          # sqrt(D)
        self.perm = None        # P: (P b)[i] = b[perm[i]]


def _tocsc(A):
    if isinstance(A, spmatrix):
        cp, ri, v = A.CCS
        return sp.csc_matrix((np.array(v, dtype=float).ravel(),
                              np.array(ri, dtype=np.int64).ravel(),
                              np.array(cp, dtype=np.int64).ravel()), shape=A.size)
    return sp.csc_matrix(np.array(A))


def _sym_from_tri(A, uplo):
    S = _tocsc(A)
    T = sp.tril(S, format='csc') if uplo == 'L' else sp.triu(S, format='csc')
    D = sp.diags(T.diagonal())
    return (T + T.T - D).tocsc()


def symbolic(A, p=None, uplo='L'):
    if A.size[0] != A.size[1]:
        raise TypeError("A is not a square sparse matrix")
    return _Factor(A.size[0], uplo)


def numeric(A, F):
    n = F.n
    if n == 0:
        F.kind, F.C = 'dense', np.zeros((0, 0))
        return
    S = _sym_from_tri(A, F.uplo)
    if n <= _DENSE_LIMIT:
        try:
            F.C = np.linalg.cholesky(S.toarray())
        except np.linalg.LinAlgError:
            raise ArithmeticError(0)
        F.kind, F.perm = 'dense', np.arange(n)
        return
    lu = spla.splu(S, permc_spec='MMD_AT_PLUS_A', diag_pivot_thresh=0.0,
                   options=dict(SymmetricMode=True))
    if not np.array_equal(lu.perm_r, lu.perm_c):
        raise ArithmeticError(0)           # a pivot was rejected: not PD
    d = lu.U.diagonal()
    if not np.all(d > 0.0):
        raise ArithmeticError(int(np.argmax(d <= 0.0)) + 1)
    F.kind = 'slu'
    F.L = lu.L.tocsr()
    F.sd = np.sqrt(d)
    # SuperLU: Pr A Pc = L U with row i of A going to row perm_r[i]  =>  (P b)[perm[i]] = b[i]
    inv = np.empty(n, dtype=np.int64)
    inv[lu.perm_r] = np.arange(n)
    F.perm = inv


def _apply(F, b, sys):
    """b: (n, k) float ndarray, returns new array."""
    if sys == 7:
        return b[F.perm, :]
    if sys == 8:
        out = np.empty_like(b)
        out[F.perm, :] = b
        return out
    if F.kind == 'dense':
        if sys == 4:
            return sla.solve_triangular(F.C, b, lower=True)
        if sys == 5:
            return sla.solve_triangular(F.C.T, b, lower=False)
        if sys == 0:
            return sla.cho_solve((F.C, True), b)
    else:
        if sys == 4:
            return spla.spsolve_triangular(F.L, b, lower=True, unit_diagonal=True) / F.sd[:, None]
        if sys == 5:
            return spla.spsolve_triangular(F.L.T.tocsr(), b / F.sd[:, None], lower=False,
                                           unit_diagonal=True)
        if sys == 0:
            return _apply(F, _apply(F, _apply(F, _apply(F, b, 7), 4), 5), 8)
    raise NotImplementedError("cholmod shim: sys=%d" % sys)


def solve(F, B, sys=0, nrhs=-1, ldB=0, offsetB=0):
    if F.kind is None:
        raise ValueError("called with symbolic factor")
    if F.n == 0 or B.size[1] == 0:
        return
    b = np.asarray(B)                      # zero-copy column-major view of the cvxopt matrix
    b[:, :] = _apply(F, np.array(b, dtype=float, order='F'), sys)


def spsolve(F, B, sys=0):
    if B.size[1] == 0 or F.n == 0:
        return spmatrix([], [], [], B.size, 'd')
    D = matrix(B)
    solve(F, D, sys)
    return sparse(D)


def linsolve(A, B, p=None, uplo='L', nrhs=-1, ldB=0, offsetB=0):
    F = symbolic(A, p, uplo)
    numeric(A, F)
    solve(F, B, 0)


def splinsolve(A, B, p=None, uplo='L'):
    F = symbolic(A, p, uplo)
    numeric(A, F)
    return spsolve(F, B, 0)


def diag(F):
    """Diagonal of the Cholesky factor as an (n,1) matrix (cholmod.c:969-1026)."""
    d = np.diag(F.C) if F.kind == 'dense' else F.sd
    return matrix(np.array(d, dtype=float).reshape(-1, 1))

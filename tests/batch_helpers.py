"""CPU KKT back end for cvxopt_amd.batch.coneqp_batch used by the tests (oracle side: SciPy Cholesky per
problem) -- lets the lock-step loop and the sharding logic be checked without a GPU."""
import numpy as np
import scipy.linalg as sla


class NumpyBatchKkt(object):
    def __init__(self, Gt, P):
        self.Gt, self.P = Gt, P
        self.B, self.n, self.m = Gt.shape

    def factor(self, di):
        info = np.zeros(self.B, dtype=np.int32)
        self.di = di.copy()
        self.L = []
        for b in range(self.B):
            Gs = di[b][:, None] * self.Gt[b].T
            S = Gs.T @ Gs + (np.tril(self.P[b]) + np.tril(self.P[b], -1).T if self.P is not None else 0.0)
            c, i = sla.lapack.dpotrf(S, lower=1)
            info[b] = i
            self.L.append(c)
        return info

    def solve(self, x, z):
        for b in range(self.B):
            G = self.Gt[b].T
            zs = self.di[b] * z[b]
            rhs = x[b] + G.T @ (self.di[b] * zs)
            u = sla.cho_solve((self.L[b], True), rhs)
            x[b] = u
            z[b] = self.di[b] * (G @ u) - zs


def numpy_local_solver(P, q, Gt, h, **opts):
    from cvxopt_amd.batch import coneqp_batch
    return coneqp_batch(P, q, Gt, h, kkt=NumpyBatchKkt(Gt, P), **opts)

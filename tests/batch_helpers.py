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


class TensorEngine(object):
    """Stands in for cvxopt_amd.batch.BatchKkt in the DEVICE-RESIDENT branch of ShardedBatch (engine_factory=): the same surface
    -- constructed with shape=(cnt, n, m), set_problem(G, P) and coneqp(q, h) on torch tensors, result dict of tensors with
    'status_code' / 'lockstep iterations' -- backed by the NumPy lock-step loop, so that the RCCL branch's ordering (scatter list
    construction, async work handles, wait() order, packed gather) is executed over gloo with host tensors."""
    log = []

    def __init__(self, shape=None, device=None):
        self.shape = tuple(shape)
        self.closed = False
        TensorEngine.log.append(("create", self.shape))

    def set_problem(self, Gt, P=None, At=None):
        assert tuple(Gt.shape) == self.shape and Gt.dtype.is_floating_point
        self.Gt = Gt.numpy().copy()
        self.P = None if P is None else P.numpy().copy()
        TensorEngine.log.append(("set_problem", self.shape[0]))

    def coneqp(self, q, h, **opts):
        import torch
        from cvxopt_amd.batch import coneqp_batch
        assert not self.closed
        res = coneqp_batch(self.P, q.numpy().copy(), self.Gt, h.numpy().copy(), kkt=NumpyBatchKkt(self.Gt, self.P), **opts)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64))
        code = np.where(np.asarray(res['status']) == 'optimal', 1, 2).astype(np.int32)
        TensorEngine.log.append(("coneqp", self.shape[0]))
        return {'x': t(res['x']), 's': t(res['s']), 'z': t(res['z']), 'y': t(np.zeros((self.shape[0], 0))),
                'status_code': torch.from_numpy(code), 'iterations': torch.from_numpy(np.asarray(res['iterations'], dtype=np.int32)),
                'primal objective': t(res['primal objective']), 'dual objective': t(res['dual objective']), 'gap': t(res['gap']),
                'lockstep iterations': int(np.max(res['iterations']))}

    def close(self):
        self.closed = True

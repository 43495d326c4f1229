"""Handle life cycle: repeated create / use / destroy of every kind of handle must not leak HBM or leave the device in a
bad state (the reference's factories are created once per solver call; a long-running service creates thousands)."""
import numpy as np
import pytest
import scipy.sparse as sp

import cvxopt_amd
from cvxopt_amd import kkt, synth
from cvxopt_amd.batch import BatchKkt, pack_problems

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def test_no_hbm_leak_over_many_handles():
    pr = synth.dense_qp(96, 200, seed=0, p=5)
    so = synth.socp(n=24, ncones=6, r=5, seed=1, ml=4)
    probs = [synth.dense_qp(32, 64, seed=i) for i in range(4)]
    Pb, qb, Gtb, hb = pack_problems(probs)

    class Sp(object):
        def __init__(self, A):
            A = sp.csc_matrix(A); A.sort_indices()
            self.size = A.shape
            self.CCS = (A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64))
    n = 100
    Ps = sp.diags([-np.ones(n - 1), 2.1 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1]).tocsc()
    Gs = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()

    def cycle():
        sol = cvxopt_amd.coneqp_lp(pr['P'], pr['q'], pr['G'], pr['h'], A=pr['A'], b=pr['b'])
        assert sol['status'] == 'optimal'
        assert cvxopt_amd.conelp_device(so['c'], so['G'], so['h'], so['dims'])['status'] == 'optimal'
        assert cvxopt_amd.coneqp_device(np.eye(24), so['c'], so['G'], so['h'], so['dims'])['status'] == 'optimal'
        assert cvxopt_amd.coneqp_lp(Sp(sp.tril(Ps)), -np.ones(n), Sp(Gs), np.ones(2 * n))['status'] == 'optimal'
        k = BatchKkt(Gtb, Pb)
        assert all(s == 'optimal' for s in k.coneqp(qb, hb)['status'])
        k.close()
        f = kkt.kkt_ldl(pr['G'], pr['dims'], pr['A'], kktreg=1e-8)
        W = synth.random_scaling(pr['dims'], seed=3, spread=1.0)
        x, y, z = np.ones(96), np.ones(5), np.ones(200)
        f(W, pr['P'])(x, y, z)
        f.engine.close()
    for _ in range(3):
        cycle()                       # warm-up: library-level one-time allocations
    before = _free_bytes()
    for _ in range(25):
        cycle()
    after = _free_bytes()
    assert before - after < 64 * 1024 * 1024, (before, after)

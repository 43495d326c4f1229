"""Config-4 class at the sizes SURVEY 8(d) names (n = 1e5 .. 1e6): the supernodal engine's factor + solve on the 64^3 and 100^3
box-QPs against reference solutions computed independently of any code of ours (tests/golden/make_golden_sparse_big.py: SuperLU at
64^3, preconditioned CG at 100^3 -- the reference's own sparse kkt_chol2 branch with the SuperLU shim is too slow there, > 200 s
per factorisation at 64^3) and through the KKT residual of the FULL solution (VERDICT r3 "weak" 3 / item 5)."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from cvxopt_amd import kkt
from helpers import record

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _Sp(object):
    def __init__(self, A):
        A = sp.csc_matrix(A)
        A.sort_indices()
        self.size = A.shape
        self.CCS = (A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64))


@pytest.mark.parametrize("k", [64, 100])
def test_sparse_engine_at_survey_sizes(k):
    import make_golden_sparse_big as gen
    path = os.path.join(GOLD, "sparse%d.npz" % k)
    if not os.path.exists(path):
        pytest.fail("tests/golden/sparse%d.npz missing: python tests/golden/make_golden_sparse_big.py %d" % (k, k))
    g = np.load(path, allow_pickle=False)
    P, W, bx, bz, S, rhs = gen.problem(k, seed=int(g['seed']))
    n = k ** 3
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()
    dims = {'l': 2 * n, 'q': [], 's': []}
    f = kkt.kkt_chol2(_Sp(G), dims, np.zeros((0, n)))
    try:
        x, y, z = bx.copy(), np.zeros(0), bz.copy()
        f(W, _Sp(sp.tril(P)))(x, y, z)
        assert f.engine._mode == "sparse"
        st = f.engine.sparse_stats()
    finally:
        f.engine.close()
    # (i) the reduced system S x = bx + G' D^2 bz, every entry of x
    res = float(np.linalg.norm(S @ x - rhs) / np.linalg.norm(rhs))
    # (ii) the independent reference solution, sampled
    ex = float(np.max(np.abs(x[::64] - g['x_sample'])) / np.max(np.abs(g['x_sample'])))
    en = abs(float(np.linalg.norm(x)) - float(g['x_norm'])) / float(g['x_norm'])
    # (iii) z = W^-T (G x - bz) as the hook returns it (misc.py:1563), from OUR x: the third block row of the KKT system
    di = W['di']
    ez = float(np.max(np.abs(z - di * (G @ x - bz))) / max(1.0, np.max(np.abs(z))))
    record("config4_sparse_%d" % k, n=n, nnzL=st['nnzL'], supernodes=st['supernodes'], levels=st['levels'], residual=res,
           x_sample_relerr=ex, x_norm_relerr=en, z_consistency=ez, reference_residual=float(g['residual']))
    assert res <= 1e-12, res
    assert ex <= max(1e-9, 100.0 * float(g['residual'])) and en <= 1e-10, (ex, en)
    assert ez <= 1e-12, ez

"""Developer probe: sparse path timing on a 2-D / 3-D Laplacian box-QP (stand-in for ssget 1288-class)."""
import os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import kkt, synth
from test_gpu_sparse import FakeSp, laplace2d, laplace3d, box
import scipy.sparse.linalg as spla

for name, P in (("2d 100x100", laplace2d(100, 100)), ("2d 316x316", laplace2d(316, 316)), ("3d 30^3", laplace3d(30)), ("3d 46^3", laplace3d(46))):
    n = P.shape[0]
    G = box(n)
    dims = {'l': 2 * n, 'q': [], 's': []}
    t = time.perf_counter()
    f = kkt.kkt_chol2(FakeSp(G), dims, np.zeros((0, n)))
    W = synth.random_scaling(dims, seed=0, spread=1.0)
    Pl = FakeSp(sp.tril(P))
    s = f(W, Pl)
    t_first = time.perf_counter() - t
    st = f.engine.sparse_stats()
    ts, tf = [], []
    rng = np.random.default_rng(0)
    for r in range(3):
        t = time.perf_counter(); s = f(W, Pl); tf.append(time.perf_counter() - t)
        x, z = rng.standard_normal(n), rng.standard_normal(2 * n)
        bx, bz = x.copy(), z.copy()
        t = time.perf_counter(); s(x, np.zeros(0), z); ts.append(time.perf_counter() - t)
    S = (P + G.T @ sp.diags(W['di'] ** 2) @ G).tocsc()
    rhs = bx + G.T @ (W['di'] ** 2 * bz)
    res = np.linalg.norm(S @ x - rhs) / np.linalg.norm(rhs)
    t = time.perf_counter(); lu = spla.splu(S, permc_spec='MMD_AT_PLUS_A', diag_pivot_thresh=0.0, options=dict(SymmetricMode=True)); t_slu = time.perf_counter() - t
    print("%s n=%d: analyse+first factor %.2fs; factor %.2f ms, solve %.2f ms; nnzL %.2e (SuperLU L nnz %.2e), supernodes %d, levels %d, flops~%.2e, resid %.1e; SuperLU factor on CPU %.1f ms" % (
        name, n, t_first, min(tf) * 1e3, min(ts) * 1e3, st['nnzL'], lu.L.nnz, st['supernodes'], st['levels'], st['flops'], res, t_slu * 1e3))
    f.engine.close()

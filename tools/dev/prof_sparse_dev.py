"""Developer probe for rocprofv3: a few sparse factor/solve calls on one Laplacian box-QP."""
import os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import kkt, synth
from test_gpu_sparse import FakeSp, laplace2d, laplace3d, box

which = sys.argv[1] if len(sys.argv) > 1 else "3d"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 46
P = laplace3d(k) if which == "3d" else laplace2d(k, k)
n = P.shape[0]
G = box(n)
dims = {'l': 2 * n, 'q': [], 's': []}
f = kkt.kkt_chol2(FakeSp(G), dims, np.zeros((0, n)))
W = synth.random_scaling(dims, seed=0, spread=1.0)
Pl = FakeSp(sp.tril(P))
s = f(W, Pl)
print(f.engine.sparse_stats())
rng = np.random.default_rng(0)
for r in range(3):
    t = time.perf_counter(); s = f(W, Pl); tf = time.perf_counter() - t
    x, z = rng.standard_normal(n), rng.standard_normal(2 * n)
    t = time.perf_counter(); s(x, np.zeros(0), z); ts = time.perf_counter() - t
    print("factor %.2f ms solve %.2f ms" % (tf * 1e3, ts * 1e3))
import cvxopt_amd
q, h = -np.ones(n), np.ones(2 * n)
for r in range(2):
    t = time.perf_counter(); sol = cvxopt_amd.coneqp_lp(Pl, q, FakeSp(G), h); t = time.perf_counter() - t
    print("resident coneqp: %.3f s wall (incl. symbolic analysis), %s, %d iterations, pobj %.9e" % (t, sol['status'], sol['iterations'], sol['primal objective']))

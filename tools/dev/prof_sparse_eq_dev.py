"""Developer probe: sparse engine with p equality constraints (46^3 Laplacian box-QP): factor / solve timings."""
import os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import kkt, synth
from test_gpu_sparse import FakeSp, laplace3d, box
k = int(sys.argv[1]) if len(sys.argv) > 1 else 46
P = laplace3d(k); n = P.shape[0]; G = box(n)
dims = {'l': 2 * n, 'q': [], 's': []}
rng = np.random.default_rng(0)
W = synth.random_scaling(dims, seed=0, spread=1.0)
for p in (0, 8, 64, 256):
    A = rng.standard_normal((p, n))
    f = kkt.kkt_chol2(FakeSp(G), dims, A)
    s = f(W, FakeSp(sp.tril(P)))
    tf, ts = [], []
    for r in range(3):
        t = time.perf_counter(); s = f(W, FakeSp(sp.tril(P))); tf.append(time.perf_counter() - t)
        x, y, z = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(2 * n)
        t = time.perf_counter(); s(x, y, z); ts.append(time.perf_counter() - t)
    print("p=%3d: factor %.2f ms, solve %.2f ms" % (p, 1e3 * min(tf), 1e3 * min(ts)))
    f.engine.close()

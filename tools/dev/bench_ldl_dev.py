"""Developer probe: the literal 3x3 'ldl' formulation of the reference on the CPU vs the device engine
behind the same factory name, at a size where the (n+m)^2 KKT matrix fits comfortably (n=512, m=1024)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from cvxopt_amd import kkt, synth
n, m = 512, 1024
pr = synth.dense_qp(n, m, seed=0)
W = synth.random_scaling(pr['dims'], seed=1, spread=1.0)
rng = np.random.default_rng(0)
for name, fac in (("chol2", kkt.kkt_chol2), ("ldl", kkt.kkt_ldl), ("ldl2", kkt.kkt_ldl2)):
    f = fac(pr['G'], pr['dims'], np.zeros((0, n)))
    ts = []
    for r in range(20):
        x, z = rng.standard_normal(n), rng.standard_normal(m)
        t = time.perf_counter(); s = f(W, pr['P']); s(x, np.zeros(0), z); s(x, np.zeros(0), z); ts.append(time.perf_counter() - t)
    print("GPU  %-5s factor + 2 solves: %.3f ms" % (name, 1e3 * np.median(ts)))
    f.engine.close()
try:
    from oracle import refloader
    refloader.load()
    from cvxopt import matrix, spmatrix, misc
    G, P, A = matrix(pr['G']), matrix(pr['P']), spmatrix([], [], [], (0, n))
    Wc = {'d': matrix(W['d']), 'di': matrix(W['di']), 'v': [], 'beta': [], 'r': [], 'rti': []}
    for name, fac in (("chol2", misc.kkt_chol2), ("chol", misc.kkt_chol), ("ldl", misc.kkt_ldl), ("ldl2", misc.kkt_ldl2)):
        f = fac(G, pr['dims'], A)
        ts = []
        for r in range(4):
            t = time.perf_counter()
            s = f(Wc, P)
            for k in range(2):
                s(matrix(rng.standard_normal(n)), matrix(0.0, (0, 1)), matrix(rng.standard_normal(m)))
            ts.append(time.perf_counter() - t)
        print("CPU  %-5s factor + 2 solves: %.1f ms (reference + MKL, %d host threads)" % (name, 1e3 * min(ts[1:]), os.cpu_count()))
except Exception as e:
    print("no reference:", e)

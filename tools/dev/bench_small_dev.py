"""Developer probe: BASELINE configs[0] (n=256, m=512) -- hook-level KKT iteration and whole-solve timings."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import cvxopt_amd
from cvxopt_amd import kkt, synth
n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 512)
pr = synth.dense_qp(n, m, seed=0)
W = synth.random_scaling(pr['dims'], seed=1, spread=1.0)
f = kkt.kkt_chol2(pr['G'], pr['dims'], np.zeros((0, n)))
rng = np.random.default_rng(0)
ts = []
for r in range(30):
    x, z = rng.standard_normal(n), rng.standard_normal(m)
    t = time.perf_counter()
    s = f(W, pr['P'])
    s(x, np.zeros(0), z); s(x, np.zeros(0), z)
    ts.append(time.perf_counter() - t)
print("hook: factor + 2 solves n=%d m=%d: median %.3f ms, min %.3f ms" % (n, m, 1e3 * np.median(ts), 1e3 * min(ts)))
for r in range(3):
    t = time.perf_counter(); sol = cvxopt_amd.coneqp_lp(pr['P'], pr['q'], pr['G'], pr['h']); t = time.perf_counter() - t
print("resident coneqp: %.2f ms for %d iterations (incl. handle creation + upload)" % (1e3 * t, sol['iterations']))
try:
    from oracle import refloader
    refloader.load()
    from cvxopt import matrix, solvers
    solvers.options['show_progress'] = False
    P, q, G, h = matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h'])
    for r in range(3):
        t = time.perf_counter(); ref = solvers.coneqp(P, q, G, h, kktsolver='chol2'); t = time.perf_counter() - t
    print("CPU reference coneqp: %.2f ms for %d iterations" % (1e3 * t, ref['iterations']))
except Exception as e:
    print("no reference:", e)

"""Developer probe: wall time of the device-resident conelp loop on max-cut relaxations min 1'x s.t. w + diag(x) >= 0."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cvxopt_amd


def maxcut(n, seed=0):
    rng = np.random.default_rng(seed)
    w = rng.standard_normal((n, n)); w = 0.5 * (w + w.T)
    G = np.zeros((n * n, n), order='F')
    for j in range(n):
        G[j * (n + 1), j] = -1.0
    return np.ones(n), G, w.ravel(order='F'), {'l': 0, 'q': [], 's': [n]}


for n in [int(a) for a in sys.argv[1:]] or [20, 60, 100]:
    c, G, h, dims = maxcut(n)
    for rep in range(2):
        t = time.perf_counter()
        sol = cvxopt_amd.conelp_device(c, G, h, dims)
        t = time.perf_counter() - t
    print("maxcut n=%d: %s, %d iterations, %.1f ms total, %.2f ms / iteration, pobj %.8f" % (
        n, sol['status'], sol['iterations'], 1e3 * t, 1e3 * t / max(1, sol['iterations']), sol['primal objective']), flush=True)

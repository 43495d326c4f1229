"""Developer probe: wall time of the device-resident conelp loop on max-cut relaxations min 1'x s.t. w + diag(x) >= 0."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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


def many_blocks(nb, mk, n=30, seed=0):
    """random strictly feasible cone LP over nb 's' blocks of order mk"""
    rng = np.random.default_rng(seed)
    dims = {'l': 0, 'q': [], 's': [mk] * nb}
    cols = []
    for _ in range(n):
        a = rng.standard_normal((nb, mk, mk))
        cols.append((0.5 * (a + a.transpose(0, 2, 1))).reshape(-1))
    G = np.asfortranarray(np.array(cols).T)

    def interior():
        a = rng.standard_normal((nb, mk, mk))
        return (a @ a.transpose(0, 2, 1) / mk + 0.5 * np.eye(mk)).reshape(-1)
    x0 = rng.standard_normal(n)
    return -(G.T @ interior()), G, G @ x0 + interior(), dims


if os.environ.get("SDP_MANY"):
    for nb, mk in ((200, 4), (1000, 3), (64, 12)):
        c, G, h, dims = many_blocks(nb, mk)
        for rep in range(2):
            t = time.perf_counter()
            sol = cvxopt_amd.conelp_device(c, G, h, dims)
            t = time.perf_counter() - t
        r = np.linalg.norm(G @ sol['x'] + sol['s'] - h) / np.linalg.norm(h)
        print("%d blocks of order %d: %s, %d iterations, %.1f ms total, %.2f ms / iteration, primal residual %.1e" % (
            nb, mk, sol['status'], sol['iterations'], 1e3 * t, 1e3 * t / max(1, sol['iterations']), r), flush=True)

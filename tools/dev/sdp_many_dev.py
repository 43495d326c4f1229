"""Developer probe: cone LPs over nb 's' blocks of order mk through the device loop; residual of the returned point."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import cvxopt_amd


def many_blocks(nb, mk, n=10, seed=0):
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


for nb, mk in [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]:
    c, G, h, dims = many_blocks(nb, mk)
    sol = cvxopt_amd.conelp_device(c, G, h, dims)
    r = np.linalg.norm(G @ sol['x'] + sol['s'] - h) / np.linalg.norm(h) if sol['x'] is not None else -1
    print("%d x %d: %s, %d iterations, gap %s, primal residual %.1e" % (nb, mk, sol['status'], sol['iterations'], sol['gap'], r), flush=True)

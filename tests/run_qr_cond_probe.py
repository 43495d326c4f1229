"""Probe script (test infrastructure: it imports the reference through oracle/refloader) (round 6): a conelp family whose W^-T G is ill conditioned from the first iteration (two
nearly dependent columns of G: cond(G) = 1 / eps), solved by the reference with kktsolver = 'qr' and 'chol' and by this backend with
'qr' (-> reduced Cholesky engine + conditional refinement, round 6), 'ldl' (reduced form + refinement against the 3 x 3 system), host-driver and device loop."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import refloader

cvx = refloader.load()
cvx.solvers.options['show_progress'] = False
import cvxopt_amd.solvers as gs


def problem(n, m, eps, seed, p=0):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((m, n))
    G[:, n - 1] = G[:, 0] + eps * rng.standard_normal(m)          # sigma_min(G) ~ eps
    x0 = rng.standard_normal(n)
    h = G @ x0 + rng.uniform(0.5, 2.0, m)
    c = -G.T @ rng.uniform(0.5, 2.0, m)
    A = b = None
    if p:
        A = rng.standard_normal((p, n))
        b = A @ x0
    return c, G, h, A, b


def run(f, c, G, h, A, b, **kw):
    M = lambda a: None if a is None else cvx.matrix(np.asfortranarray(np.atleast_2d(a.T).T if a.ndim == 1 else a))
    try:
        sol = f(M(c), M(G), M(h), A=M(A), b=M(b), **kw) if A is not None else f(M(c), M(G), M(h), **kw)
        return "%-9s it %2d pobj %.12e gap %.1e pres %.1e" % (sol['status'], sol['iterations'], sol['primal objective'] or 0.0,
                                                            sol['gap'] or 0.0, sol['primal infeasibility'] or 0.0)
    except Exception as e:
        return "raised %s(%s)" % (type(e).__name__, e)


for (n, m, p) in ((40, 120, 0), (200, 500, 5)):
    for eps in (1e-3, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9):
        c, G, h, A, b = problem(n, m, eps, 1, p)
        print("n=%d m=%d p=%d eps=%g cond(G)=%.1e" % (n, m, p, eps, np.linalg.cond(G)))
        print("   reference qr   :", run(cvx.solvers.conelp, c, G, h, A, b, kktsolver='qr'))
        print("   reference chol :", run(cvx.solvers.conelp, c, G, h, A, b, kktsolver='chol'))
        print("   reference ldl  :", run(cvx.solvers.conelp, c, G, h, A, b, kktsolver='ldl'))
        print("   backend qr host:", run(gs.conelp, c, G, h, A, b, kktsolver='qr', device_loop=False))
        print("   backend qr dev :", run(gs.conelp, c, G, h, A, b, kktsolver='qr'))
        print("   backend ldl hst:", run(gs.conelp, c, G, h, A, b, kktsolver='ldl', device_loop=False))
        print("   backend ldl dev:", run(gs.conelp, c, G, h, A, b, kktsolver='ldl'))

"""Reference solutions of the sparse KKT solve at the sizes SURVEY 8(d) names for config 4 (n = 1e5 .. 1e6): box-QP on the 64^3 and
100^3 7-point Laplacians (n = 262 144 and 1 000 000), one Nesterov-Todd scaling, one right-hand side.

The reference's own sparse kkt_chol2 branch (oracle/_ref with the SuperLU-backed cholmod shim) is too slow at these sizes to be
run inside a test, so the reduced system  S x = bx + G' D^2 bz,  S = P + G' D^2 G  (misc.py:1401-1447, :1489-1565 for the LP cone)
is solved here, once, independently of any code of ours:
    64^3   scipy.sparse.linalg.splu (SuperLU, symmetric mode, MMD ordering; ~200 s)   -> KKT residual 1e-15
    100^3  preconditioned conjugate gradients to a relative residual of 1e-14 (SuperLU does not fit in this container's memory)
The fixture keeps every 64th entry of x, |x|_2 and the achieved residual; tests/test_gpu_sparse_big.py rebuilds the same problem
from the seeds, compares the sampled entries and checks the FULL residual of the device solution.

    python tests/golden/make_golden_sparse_big.py [64] [100]
"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cvxopt_amd import synth            # noqa: E402


def problem(k, seed):
    """the same construction as bench.py --workload sparse / tests: P, G = [I; -I], W, right-hand sides"""
    n = k ** 3
    P = synth.grid_laplacian(k)
    dims = {'l': 2 * n, 'q': [], 's': []}
    W = synth.random_scaling(dims, seed=seed, spread=1.0)
    rng = np.random.default_rng(seed)
    bx, bz = rng.standard_normal(n), rng.standard_normal(2 * n)
    di2 = W['di'] ** 2
    S = (P + sp.diags(di2[:n] + di2[n:])).tocsc()
    rhs = bx + di2[:n] * bz[:n] - di2[n:] * bz[n:]
    return P, W, bx, bz, S, rhs


def main(sizes):
    for k in sizes:
        P, W, bx, bz, S, rhs = problem(k, seed=k)
        t = time.time()
        if k <= 64:
            lu = spl.splu(S, permc_spec='MMD_AT_PLUS_A', diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
            x = lu.solve(rhs)
            how = "scipy.sparse.linalg.splu (SuperLU, symmetric mode)"
        else:
            Minv = sp.diags(1.0 / S.diagonal())
            x, info = spl.cg(S, rhs, rtol=1e-15, atol=0.0, maxiter=20000, M=Minv)
            r = rhs - S @ x
            for _ in range(3):                      # a few steps of refinement on the recurrence's residual drift
                dx, info = spl.cg(S, r, rtol=1e-10, atol=0.0, maxiter=20000, M=Minv)
                x = x + dx
                r = rhs - S @ x
            how = "scipy.sparse.linalg.cg (Jacobi preconditioner) + residual refinement"
        res = np.linalg.norm(S @ x - rhs) / np.linalg.norm(rhs)
        print("k = %d  n = %d  %s: %.1f s, relative residual %.2e" % (k, k ** 3, how, time.time() - t, res))
        np.savez_compressed(os.path.join(HERE, "sparse%d.npz" % k), k=np.array(k), seed=np.array(k), x_sample=x[::64],
                            x_norm=np.array(np.linalg.norm(x)), residual=np.array(res), how=np.array(how))


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [64, 100])

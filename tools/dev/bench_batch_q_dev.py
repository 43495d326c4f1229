"""Developer probe: the batched engine with second-order cones.  B cone QPs of n variables with dims = {'l': nl, 'q': [r] * nc}:
whole device-resident coneqp of the batch (one workgroup per problem around the batched KKT kernels) vs the single-problem
device loop run problem after problem.

    python tools/dev/bench_batch_q_dev.py [B n nl nc r]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cvxopt_amd
from cvxopt_amd.batch import BatchKkt

B, n, nl, nc, r = (int(a) for a in (sys.argv[1:6] if len(sys.argv) >= 6 else (512, 256, 128, 112, 8)))
dims = {'l': nl, 'q': [r] * nc, 's': []}
m = nl + nc * r
rng = np.random.default_rng(0)


def interior(rows):
    u = np.empty((rows, m))
    u[:, :nl] = rng.uniform(0.5, 1.5, (rows, nl))
    w = rng.standard_normal((rows, nc, r))
    w[:, :, 0] = np.linalg.norm(w[:, :, 1:], axis=2) + rng.uniform(0.5, 1.5, (rows, nc))
    u[:, nl:] = w.reshape(rows, nc * r)
    return u


t = time.time()
Bm = rng.standard_normal((B, n, n)) / np.sqrt(n)
P = np.einsum('bij,bkj->bik', Bm, Bm) + 1e-2 * np.eye(n)
Gt = rng.standard_normal((B, n, m))                      # G_b' (n x m, C order) = G_b column-major
x0 = rng.standard_normal((B, n))
z0 = interior(B)
q = -(np.einsum('bij,bj->bi', P, x0) + np.einsum('bnm,bm->bn', Gt, z0))
h = np.einsum('bnm,bn->bm', Gt, x0) + interior(B)
print("generated %d problems (n=%d, cdim=%d: l=%d + %d cones of %d) in %.1f s" % (B, n, m, nl, nc, r, time.time() - t))
bk = BatchKkt(Gt, P, dims=dims)
for rep in range(3):
    t = time.perf_counter()
    res = bk.coneqp(q, h)
    t = time.perf_counter() - t
    its = int(res['iterations'].sum())
    print("batched coneqp: %.3f s, %d lock-step iterations, iterations min/mean/max %d/%.1f/%d, all optimal %s, "
          "%.0f problem-iterations/s" % (t, res['lockstep iterations'], res['iterations'].min(), res['iterations'].mean(),
                                          res['iterations'].max(), bool(np.all(res['status'] == 'optimal')), its / t))
bk.close()
K = min(B, 8)
t = time.perf_counter()
for k in range(K):
    sol = cvxopt_amd.coneqp_device(P[k], q[k], np.asfortranarray(Gt[k].T), h[k], dims)
    assert sol['status'] == 'optimal' and sol['iterations'] == res['iterations'][k], (k, sol['iterations'], res['iterations'][k])
t = time.perf_counter() - t
print("single-problem device loop: %.1f ms per problem (%d problems) -> %.0f problem-iterations/s" % (
    1e3 * t / K, K, res['iterations'][:K].sum() / t))

"""Developer probe: batched config-5 timing on one GPU (B problems of n=512, m=1024).

    python tools/dev/bench_batch_dev.py B [KNOB=VALUE ...]      (test knobs of include/mi355kkt_test.h, e.g. MI355KKT_BATCH_TILES=0)
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import synth
from cvxopt_amd.batch import BatchKkt, coneqp_batch, pack_problems

B, n, m = int(sys.argv[1]), 512, 1024
for kv in sys.argv[2:]:
    from cvxopt_amd import _capi
    _capi.set_knob(*kv.split("=", 1))
    print("knob", kv)
t = time.time()
rng = np.random.default_rng(0)
base = synth.dense_qp(n, m, seed=0)
probs = []
for i in range(B):
    p = dict(base)
    p['q'] = rng.standard_normal(n)
    p['h'] = base['G'] @ rng.standard_normal(n) + rng.uniform(0.1, 1.0, m)
    probs.append(p)
P, q, Gt, h = pack_problems(probs)
print("gen %.1fs" % (time.time() - t))
k = BatchKkt(Gt, P)
di = 10.0 ** rng.uniform(-1, 1, (B, m))
for r in range(3):
    t = time.perf_counter(); info = k.factor(di); t1 = time.perf_counter() - t
    x, z = rng.standard_normal((B, n)), rng.standard_normal((B, m))
    t = time.perf_counter(); k.solve(x, z); t2 = time.perf_counter() - t
    print("B=%d factor %.2f ms (device %.2f ms)  solve %.2f ms  -> %.1f problem-iters/s" % (B, t1 * 1e3, k.factor_ms(), t2 * 1e3, B / (t1 + 2 * t2)))
t = time.perf_counter()
res = coneqp_batch(P, q, Gt, h, kkt=k)
t = time.perf_counter() - t
print("coneqp_batch: %.2f s, iterations min/mean/max %d/%.1f/%d, all optimal %s, %.1f problem-iterations/s" % (
    t, res['iterations'].min(), res['iterations'].mean(), res['iterations'].max(), bool(np.all(res['status'] == 'optimal')),
    res['iterations'].sum() / t))
for r in range(2):
    t = time.perf_counter()
    res2 = coneqp_batch(P, q, Gt, h, kkt=k, resident=True)
    t = time.perf_counter() - t
    print("resident loop: %.3f s, iterations max %d, all optimal %s, %.1f problem-iterations/s, same iters %s" % (
        t, res2['iterations'].max(), bool(np.all(res2['status'] == 'optimal')), res2['iterations'].sum() / t,
        bool(np.array_equal(res2['iterations'], res['iterations']))))

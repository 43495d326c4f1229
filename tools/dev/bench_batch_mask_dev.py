"""Developer probe: the config-5 batch on one GPU with and without the block masks on the diagonal SYRK tiles (g_syrk_skip bit 6)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import synth, _capi
from cvxopt_amd.batch import BatchKkt, coneqp_batch, pack_problems
L = _capi.lib()
B, n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 512, 1024
rng = np.random.default_rng(0)
base = synth.dense_qp(n, m, seed=0)
probs = []
for i in range(B):
    p = dict(base)
    p['q'] = rng.standard_normal(n)
    p['h'] = base['G'] @ rng.standard_normal(n) + rng.uniform(0.1, 1.0, m)
    probs.append(p)
P, q, Gt, h = pack_problems(probs)
k = BatchKkt(Gt, P)
di = 10.0 ** rng.uniform(-1, 1, (B, m))
ref = None
for mask in (0, 64, 0, 64):
    L.mi355kkt_debug_syrk_skip(mask)
    for r in range(3):
        k.factor(di)
    fm = k.factor_ms()
    ts = []
    for r in range(2):
        t = time.perf_counter()
        res = coneqp_batch(P, q, Gt, h, kkt=k, resident=True)
        ts.append(time.perf_counter() - t)
    if ref is None:
        ref = res
    print("skip mask %3d: factor (device) %.3f ms; resident loop %.3f s, %.1f problem-iterations/s, all optimal %s, x == first run: %s" % (
        mask, fm, min(ts), res['iterations'].sum() / min(ts), bool(np.all(res['status'] == 'optimal')),
        float(np.max(np.abs(res['x'] - ref['x'])))), flush=True)
L.mi355kkt_debug_syrk_skip(0)

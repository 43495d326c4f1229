"""Developer probe: handle churn.  Thousands of create / solve / destroy cycles over small problems of every cone type,
with "foreign" contents written into device memory between them (a large buffer filled with NaN bytes, allocated and freed: the next
handles get recycled blocks of it), the solutions compared with the first pass of the same problems (bit for bit).
    python tools/dev/handle_churn_dev.py [cycles] [seed]
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi, kkt, synth

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
L = _capi.lib()


def poison(mbytes):
    import ctypes as C
    b = _capi.DeviceBuffer(mbytes << 20)
    a = np.full((mbytes << 20) // 8, np.nan)
    _capi.check(L.mi355kkt_memcpy_h2d(b.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes), "h2d")
    b.free()


def problem(kind, seed):
    r = np.random.default_rng(seed)
    if kind == 0:      # LP cone QP
        pr = synth.dense_qp(int(r.integers(4, 40)), int(r.integers(4, 80)), seed=seed)
        return ("qp", pr)
    n = int(r.integers(3, 12))
    if kind == 1:      # second-order cones
        dims = {'l': int(r.integers(0, 5)), 'q': [int(k) for k in r.integers(2, 6, int(r.integers(1, 4)))], 's': []}
    else:              # semidefinite blocks
        dims = {'l': int(r.integers(0, 4)), 'q': [], 's': [int(k) for k in r.integers(2, 5, int(r.integers(1, 5)))]}
    m = dims['l'] + sum(dims['q']) + sum(k * k for k in dims['s'])
    G = r.standard_normal((m, n))
    return ("cone", dict(G=G, dims=dims, n=n, m=m, seed=seed))


def run(tag, pr):
    if tag == "qp":
        sol = kkt.coneqp_device(pr['P'], pr['q'], pr['G'], pr['h'], pr['dims'])
        return np.concatenate([np.asarray(sol['x']).ravel(), [sol['iterations']]])
    W = synth.random_scaling(pr['dims'], seed=pr['seed'], spread=1.0)
    f = kkt.kkt_chol(pr['G'], pr['dims'], np.zeros((0, pr['n'])))
    r = np.random.default_rng(pr['seed'] + 1)
    x, z = r.standard_normal(pr['n']), r.standard_normal(pr['m'])
    f(W)(x, np.zeros(0), z)
    f.engine.close()
    return np.concatenate([x, z])


probs = [problem(int(rng.integers(0, 3)), 1000 + i) for i in range(60)]
first = [run(*p) for p in probs]
t0 = time.time()
bad = 0
for c in range(cycles):
    if c % 50 == 0:
        poison(int(rng.integers(1, 64)))
    i = int(rng.integers(0, len(probs)))
    got = run(*probs[i])
    if not (got.shape == first[i].shape and np.array_equal(got, first[i])):
        bad += 1
        print("cycle %d problem %d (%s): differs from its first run, max abs diff %s" % (
            c, i, probs[i][0], np.max(np.abs(got - first[i])) if got.shape == first[i].shape else "shape"), flush=True)
    if c % 200 == 199:
        print("%d cycles, %d mismatches, %.1f s" % (c + 1, bad, time.time() - t0), flush=True)
print("done: %d cycles, %d mismatches" % (cycles, bad))
sys.exit(1 if bad else 0)

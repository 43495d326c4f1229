"""Developer probe: BASELINE configs[2] (SOCP, n=2048, cdim=8192) for cone dimensions r = 4, 8, 64:
hook-level KKT factor / solve times, the device-resident conelp solve, and the CPU reference conelp once."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import cvxopt_amd
from cvxopt_amd import kkt, synth

n, cdim = 2048, 8192
for r in (4, 8, 64):
    pr = synth.socp(n=n, ncones=cdim // r, r=r, seed=0)
    W = synth.random_scaling(pr['dims'], seed=1, spread=1.0)
    f = kkt.kkt_chol(pr['G'], pr['dims'], np.zeros((0, n)))
    rng = np.random.default_rng(0)
    tf, ts = [], []
    for k in range(5):
        t = time.perf_counter(); s = f(W); tf.append(time.perf_counter() - t)
        x, z = rng.standard_normal(n), rng.standard_normal(cdim)
        t = time.perf_counter(); s(x, np.zeros(0), z); ts.append(time.perf_counter() - t)
    tm = f.engine.timings()
    f.engine.close()
    for k in range(2):
        t = time.perf_counter(); sol = cvxopt_amd.conelp_device(pr['c'], pr['G'], pr['h'], pr['dims']); t = time.perf_counter() - t
    print("r=%2d (%4d cones): hook factor %.2f ms (device %.2f: assemble %.2f potrf %.2f), solve %.2f ms (device %.2f); "
          "resident conelp %.3f s, %d iterations, %s" % (r, cdim // r, 1e3 * min(tf), tm['factor_ms'], tm['assemble_ms'], tm['potrf_ms'],
                                                        1e3 * min(ts), tm['solve_ms'], t, sol['iterations'], sol['status']))
try:
    from oracle import refloader
    refloader.load()
    from cvxopt import matrix, solvers
    solvers.options['show_progress'] = False
    pr = synth.socp(n=n, ncones=cdim // 8, r=8, seed=0)
    t = time.perf_counter(); ref = solvers.conelp(matrix(pr['c']), matrix(pr['G']), matrix(pr['h']), pr['dims']); t = time.perf_counter() - t
    print("CPU reference conelp r=8: %.2f s, %d iterations (%s)" % (t, ref['iterations'], ref['status']))
except Exception as e:
    print("no reference:", e)

"""Developer probe (round 5): the two-sweep triangular solves with and without the dedicated poller wave (test knob
MI355KKT_TRSV_AHEAD=0: every thread polls its own granule behind its strip loads, the round-4 kernel), same process, same factor:
solve() time (gemv + 2 trsv + gemv), bitwise agreement of the two results, residual of the reduced system."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi, kkt, synth

for n, m, spread in ((8192, 1024, 1.0), (2048, 1024, 1.0), (4096, 1024, 3.0), (1024, 2048, 4.0), (512, 600, 2.0)):
    pr = synth.dense_qp(n, m, seed=1)
    eng = kkt._Engine(_capi.CHOL2, pr['G'], pr['dims'], kkt._EmptyA(n))
    Hd = _capi.DeviceBuffer.from_array(np.asfortranarray(pr['P']))
    eng._mode = "dense"
    eng.set_H_device(Hd.ptr, n)
    rng = np.random.default_rng(0)
    di = 10.0 ** rng.uniform(-spread, spread, m)
    did = _capi.DeviceBuffer.from_array(di)
    bx, bz = rng.standard_normal(n), rng.standard_normal(m)
    xd, zd, yd = _capi.DeviceBuffer(8 * n), _capi.DeviceBuffer(8 * m), _capi.DeviceBuffer(8)
    S = pr['P'] + (pr['G'] * (di * di)[:, None]).T @ pr['G']
    rhs = bx + pr['G'].T @ (di * di * bz)
    eng.factor_device(di_ptr=did.ptr)
    eng.sync()
    res = {}
    for poller in (0, 1, 0, 1):
        _capi.set_knob("MI355KKT_TRSV_AHEAD", poller)
        best = 1e9
        for rep in range(8):
            _capi.check(_capi.lib().mi355kkt_memcpy_h2d(xd.ptr, bx.ctypes.data, 8 * n), "h2d")
            _capi.check(_capi.lib().mi355kkt_memcpy_h2d(zd.ptr, bz.ctypes.data, 8 * m), "h2d")
            eng.solve_device(xd.ptr, yd.ptr, zd.ptr)
            eng.sync()
            best = min(best, eng.timings()["solve_ms"])
        x = xd.to_array((n,))
        r = float(np.linalg.norm(S @ x - rhs) / np.linalg.norm(rhs))
        res[poller] = x
        print("n=%d spread=%g poller=%d: solve %.4f ms, relative residual of S x = rhs %.2e" % (n, spread, poller, best, r))
    print("    bitwise equal: %s" % np.array_equal(res[0], res[1]))
    _capi.set_knob(None, None)
    eng.close()

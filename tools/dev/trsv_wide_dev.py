"""Developer probe (round 6): 512-row all-CU triangular solves (trsv_wide_kernel, trsv512.hip) against the round-4 pair kernel with global
refinement, same process, same factor: solve() time, agreement of the two results, KKT residual of both."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi, kkt, synth

for n, m, spread in ((8192, 1024, 1.0), (2048, 1024, 1.0), (4096, 1024, 3.0), (1024, 2048, 4.0), (256, 300, 2.0)):
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
    eng.factor_device(di_ptr=did.ptr)
    eng.sync()
    S = pr['P'] + (pr['G'] * (di * di)[:, None]).T @ pr['G']
    rhs = bx + pr['G'].T @ (di * di * bz)
    res = {}
    for pair in (0, 1, 0, 1):
        _capi.set_knob("MI355KKT_TRSV_WIDE", pair)
        best = 1e9
        fbest = 1e9
        for rep in range(4):
            eng.factor_device(di_ptr=did.ptr)
            eng.sync()
            fbest = min(fbest, eng.timings()["factor_ms"])
        for rep in range(6):
            _capi.check(_capi.lib().mi355kkt_memcpy_h2d(xd.ptr, bx.ctypes.data, 8 * n), "h2d")
            _capi.check(_capi.lib().mi355kkt_memcpy_h2d(zd.ptr, bz.ctypes.data, 8 * m), "h2d")
            eng.solve_device(xd.ptr, yd.ptr, zd.ptr)
            eng.sync()
            best = min(best, eng.timings()["solve_ms"])
        x = xd.to_array((n,))
        r = float(np.linalg.norm(S @ x - rhs) / np.linalg.norm(rhs))
        res[pair] = x
        print("n=%d spread=%g wide=%d: factor %.4f ms, solve %.4f ms, relative residual of S x = rhs %.2e" % (n, spread, pair, fbest, best, r))
    print("    max relative difference wide vs pair: %.2e" % (np.max(np.abs(res[1] - res[0])) / np.max(np.abs(res[0]))))
    _capi.set_knob(None, None)
    eng.close()

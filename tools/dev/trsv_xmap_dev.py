"""Developer probe (round 4): the persistent triangular solves with the chain's workgroups grouped per XCD (knob MI355KKT_TRSV_XMAP=1:
eight consecutive block rows on one XCD) against the default mapping, same process, same factor; solve() = gemv + 2 trsv + gemv."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi, kkt, synth

for n, m in ((8192, 1024), (2048, 1024), (4096, 1024)):
    pr = synth.dense_qp(n, m, seed=1)
    eng = kkt._Engine(_capi.CHOL2, pr['G'], pr['dims'], kkt._EmptyA(n))
    Hd = _capi.DeviceBuffer.from_array(np.asfortranarray(pr['P']))
    eng._mode = "dense"
    eng.set_H_device(Hd.ptr, n)
    rng = np.random.default_rng(0)
    did = _capi.DeviceBuffer.from_array(rng.uniform(0.5, 2.0, m))
    bx, bz = rng.standard_normal(n), rng.standard_normal(m)
    xd, zd, yd = _capi.DeviceBuffer(8 * n), _capi.DeviceBuffer(8 * m), _capi.DeviceBuffer(8)
    eng.factor_device(di_ptr=did.ptr)
    eng.sync()
    ref = None
    for xmap in (0, 1, 0, 1):
        _capi.set_knob("MI355KKT_TRSV_XMAP", xmap)
        best = 1e9
        for rep in range(6):
            _capi.check(_capi.lib().mi355kkt_memcpy_h2d(xd.ptr, bx.ctypes.data, 8 * n), "h2d")
            _capi.check(_capi.lib().mi355kkt_memcpy_h2d(zd.ptr, bz.ctypes.data, 8 * m), "h2d")
            eng.solve_device(xd.ptr, yd.ptr, zd.ptr)
            eng.sync()
            best = min(best, eng.timings()["solve_ms"])
        x = xd.to_array((n,))
        if ref is None:
            ref = x
        print("n=%d xmap=%d: solve %.4f ms (event, best of 6); bitwise equal to the default mapping: %s" % (n, xmap, best, np.array_equal(x, ref)))
    _capi.set_knob(None, None)
    eng.close()

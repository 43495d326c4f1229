"""Developer probe (round 6): a few factor + solve calls of one dense engine for a kernel trace (rocprofv3 --kernel-trace --stats):
durations of block_inverse512_kernel, trsv_wide_kernel and the kernels around them.  usage: wide_prof_dev.py n [wide]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi, kkt, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
wide = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = 1024
pr = synth.dense_qp(n, m, seed=1)
eng = kkt._Engine(_capi.CHOL2, pr['G'], pr['dims'], kkt._EmptyA(n))
Hd = _capi.DeviceBuffer.from_array(np.asfortranarray(pr['P']))
eng._mode = "dense"
eng.set_H_device(Hd.ptr, n)
rng = np.random.default_rng(0)
di = 10.0 ** rng.uniform(-1, 1, m)
did = _capi.DeviceBuffer.from_array(di)
bx, bz = rng.standard_normal(n), rng.standard_normal(m)
xd, zd, yd = _capi.DeviceBuffer(8 * n), _capi.DeviceBuffer(8 * m), _capi.DeviceBuffer(8)
_capi.set_knob("MI355KKT_TRSV_WIDE", wide)
for rep in range(6):
    eng.factor_device(di_ptr=did.ptr)
    for s in range(3):
        _capi.check(_capi.lib().mi355kkt_memcpy_h2d(xd.ptr, bx.ctypes.data, 8 * n), "h2d")
        _capi.check(_capi.lib().mi355kkt_memcpy_h2d(zd.ptr, bz.ctypes.data, 8 * m), "h2d")
        eng.solve_device(xd.ptr, yd.ptr, zd.ptr)
    eng.sync()
print("n=%d wide=%d" % (n, wide), eng.timings())
_capi.set_knob(None, None)
eng.close()

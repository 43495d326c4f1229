"""Developer probe (round 6): timeline of one 512-row all-CU triangular solve (trsv_wide_kernel, debug library: CVXOPT_AMD_LIB =
cvxopt_amd/libmi355kkt_debug.so).  16 stamps of the 100 MHz s_memrealtime clock per workgroup, printed per 512-block in us relative
to the first workgroup's entry: min / max over the block's workgroups."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi, kkt, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
m = 1024
L = _capi.lib()
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
eng.factor_device(di_ptr=did.ptr)
eng.sync()
R = 8 if n <= 2048 else 16
nwg = n // R
ts = _capi.DeviceBuffer.from_array(np.zeros(nwg * 16))
for rep in range(3):
    _capi.check(L.mi355kkt_memcpy_h2d(xd.ptr, bx.ctypes.data, 8 * n), "h2d")
    _capi.check(L.mi355kkt_memcpy_h2d(zd.ptr, bz.ctypes.data, 8 * m), "h2d")
    if rep == 2:
        _capi.check(L.mi355kkt_debug_wide_ts(C.c_void_p(ts.ptr)), "ts")
    eng.solve_device(xd.ptr, yd.ptr, zd.ptr)
    eng.sync()
_capi.check(L.mi355kkt_debug_wide_ts(None), "off")
# the stamps of the LAST launch (the backward solve): blocks in dispatch order are the reversed block rows
t = ts.to_array((nwg, 16), dtype="int64", order="C").astype(float) / 100.0
t0 = t[:, 0].min()
t = np.where(t > 0, t - t0, np.nan)
names = ["entry", "Ms in LDS", "far field done", "x0[j-1] seen", "t published", "t_j seen", "x0 published", "d[j-1] used",
         "x0_j seen", "b published", "b_j seen", "end", "far0 waited", "far0 done", "far1 waited", "far1 done"]
per = 512 // R
print("n = %d, R = %d, %d workgroups, solve() %.4f ms; us since the first entry, min..max over the %d workgroups of a block"
      % (n, R, nwg, eng.timings()["solve_ms"], per))
for b in range(nwg // per):
    blk = t[b * per:(b + 1) * per]
    print("block %2d (dispatch order)" % b)
    for k, nm in enumerate(names):
        col = blk[:, k]
        if np.all(np.isnan(col)):
            continue
        print("    %-16s %8.2f .. %8.2f" % (nm, np.nanmin(col), np.nanmax(col)))
eng.close()

"""Developer probe: phase timestamps (shader clocks) of potf2_la_kernel on one 128 x 128 block."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi
L = _capi.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = np.random.default_rng(0)
B = rng.standard_normal((n, n)) / np.sqrt(n)
S = np.asfortranarray(B.T @ B + np.eye(n))
ts = _capi.DeviceBuffer.from_array(np.zeros(48))
_capi.check(L.mi355kkt_debug_potf2_ts(C.c_void_p(ts.ptr)), "ts")
for rep in range(3):
    dS = _capi.DeviceBuffer.from_array(S)
    ms, info = C.c_float(), C.c_int()
    _capi.check(L.mi355kkt_op_potrf(dS.ptr, n, n, C.byref(info), C.byref(ms)), "potrf")
    t = ts.to_array((48,), dtype="int64")
    t0 = t[0]
    print("rep %d: kernel-level ms %.4f info %d total cycles %d" % (rep, ms.value, info.value, t[42] - t0))
    for j in range(8):
        b = 1 + 5 * j
        print("  jb %3d: start %7d | load->D %5d | D %6d | store+misc %5d | wait B1 %5d | U_first+B2 %6d" % (
            16 * j, t[b] - t0, t[b + 1] - t[b], t[b + 2] - t[b + 1], t[b + 3] - t[b + 2], t[b + 4] - t[b + 3],
            (t[b + 5] if j < 7 else t[41]) - t[b + 4]))
    print("  inverses %d" % (t[42] - t[41]))
_capi.check(L.mi355kkt_debug_potf2_ts(None), "ts off")

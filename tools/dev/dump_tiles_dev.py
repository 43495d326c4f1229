"""Developer probe: all per-tile stamps of one persistent tile Cholesky (shader clocks) -> .npy, for offline analysis."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi
L = _capi.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
out = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/tiles_%d.npy" % n
NT = (n + 127) // 128
ntiles = NT * (NT + 1) // 2
rng = np.random.default_rng(0)
B = rng.standard_normal((n, n)) / np.sqrt(n)
S = np.asfortranarray(B.T @ B + np.eye(n))
ts = _capi.DeviceBuffer.from_array(np.zeros(ntiles * 8))
_capi.check(L.mi355kkt_debug_tile_ts(C.c_void_p(ts.ptr)), "ts")
for rep in range(2):
    dS = _capi.DeviceBuffer.from_array(S)
    ms, info = C.c_float(), C.c_int()
    _capi.check(L.mi355kkt_op_potrf(dS.ptr, n, n, C.byref(info), C.byref(ms)), "potrf")
t = ts.to_array((ntiles, 8), dtype="int64", order="C")
_capi.check(L.mi355kkt_debug_tile_ts(None), "off")
np.save(out, t)
print("n %d: %.3f ms info %d -> %s" % (n, ms.value, info.value, out))

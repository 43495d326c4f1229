"""Developer probe: dense potrf timing only (n = 8192 by default), for A/B runs of env-selected variants."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi
L = _capi.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(0)
B = rng.standard_normal((n, n)) / np.sqrt(n)
S = np.asfortranarray(B.T @ B + np.eye(n))
ts = []
for r in range(6):
    dS = _capi.DeviceBuffer.from_array(S)
    ms, info = C.c_float(), C.c_int()
    _capi.check(L.mi355kkt_op_potrf(dS.ptr, n, n, C.byref(info), C.byref(ms)), "potrf")
    ts.append(ms.value)
print("potrf %d: min %.3f ms  median %.3f ms  (%.2f TF/s at min) info=%d  env=%s" % (
    n, min(ts), sorted(ts)[len(ts) // 2], float(n) ** 3 / 3 / min(ts) / 1e9, info.value,
    {k: v for k, v in os.environ.items() if k.startswith("MI355KKT")}))

"""Developer probe: a handful of launches of the headline SYRK (n = 8192, m = 16384) and nothing else, for rocprofv3 --pmc passes."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi
L = _capi.lib()
n, m = 8192, 16384
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(0)
G = np.asfortranarray(rng.standard_normal((m, n)))
di = rng.uniform(0.5, 2, m)
dG, ddi, dS = _capi.DeviceBuffer.from_array(G), _capi.DeviceBuffer.from_array(di), _capi.DeviceBuffer(n * n * 8)
ms = C.c_float()
ts = []
for r in range(reps):
    _capi.check(L.mi355kkt_op_syrk_scaled(dG.ptr, m, m, n, ddi.ptr, None, n, dS.ptr, n, C.byref(ms)), "syrk")
    ts.append(ms.value)
print("syrk n %d m %d: min %.3f ms  median %.3f ms  %.2f TF/s" % (n, m, min(ts), sorted(ts)[len(ts) // 2], m * float(n) * n / min(ts) / 1e9))

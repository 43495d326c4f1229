import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi
L = _capi.lib()
n, m = 8192, 16384
rng = np.random.default_rng(0)
G = np.asfortranarray(rng.standard_normal((m, n)))
di = rng.uniform(0.5, 2, m)
dG, ddi, dS = _capi.DeviceBuffer.from_array(G), _capi.DeviceBuffer.from_array(di), _capi.DeviceBuffer(n * n * 8)
ms = C.c_float()
for mask in (0, 1, 2, 3, 7, 0):
    L.mi355kkt_debug_syrk_skip(mask)
    best = 1e9
    for r in range(3):
        _capi.check(L.mi355kkt_op_syrk_scaled(dG.ptr, m, m, n, ddi.ptr, None, n, dS.ptr, n, C.byref(ms)), "syrk")
        best = min(best, ms.value)
    print("skip mask %d: syrk %.3f ms  %.2f TF/s" % (mask, best, m * float(n) * n / best / 1e9))
L.mi355kkt_debug_syrk_skip(0)

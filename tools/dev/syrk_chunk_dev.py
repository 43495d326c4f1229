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
for r in range(4):
    _capi.check(L.mi355kkt_op_syrk_scaled(dG.ptr, m, m, n, ddi.ptr, None, n, dS.ptr, n, C.byref(ms)), "syrk")
    print("KCHUNK=%s syrk %.3f ms  %.2f TF/s" % (os.environ.get("MI355KKT_SYRK_KCHUNK", "-"), ms.value, m * float(n) * n / ms.value / 1e9))

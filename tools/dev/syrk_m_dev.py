"""Developer probe: scaled SYRK rate vs the contraction length m (operand footprint vs the 256 MB Infinity Cache)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi
L = _capi.lib()
n = 8192
rng = np.random.default_rng(0)
ms = C.c_float()
for m in (1024, 2048, 4096, 8192, 16384, 32768):
    G = np.asfortranarray(rng.standard_normal((m, n)))
    di = rng.uniform(0.5, 2, m)
    dG, ddi, dS = _capi.DeviceBuffer.from_array(G), _capi.DeviceBuffer.from_array(di), _capi.DeviceBuffer(n * n * 8)
    ts = []
    for r in range(5):
        _capi.check(L.mi355kkt_op_syrk_scaled(dG.ptr, m, m, n, ddi.ptr, None, n, dS.ptr, n, C.byref(ms)), "syrk")
        ts.append(ms.value)
    print("m %6d (G %5.0f MB): syrk min %.3f ms  %.2f TF/s" % (m, m * n * 8 / 1e6, min(ts), m * float(n) * n / min(ts) / 1e9))
    del dG, ddi, dS

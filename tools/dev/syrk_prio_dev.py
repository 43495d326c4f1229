"""Developer probe: static wave-priority modes of the scaled SYRK main loop (g_syrk_skip bits 4..5)."""
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
for rnd in range(2):
    for mode in (0, 1, 2, 3):
        L.mi355kkt_debug_syrk_skip(mode << 4)
        ts = []
        for r in range(4):
            _capi.check(L.mi355kkt_op_syrk_scaled(dG.ptr, m, m, n, ddi.ptr, None, n, dS.ptr, n, C.byref(ms)), "syrk")
            ts.append(ms.value)
        print("prio mode %d: syrk min %.3f ms median %.3f ms  %.2f TF/s" % (mode, min(ts), sorted(ts)[2], m * float(n) * n / min(ts) / 1e9))
L.mi355kkt_debug_syrk_skip(0)

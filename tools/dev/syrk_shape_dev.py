import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi
L = _capi.lib()
for n, m in [(2048, 65536), (1024, 131072), (4096, 32768), (8192, 16384), (8192, 4096)]:
    rng = np.random.default_rng(0)
    G = np.asfortranarray(rng.standard_normal((m, n)))
    di = rng.uniform(0.5, 2, m)
    dG, ddi, dS = _capi.DeviceBuffer.from_array(G), _capi.DeviceBuffer.from_array(di), _capi.DeviceBuffer(n * n * 8)
    ms = C.c_float()
    best = 1e9
    for r in range(3):
        _capi.check(L.mi355kkt_op_syrk_scaled(dG.ptr, m, m, n, ddi.ptr, None, n, dS.ptr, n, C.byref(ms)), "syrk")
        best = min(best, ms.value)
    nt = (n + 127) // 128
    print("n=%d m=%d tiles=%d: syrk %.3f ms  %.2f TF/s (lower-tri flops) ; incl. diag-tile waste %.2f TF/s" % (
        n, m, nt * (nt + 1) // 2, best, m * float(n) * n / best / 1e9, m * 128.0 * 128 * 2 * (nt * (nt + 1) // 2) / best / 1e9))
    del dG, ddi, dS

"""Developer probe: the SYRK with / without the balanced diagonal-tile path (ablation bit 5) at the headline and batch shapes."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi
L = _capi.lib()
rng = np.random.default_rng(0)
for n, m in ((8192, 16384), (2048, 8192), (512, 1024)):
    G = np.asfortranarray(rng.standard_normal((m, n)))
    di = rng.uniform(0.5, 2, m)
    dG, ddi, dS = _capi.DeviceBuffer.from_array(G), _capi.DeviceBuffer.from_array(di), _capi.DeviceBuffer(n * n * 8)
    ms = C.c_float()
    out = {}
    for mask in (0, 32, 0, 32):
        L.mi355kkt_debug_syrk_skip(mask)
        ts = []
        for r in range(5):
            _capi.check(L.mi355kkt_op_syrk_scaled(dG.ptr, m, m, n, ddi.ptr, None, n, dS.ptr, n, C.byref(ms)), "syrk")
            ts.append(ms.value)
        out.setdefault(mask, []).append(min(ts))
    L.mi355kkt_debug_syrk_skip(0)
    print("n %5d m %5d: balanced diagonal tiles %.4f ms, general path %.4f ms" % (n, m, min(out[0]), min(out[32])))

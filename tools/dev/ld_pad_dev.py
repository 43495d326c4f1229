"""Developer probe: does a leading dimension that is a power of two (column stride 128 KB for G, 64 KB for S) cost cache-set conflicts?
Times the headline SYRK for several ldG and the dense potrf / the triangular solves for several lda (same matrices, padded copies)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi
L = _capi.lib()
n, m = 8192, 16384
what = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(0)
ms = C.c_float()
if what in ("all", "syrk"):
    G = rng.standard_normal((m, n))
    di = rng.uniform(0.5, 2, m)
    ddi, dS = _capi.DeviceBuffer.from_array(di), _capi.DeviceBuffer(n * n * 8)
    for ld in (m, m + 16, m + 32, m + 144, m, m + 16):
        Gp = np.zeros((ld, n), order='F')
        Gp[:m, :] = G
        dG = _capi.DeviceBuffer.from_array(Gp)
        del Gp
        ts = []
        for r in range(reps):
            _capi.check(L.mi355kkt_op_syrk_scaled(dG.ptr, ld, m, n, ddi.ptr, None, n, dS.ptr, n, C.byref(ms)), "syrk")
            ts.append(ms.value)
        print("syrk ldG %6d: min %.3f ms  median %.3f ms  %.2f TF/s" % (ld, min(ts), sorted(ts)[len(ts) // 2], m * float(n) * n / min(ts) / 1e9), flush=True)
        del dG
    del G
if what in ("all", "potrf"):
    B = rng.standard_normal((n, n)) / np.sqrt(n)
    S = B.T @ B + np.eye(n)
    del B
    for ld in (n, n + 16, n + 32, n + 144, n, n + 16):
        Sp = np.zeros((ld, n), order='F')
        Sp[:n, :] = S
        ts = []
        for r in range(reps):
            dS = _capi.DeviceBuffer.from_array(Sp)
            info = C.c_int()
            _capi.check(L.mi355kkt_op_potrf(dS.ptr, ld, n, C.byref(info), C.byref(ms)), "potrf")
            ts.append(ms.value)
        print("potrf lda %6d: min %.3f ms  median %.3f ms  info %d" % (ld, min(ts), sorted(ts)[len(ts) // 2], info.value), flush=True)

"""Developer probe (not a test): per-stage timings at a given size through the C ABI."""
import ctypes as C, sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi, kkt, synth

def main(n, m, reps=3):
    L = _capi.lib()
    rng = np.random.default_rng(0)
    G = np.asfortranarray(rng.standard_normal((m, n)))
    B = rng.standard_normal((n, n)) / np.sqrt(n)
    P = np.asfortranarray(B.T @ B + 1e-2 * np.eye(n))
    di = 10.0 ** rng.uniform(-1, 1, m)
    dG, ddi, dP = _capi.DeviceBuffer.from_array(G), _capi.DeviceBuffer.from_array(di), _capi.DeviceBuffer.from_array(P)
    dS = _capi.DeviceBuffer(n * n * 8)
    ms = C.c_float()
    for r in range(reps):
        _capi.check(L.mi355kkt_op_syrk_scaled(dG.ptr, m, m, n, ddi.ptr, dP.ptr, n, dS.ptr, n, C.byref(ms)), "syrk")
        print("syrk %d x %d: %.3f ms  %.2f TF/s" % (m, n, ms.value, m * float(n) * n / ms.value / 1e9))
    info = C.c_int()
    for r in range(reps):
        _capi.check(L.mi355kkt_op_syrk_scaled(dG.ptr, m, m, n, ddi.ptr, dP.ptr, n, dS.ptr, n, None), "syrk")
        _capi.check(L.mi355kkt_op_potrf(dS.ptr, n, n, C.byref(info), C.byref(ms)), "potrf")
        print("potrf %d: %.3f ms  %.2f TF/s info=%d" % (n, ms.value, float(n) ** 3 / 3 / ms.value / 1e9, info.value))
    x = rng.standard_normal(n)
    dx = _capi.DeviceBuffer.from_array(x)
    for tr in (0, 1):
        for r in range(reps):
            _capi.check(L.mi355kkt_op_trsm_lower(dS.ptr, n, n, dx.ptr, n, 1, tr, C.byref(ms)), "trsm")
            print("trsv trans=%d: %.3f ms" % (tr, ms.value))
    z = rng.standard_normal(m)
    dz, dzs, dy = _capi.DeviceBuffer.from_array(z), _capi.DeviceBuffer(m * 8), _capi.DeviceBuffer.from_array(x)
    for r in range(reps):
        _capi.check(L.mi355kkt_op_gemv_t_scaled(dG.ptr, m, m, n, ddi.ptr, dz.ptr, dzs.ptr, dy.ptr, C.byref(ms)), "gemv_t")
        print("gemv_t: %.3f ms  %.2f TB/s" % (ms.value, m * n * 8 / ms.value / 1e9))
    for r in range(reps):
        _capi.check(L.mi355kkt_op_gemv_n_scaled(dG.ptr, m, m, n, ddi.ptr, dx.ptr, dzs.ptr, dz.ptr, C.byref(ms)), "gemv_n")
        print("gemv_n: %.3f ms  %.2f TB/s" % (ms.value, m * n * 8 / ms.value / 1e9))
    # whole hook path
    A = np.zeros((0, n))
    f = kkt.kkt_chol2(G, {'l': m, 'q': [], 's': []}, A)
    W = {'d': 1.0 / di, 'di': di, 'v': [], 'beta': [], 'r': [], 'rti': []}
    for r in range(reps):
        t = time.perf_counter(); s = f(W, P); t1 = time.perf_counter() - t
        bx, by, bz = rng.standard_normal(n), np.zeros(0), rng.standard_normal(m)
        t = time.perf_counter(); s(bx, by, bz); t2 = time.perf_counter() - t
        print("hook factor %.2f ms solve %.2f ms" % (t1 * 1e3, t2 * 1e3), f.engine.timings())

if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]))

"""Developer probe (round 4): the scaled SYRK with (a) the stream-K remainder round and (b) the XCD-local k synchronisation.

Checks the result against NumPy (small shapes) / against the free-running kernel (headline shape), times every variant
(best and median of `reps` launches) and prints the number of waits that ran into their bound.
    python tools/dev/syrk_sync_dev.py [reps]
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi
L = _capi.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7


def run(dG, ddi, dS, m, n, knob, reps):
    _capi.set_knob("MI355KKT_SYRK_SYNC", knob)
    ms = C.c_float()
    ts = []
    for r in range(reps + 1):
        _capi.check(L.mi355kkt_op_syrk_scaled(dG.ptr, m, m, n, ddi.ptr, None, n, dS.ptr, n, C.byref(ms)), "syrk")
        if r:
            ts.append(ms.value)
    to = C.c_int()
    ng = L.mi355kkt_test_syrk_sync_state(C.byref(to))
    return min(ts), sorted(ts)[len(ts) // 2], ng, to.value


def lower(S):
    return np.tril(S)


# ---- small shapes: against NumPy (stream-K remainder; groups only where K >= 2048 and full rounds exist)
for n, m in [(2048, 8192), (1024, 8192), (4096, 8192), (2176, 4096), (640, 2048), (300, 77)]:
    rng = np.random.default_rng(n + m)
    G = np.asfortranarray(rng.standard_normal((m, n)))
    di = rng.uniform(0.5, 2, m)
    ref = (G * (di * di)[:, None]).T @ G
    dG, ddi, dS = _capi.DeviceBuffer.from_array(G), _capi.DeviceBuffer.from_array(di), _capi.DeviceBuffer(n * n * 8)
    for knob in ("off", None):
        best, med, ng, to = run(dG, ddi, dS, m, n, knob, reps)
        S = dS.to_array((n, n), order="F") if hasattr(dS, "to_array") else None
        if S is None:
            S = np.empty((n, n), order="F")
            _capi.check(L.mi355kkt_memcpy_d2h(S.ctypes.data_as(C.c_void_p), dS.ptr, n * n * 8), "d2h")
        err = np.max(np.abs(lower(S) - lower(ref))) / np.max(np.abs(ref))
        print("n=%d K=%d sync=%s: best %.3f median %.3f ms  %.1f TF/s  groups %d timeouts %d  max rel err vs NumPy %.2e" % (
            n, m, knob or "default", best, med, m * float(n) * n / best / 1e9, ng, to, err), flush=True)
    del dG, ddi, dS

# ---- headline shape: every variant against the free-running kernel
n, m = 8192, 16384
rng = np.random.default_rng(0)
G = np.asfortranarray(rng.standard_normal((m, n)))
di = rng.uniform(0.5, 2, m)
dG, ddi, dS = _capi.DeviceBuffer.from_array(G), _capi.DeviceBuffer.from_array(di), _capi.DeviceBuffer(n * n * 8)
base = None
for knob in ("off", "1,2", "1,1", "2,1", "2,2", "0,2", "0,4", "3,1", "3,2", "off", "1,2"):
    best, med, ng, to = run(dG, ddi, dS, m, n, knob, reps)
    S = np.empty((n, n), order="F")
    _capi.check(L.mi355kkt_memcpy_d2h(S.ctypes.data_as(C.c_void_p), dS.ptr, n * n * 8), "d2h")
    S = lower(S)
    if base is None:
        base = S
        sub = slice(0, 1024)
        ref = (G[:, sub] * (di * di)[:, None]).T @ G[:, sub]
        err = np.max(np.abs(S[sub, sub] - lower(ref))) / np.max(np.abs(ref))
        note = "leading 1024 x 1024 vs NumPy %.2e" % err
    else:
        note = "max rel diff vs free-running %.2e" % (np.max(np.abs(S - base)) / np.max(np.abs(base)))
    print("n=%d K=%d sync=%s: best %.3f median %.3f ms  %.2f TF/s  groups %d timeouts %d  %s" % (
        n, m, knob, best, med, m * float(n) * n / best / 1e9, ng, to, note), flush=True)

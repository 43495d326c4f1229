"""Developer probe: per-tile timeline of the persistent tile Cholesky (shader-clock stamps)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi
L = _capi.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NT = (n + 127) // 128
ntiles = NT * (NT + 1) // 2
rng = np.random.default_rng(0)
B = rng.standard_normal((n, n)) / np.sqrt(n)
S = np.asfortranarray(B.T @ B + np.eye(n))
ts = _capi.DeviceBuffer.from_array(np.zeros(ntiles * 8))
_capi.check(L.mi355kkt_debug_tile_ts(C.c_void_p(ts.ptr)), "ts")
for rep in range(2):
    dS = _capi.DeviceBuffer.from_array(S)
    ms, info = C.c_float(), C.c_int()
    _capi.check(L.mi355kkt_op_potrf(dS.ptr, n, n, C.byref(info), C.byref(ms)), "potrf")
t = ts.to_array((ntiles, 8), dtype="int64", order="C")
_capi.check(L.mi355kkt_debug_tile_ts(None), "off")
t0 = t[0, 0]
print("n %d: %.3f ms, info %d; columns: stamps in shader clocks relative to the first ticket" % (n, ms.value, info.value))
idx = {}
k = 0
for j in range(NT):
    for i in range(j, NT):
        idx[(i, j)] = k
        k += 1
print("col | diag: last-k accumulate  potf2(+dump)  publish | sub-diag: wait-for-Ljj  stage+trsm  publish   (shader clocks; per-CU counters, so only differences inside one tile mean anything)")
for j in range(NT):
    d = t[idx[(j, j)]]
    line = "%3d | %8d %8d %8d" % (j, (d[1] - d[5]) if j else 0, d[3] - d[1], d[4] - d[3])
    if j + 1 < NT:
        o = t[idx[(j + 1, j)]]
        line += " | %8d %8d %8d" % (o[2] - o[1], o[3] - o[2], o[4] - o[3])
    print(line)

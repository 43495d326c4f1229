import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi
L = _capi.lib()
n = 4096
rng = np.random.default_rng(0)
B = rng.standard_normal((n, n)) / np.sqrt(n)
A = B @ B.T + np.eye(n)
for mask in (0, 1, 2, 4, 8, 15):
    L.mi355kkt_debug_potf2_skip(mask)
    ts = []
    for r in range(3):
        dA = _capi.DeviceBuffer.from_array(A)
        ms, info = C.c_float(), C.c_int()
        L.mi355kkt_op_potrf(dA.ptr, n, n, C.byref(info), C.byref(ms))
        ts.append(ms.value)
    print("skip mask %2d: potrf(%d) %.3f ms" % (mask, n, min(ts)))
L.mi355kkt_debug_potf2_skip(0)

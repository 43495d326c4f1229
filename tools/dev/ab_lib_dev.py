"""Developer probe: the same small set of timings from ANY build of the library (raw ctypes, only entry points that exist since
round 1), one process per library, for same-box A/B comparisons:   python tools/dev/ab_lib_dev.py <path to .so> [label]"""
import ctypes as C
import sys
import time

import numpy as np

L = C.CDLL(sys.argv[1], mode=C.RTLD_GLOBAL)
label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
vp, dbl, i64 = C.c_void_p, C.c_double, C.c_int64
L.mi355kkt_dev_malloc.argtypes = [C.POINTER(vp), C.c_size_t]
L.mi355kkt_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
L.mi355kkt_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
L.mi355kkt_op_potrf.argtypes = [vp, i64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]
L.mi355kkt_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]
L.mi355kkt_set_G_dense.argtypes = [vp, vp, i64]
L.mi355kkt_set_H_device.argtypes = [vp, vp, i64]
L.mi355kkt_solve_device.argtypes = [vp, vp, vp, vp]
L.mi355kkt_sync.argtypes = [vp]
L.mi355kkt_get_timings.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
L.mi355kkt_destroy.argtypes = [vp]
L.mi355kkt_dev_free.argtypes = [vp]


class Scaling(C.Structure):
    _fields_ = [(k, vp) for k in ("d", "di", "v", "beta", "r", "rti")]


L.mi355kkt_factor_device.argtypes = [vp, C.POINTER(Scaling)]


def dev(a):
    p = vp()
    assert L.mi355kkt_dev_malloc(C.byref(p), a.nbytes) == 0
    assert L.mi355kkt_memcpy_h2d(p, a.ctypes.data, a.nbytes) == 0
    return p


rng = np.random.default_rng(0)
for n in (1024, 2048, 4096, 8192):
    B = rng.standard_normal((n, n)) / np.sqrt(n)
    S = np.asfortranarray(B.T @ B + np.eye(n))
    ms, info, best = C.c_float(0), C.c_int(0), 1e9
    for rep in range(5):
        d = dev(S)
        assert L.mi355kkt_op_potrf(d, n, n, C.byref(info), C.byref(ms)) == 0 and info.value == 0
        if rep:
            best = min(best, ms.value)
        L.mi355kkt_dev_free(d)
    print("%s potrf n=%d: %.3f ms" % (label, n, best))
for n, m in ((2048, 2048), (8192, 1024)):
    G = np.asfortranarray(rng.standard_normal((m, n)))
    P = np.asfortranarray(np.eye(n) * 2.0)
    h = vp()
    assert L.mi355kkt_create(C.byref(h), 0, 0, n, 0, m, 0, None, 0, None) == 0
    assert L.mi355kkt_set_G_dense(h, G.ctypes.data, m) == 0
    dP, ddi = dev(P), dev(rng.uniform(0.5, 2.0, m))
    assert L.mi355kkt_set_H_device(h, dP, n) == 0
    sc = Scaling()
    sc.di = ddi
    dx, dz, dy = dev(rng.standard_normal(n)), dev(rng.standard_normal(m)), dev(np.zeros(1))
    tm = (C.c_float * 6)()
    fs, ss = [], []
    for rep in range(8):
        assert L.mi355kkt_factor_device(h, C.byref(sc)) == 0
        assert L.mi355kkt_solve_device(h, dx, dy, dz) == 0
        assert L.mi355kkt_sync(h) == 0
        L.mi355kkt_get_timings(h, tm, 6)
        if rep >= 2:
            fs.append(tm[1]); ss.append(tm[4])
    t = time.perf_counter()
    for rep in range(20):
        L.mi355kkt_solve_device(h, dx, dy, dz)
    L.mi355kkt_sync(h)
    t = (time.perf_counter() - t) / 20
    print("%s handle n=%d m=%d: potrf %.3f ms, solve %.3f ms (event), %.3f ms (wall, 20 back to back)" % (label, n, m, min(fs), min(ss), 1e3 * t))
    L.mi355kkt_destroy(h)

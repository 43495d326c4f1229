"""Developer probe (round 4): the k-split of the SYRK's last (only) round at small n.  n = 2048, K = 8192 (the SOCP step): 136 tiles
on 512 slots; the plan splits them 3 ways (408 workgroups).  Sweeps the split factor through the knob MI355KKT_SYRK_SPLIT and
times mi355kkt_op_syrk_scaled (plan rebuilt per setting by alternating the shape)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cvxopt_amd import _capi

L = _capi.lib()
shapes = [(2048, 8192), (1024, 8192), (4096, 8192), (2048, 4096)]
rng = np.random.default_rng(0)
for n, m in shapes:
    G = _capi.DeviceBuffer.from_array(np.asfortranarray(rng.standard_normal((m, n))))
    S = _capi.DeviceBuffer(8 * n * n)
    G2 = _capi.DeviceBuffer.from_array(np.asfortranarray(rng.standard_normal((64, 128))))
    S2 = _capi.DeviceBuffer(8 * 128 * 128)
    ms = C.c_float(0)
    ref = None
    for split in (None, 2, 3, 4, 5, 6, 8, 12):
        _capi.set_knob("MI355KKT_SYRK_SPLIT", split)
        # another shape in between: the operator caches its plan per shape
        _capi.check(L.mi355kkt_op_syrk_scaled(C.c_void_p(G2.ptr), 64, 64, 128, None, None, 0, C.c_void_p(S2.ptr), 128, C.byref(ms)), "syrk")
        best = 1e9
        for rep in range(6):
            _capi.check(L.mi355kkt_op_syrk_scaled(C.c_void_p(G.ptr), m, m, n, None, None, 0, C.c_void_p(S.ptr), n, C.byref(ms)), "syrk")
            if rep:
                best = min(best, ms.value)
        out = S.to_array((n, n))
        if ref is None:
            ref = np.tril(out)
            err = 0.0
        else:
            err = float(np.max(np.abs(np.tril(out) - ref)) / np.max(np.abs(ref)))
        print("n=%d K=%d split=%s: %.3f ms  (%.1f TF/s)  max rel diff vs default plan %.1e" % (n, m, split, best, m * n * n / best / 1e9, err))
    _capi.set_knob(None, None)
    for b in (G, S, G2, S2):
        b.free()

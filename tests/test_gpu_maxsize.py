"""The largest G the reference's `matrix` type can hold (nrows * ncols <= INT_MAX, src/C/dense.c:152-155) through the C ABI
with device-resident inputs: n = 16384, m = 131071 (m n = 2^31 - 16384 doubles = 17.2 GB), 4 x the headline's n and 8 x its
m.  No CPU oracle finishes at this size, so the check is the size-independent property of the path — the KKT equations
(misc.py:1499-1503) — evaluated by an independent implementation (rocBLAS through torch) on the same device buffers:

    H ux + G' uz = bx,      G ux - W^2 uz = bz,      returned x = ux, z = W uz     (W = diag(d), LP cone, p = 0).

The row count is odd (SYRK k-tail, GEMV row tails) and every index product of the engine crosses 2^31 bytes by far."""
import ctypes as C

import numpy as np
import pytest

from cvxopt_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,m", [(16384, 131071)])
def test_kkt_equations_at_the_reference_matrix_size_limit(n, m):
    import torch
    assert n * m <= 2**31 - 1                      # the reference could still hold this G
    free, _total = torch.cuda.mem_get_info()
    if free < 40e9:
        pytest.skip("needs ~30 GB of free HBM")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(2024)
    # G is m x n column-major = a contiguous (n, m) tensor; written in slabs so that no second 17 GB temporary exists
    Gt = torch.empty((n, m), dtype=torch.float64, device=dev)
    for j0 in range(0, n, 2048):
        Gt[j0:j0 + 2048].normal_(generator=g)
    B = torch.randn((n, 32), dtype=torch.float64, device=dev, generator=g)
    H = B @ B.T / 32.0
    H.diagonal().add_(1.0)                          # symmetric positive definite, both triangles valid
    di = 0.5 + torch.rand(m, dtype=torch.float64, device=dev, generator=g)
    bx = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    bz = torch.randn(m, dtype=torch.float64, device=dev, generator=g)
    x, z = bx.clone(), bz.clone()
    y = torch.zeros(1, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    L = _capi.lib()
    h = C.c_void_p()
    q = (C.c_int * 1)(0)
    _capi.check(L.mi355kkt_create(C.byref(h), 0, _capi.CHOL2, n, 0, m, 0, q, 0, q), "mi355kkt_create")
    try:
        ptr = lambda t: C.cast(C.c_void_p(t.data_ptr()), _capi.c_double_p)
        _capi.check(L.mi355kkt_set_G_device(h, ptr(Gt), m), "set_G_device")
        _capi.check(L.mi355kkt_set_H_device(h, ptr(H), n), "set_H_device")
        sc = _capi.Scaling()
        sc.di = ptr(di)
        _capi.check(L.mi355kkt_factor_device(h, C.byref(sc)), "factor_device")
        _capi.check(L.mi355kkt_solve_device(h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_void_p(z.data_ptr())),
                    "solve_device")
        _capi.check(L.mi355kkt_sync(h), "sync")
    finally:
        L.mi355kkt_destroy(h)
    uz = z * di                                      # uz = W^-1 (W uz)
    r1 = H @ x + Gt @ uz - bx                        # Gt @ . = G' .
    r2 = Gt.T @ x - uz / (di * di) - bz
    s1 = float(r1.abs().max() / max(1.0, float(bx.abs().max())))
    s2 = float(r2.abs().max() / max(1.0, float(bz.abs().max())))
    assert np.isfinite(s1) and np.isfinite(s2)
    assert s1 < 1e-9 and s2 < 1e-9, (s1, s2)
    # and the solution is not trivially small / large: |ux| ~ |S^-1 (bx + G' D^2 bz)|
    assert 1e-6 < float(x.abs().max()) < 1e3

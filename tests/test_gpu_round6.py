"""Round 6 parity corners (VERDICT r5 "next round" item 5).

(a) ArithmeticError(info): the reference's lapack.potrf raises ArithmeticError with LAPACK's `info` -- the order of the first
    leading minor that is not positive definite (/root/reference/src/C/lapack.c:32-34, :1508-1521) -- and kkt_chol2 lets it
    through (misc.py:1440-1447).  The device Cholesky promises the same 1-based column; here the VALUE is compared with the real
    reference's, for a first bad pivot at the start, on both sides of a 128-column tile boundary, deep inside a large matrix and
    in a ragged last tile, through the raw kernel entry point, through the hook, per problem of a batch, and (as the permuted
    column of the engine's own fill-reducing order) inside a sparse front.
(b) the 512-row all-CU triangular solves (trsv512.hip) against the round-4 pair kernel and the oracle."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from cvxopt_amd import _capi, kkt, synth
from helpers import record, relerr
from oracle import kkt_oracle as ko

pytestmark = pytest.mark.gpu


def indefinite_at(n, k, seed=0):
    """symmetric A = L D L' with D = +1 except D[k-1] = -1, L unit-ish lower triangular and well conditioned: the leading minors of
    order < k are positive definite with pivots ~1, the k-th pivot is ~ -1 (nothing near rounding)"""
    rng = np.random.default_rng(seed)
    L = np.tril(rng.standard_normal((n, n)) / np.sqrt(n), -1) + np.diag(1.0 + rng.random(n))
    d = np.ones(n)
    if k:
        d[k - 1] = -1.0                                         # (k = 0: positive definite)
    A = (L * d) @ L.T
    return np.asfortranarray(0.5 * (A + A.T))


def ref_info(cvx, A):
    """info of the reference's lapack.potrf (0: positive definite)"""
    M = cvx.matrix(np.asfortranarray(A))
    try:
        cvx.lapack.potrf(M)
    except ArithmeticError as e:
        return int(e.args[0])
    return 0


def dev_potrf_info(A):
    n = A.shape[0]
    L = _capi.lib()
    dA = _capi.DeviceBuffer.from_array(np.asfortranarray(A))
    ms, info = C.c_float(), C.c_int()
    rc = L.mi355kkt_op_potrf(dA.ptr, n, n, C.byref(info), C.byref(ms))
    assert rc == 0 or rc == info.value, (rc, info.value)
    return info.value


@pytest.mark.parametrize("n,k", [(200, 1), (200, 127), (200, 128), (200, 129), (200, 200), (1600, 1), (1600, 127), (1600, 128),
                                 (1600, 129), (1600, 1500), (1600, 1537), (1600, 1600), (2048, 1920), (2048, 2048)])
def test_potrf_info_is_the_references_first_bad_minor(ref_cvxopt, n, k):
    """n = 200: the launch chain (two panels, the second ragged); n = 1600: the persistent tile kernel, ragged last tile
    (1537 = its first column, 1600 = its last); n = 2048: whole tiles"""
    A = indefinite_at(n, k, seed=n + k)
    want = ref_info(ref_cvxopt, A)
    assert want == k                                            # the construction does what it says
    assert dev_potrf_info(A) == want


@pytest.mark.parametrize("n,k", [(300, 129), (1600, 128), (1600, 1500)])
def test_hook_raises_arithmetic_error_with_the_references_info(ref_cvxopt, n, k):
    """through kkt_chol2's factor(W, H): S = H + G'W^-2 G with G = 0 rows, so S = H exactly"""
    m = 4
    H = indefinite_at(n, k, seed=7 * n + k)
    G = np.asfortranarray(np.zeros((m, n)))
    dims = {'l': m, 'q': [], 's': []}
    W = synth.random_scaling(dims, seed=0)
    f = kkt.kkt_chol2(G, dims, np.zeros((0, n)))
    try:
        with pytest.raises(ArithmeticError) as ei:
            f(W, H)
    finally:
        f.engine.close()
    assert int(ei.value.args[0]) == ref_info(ref_cvxopt, H) == k
    # and the oracle (the restated kkt_chol2) raises the same
    with pytest.raises(ArithmeticError) as eo:
        ko.KktChol2(G, dims, np.zeros((0, n))).factor(W, H)
    assert int(eo.value.args[0]) == k


def test_batched_info_per_problem_is_the_references(ref_cvxopt):
    """mi355kkt_batch_factor: info[b] of every problem equals the reference's lapack.potrf on that problem's S"""
    from cvxopt_amd.batch import BatchKkt
    B, n, m = 6, 270, 3
    ks = [0, 1, 128, 129, 257, 270]                             # 0: positive definite
    P = np.zeros((B, n, n))
    for b, k in enumerate(ks):
        P[b] = indefinite_at(n, k, seed=b)
    Gt = np.zeros((B, n, m))
    g = BatchKkt(Gt, P)
    try:
        info = g.factor(np.ones((B, m)))
    finally:
        g.close()
    want = [ref_info(ref_cvxopt, P[b]) for b in range(B)]
    assert want == ks
    assert [int(v) for v in info] == want


def test_sparse_info_is_the_first_bad_column_of_the_engines_own_order():
    """The reference's cholmod.numeric raises ArithmeticError(k) with CHOLMOD's L->minor: a column of ITS fill-reducing order
    (/root/reference/src/C/cholmod.c:428-438).  The orders differ, so the value that can be pinned is: the first non-positive pivot
    of a sequential Cholesky of S in the engine's own order (csrc/ordering.cpp via mi355kkt_op_symbolic), 1-based -- inside a
    front of the multifrontal factorisation, fronts of independent subtrees notwithstanding."""
    from test_gpu_sparse import FakeSp
    from test_sparse_symbolic_cpu import analyse, grid
    import scipy.linalg as sla
    nx = 24
    n = nx * nx
    P = grid(nx, nx).tolil()
    bad = [5, 300, 431]                                         # three indefinite diagonal entries: the smallest permuted one counts
    for i in bad:
        P[i, i] = -50.0
    P = sp.csc_matrix(P)
    m = 2 * n
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc() * 1e-3
    dims = {'l': m, 'q': [], 's': []}
    W = {'di': np.ones(m), 'd': np.ones(m), 'v': [], 'beta': [], 'r': [], 'rti': [], 'dnl': np.zeros(0), 'dnli': np.zeros(0)}
    f = kkt.kkt_chol2(FakeSp(G), dims, np.zeros((0, n)))
    try:
        with pytest.raises(ArithmeticError) as ei:
            f(W, FakeSp(sp.tril(P)))
        assert f.engine._mode == "sparse"
    finally:
        f.engine.close()
    perm, _, _, _ = analyse(G, P)
    S = (P + G.T @ G).toarray()[np.ix_(perm, perm)]
    _, info = sla.lapack.dpotrf(np.asfortranarray(S), lower=1)
    assert info > 0 and int(ei.value.args[0]) == info


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_batch_over_ipc_without_a_collective(world):
    """ShardedBatch(transport="ipc"): `world` processes -- sharing this box's GPU(s) -- pull their shards from
    the root's exported buffers and push their results into its result buffer; per-problem equality with the single-GPU solve,
    uneven and empty shards, nsub = 1, 2, 4 (tests/run_batch_sharded_ipc.py)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                          "--master-addr", "127.0.0.1", "--master-port", str(29540 + world),
                          os.path.join(here, "run_batch_sharded_ipc.py")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "SHARDED_IPC_OK world=%d" % world in out.stdout


# ---- (b) the 512-row all-CU triangular solves -----------------------------------------------------------------------------------
def _engine_solves(n, m, p, wide, rhs, spread=1.5):
    """solve() of one dense engine through the hook with the wide solves on / off (test knob MI355KKT_TRSV_WIDE)"""
    pr = synth.dense_qp(n, m, seed=n + 3, p=p)
    G, P, dims = pr['G'], pr['P'], pr['dims']
    A = pr.get('A', np.zeros((0, n)))
    W = synth.random_scaling(dims, seed=5, spread=spread)
    _capi.set_knob("MI355KKT_TRSV_WIDE", wide)
    f = kkt.kkt_chol2(G, dims, A)
    try:
        s = f(W, P)
        out = []
        for bx, by, bz in rhs:
            x, y, z = bx.copy(), by.copy(), bz.copy()
            s(x, y, z)
            out.append((x, y, z))
    finally:
        f.engine.close()
        _capi.set_knob("MI355KKT_TRSV_WIDE", None)
    return pr, W, out


@pytest.mark.parametrize("n,p", [(1024, 0), (1152, 0), (1920, 4), (2048, 0), (2176, 0), (3072, 9), (4096, 0), (4096, 16), (4224, 0),
                                 (8192, 16)])
def test_wide_triangular_solves_match_the_pair_kernel_and_the_oracle(n, p):
    """orders that are multiples of 128 from 1024 up: whole 512-blocks (1024, 2048, 3072, 4096), ragged last block of 128 / 256 / 384
    rows (1152, 2176 / 1920 / 4224), 8 rows per workgroup (n <= 2048) and 16; with and without the Schur complement of equality
    constraints behind the solves.  Same answer as the round-4 two-sweep kernel to 1e-9, KKT residual no worse than 3 x the oracle's
    (LAPACK on the CPU), bit for bit the same under repetition."""
    m = n + 64 if n < 8192 else 1024                            # (8192: the oracle's dense m x n products stay in seconds)
    rng = np.random.default_rng(n)
    rhs = [(rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)) for _ in range(2)]
    pr, W, wide = _engine_solves(n, m, p, 1, rhs)
    _, _, again = _engine_solves(n, m, p, 1, rhs)
    _, _, pair = _engine_solves(n, m, p, 0, rhs)
    G, P, dims = pr['G'], pr['P'], pr['dims']
    A = pr.get('A', np.zeros((0, n)))
    oracle = ko.KktChol2(G, dims, A).factor(W, P)
    worst_pair, res_w, res_p, res_o = 0.0, 0.0, 0.0, 0.0
    for (bx, by, bz), w3, a3, p3 in zip(rhs, wide, again, pair):
        for u, v in zip(w3, a3):
            assert np.array_equal(u, v)                         # deterministic
        worst_pair = max(worst_pair, relerr(w3[0], p3[0]), relerr(w3[2], p3[2]))
        xo, yo, zo = bx.copy(), by.copy(), bz.copy()
        oracle(xo, yo, zo)
        res_w = max(res_w, ko.kkt_residual(P, A, G, W, dims, bx, by, bz, *w3))
        res_p = max(res_p, ko.kkt_residual(P, A, G, W, dims, bx, by, bz, *p3))
        res_o = max(res_o, ko.kkt_residual(P, A, G, W, dims, bx, by, bz, xo, yo, zo))
    record("round6_wide_%d_%d" % (n, p), wide_vs_pair=worst_pair, kkt_residual_wide=res_w, kkt_residual_pair=res_p,
           kkt_residual_oracle=res_o)
    assert worst_pair < 1e-9, worst_pair
    assert res_w <= max(1e-12, 3.0 * res_o), (res_w, res_o)


@pytest.mark.parametrize("n,p", [(1100, 0), (1500, 7), (2100, 0), (4200, 3)])
def test_wide_solves_for_orders_that_are_not_multiples_of_128(n, p):
    """any order >= 1024 (the dense engine's default, and the shape the sparse engine's dense root takes): partial last slice of rows,
    ragged last 128-block in the formation of the 512 x 512 inverses -- against the round-4 kernels and the oracle"""
    m = n + 64
    rng = np.random.default_rng(n)
    rhs = [(rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)) for _ in range(2)]
    pr, W, wide = _engine_solves(n, m, p, 1, rhs)
    _, _, again = _engine_solves(n, m, p, 1, rhs)
    _, _, base = _engine_solves(n, m, p, 0, rhs)
    G, P, dims = pr['G'], pr['P'], pr['dims']
    A = pr.get('A', np.zeros((0, n)))
    oracle = ko.KktChol2(G, dims, A).factor(W, P)
    for (bx, by, bz), w3, a3, b3 in zip(rhs, wide, again, base):
        for u, v in zip(w3, a3):
            assert np.array_equal(u, v)
        assert relerr(w3[0], b3[0]) < 1e-9 and relerr(w3[2], b3[2]) < 1e-9
        xo, yo, zo = bx.copy(), by.copy(), bz.copy()
        oracle(xo, yo, zo)
        res_w = ko.kkt_residual(P, A, G, W, dims, bx, by, bz, *w3)
        res_o = ko.kkt_residual(P, A, G, W, dims, bx, by, bz, xo, yo, zo)
        assert res_w <= max(1e-12, 3.0 * res_o), (res_w, res_o)


def test_wide_solves_two_slices_per_workgroup_at_8192():
    """n = 8192: 512 slices of 16 rows on 256 compute units -- every workgroup takes a second slice when its first is done"""
    n, m = 8192, 1024
    rng = np.random.default_rng(1)
    rhs = [(rng.standard_normal(n), np.zeros(0), rng.standard_normal(m))]
    pr, W, wide = _engine_solves(n, m, 0, 1, rhs, spread=1.0)
    _, _, pair = _engine_solves(n, m, 0, 0, rhs, spread=1.0)
    G, P, dims = pr['G'], pr['P'], pr['dims']
    (bx, by, bz), w3, p3 = rhs[0], wide[0], pair[0]
    assert relerr(w3[0], p3[0]) < 1e-9 and relerr(w3[2], p3[2]) < 1e-9
    A = np.zeros((0, n))
    res_w = ko.kkt_residual(P, A, G, W, dims, bx, by, bz, *w3)
    res_p = ko.kkt_residual(P, A, G, W, dims, bx, by, bz, *p3)
    record("round6_wide_8192", kkt_residual_wide=res_w, kkt_residual_pair=res_p)
    assert res_w <= max(1e-12, 2.0 * res_p), (res_w, res_p)


# ---- (c) kkt_qr where QR and normal equations differ ---------------------------------------------------------------------------
def _illcond_lp(n, m, eps, seed):
    """an LP whose G has two nearly dependent columns: sigma_min(G) ~ eps, so cond(W^-T G) >= ~1 / eps from the first iteration"""
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((m, n))
    G[:, n - 1] = G[:, 0] + eps * rng.standard_normal(m)
    x0 = rng.standard_normal(n)
    h = G @ x0 + rng.uniform(0.5, 2.0, m)
    c = -G.T @ rng.uniform(0.5, 2.0, m)
    return c, G, h


@pytest.mark.parametrize("eps", [1e-3, 1e-5, 1e-6, 1e-7, 1e-8])
@pytest.mark.parametrize("device_loop", [True, False])
def test_kkt_qr_mapping_follows_the_references_qr_where_cholesky_alone_does_not(ref_cvxopt, eps, device_loop):
    """VERDICT r5 item 5(b).  The reference's kkt_qr (misc.py:1570-1699) factors W^-T G by QR; the backend maps 'qr' onto its reduced
    Cholesky engine.  With cond(G) = 3e5 / 3e6 the reference's own 'chol' loses digits or stalls and so did the plain mapping of
    rounds 1-5 (tests/run_qr_cond_probe.py, profiles/r06_kkt_qr_conditioning.txt).  The mapping now (i) refines its solves against
    the 3 x 3 system when the factor shows (max L_ii / min L_ii)^2 >= 1e8, (ii) repairs the factor by CholeskyQR2 from 1e10, and
    (iii) where even chol(Gs'Gs) breaks down (cond(G) = 3e8) starts from a shifted Cholesky and takes two repair passes (shifted
    CholeskyQR3): the reference's status, iteration count and objectives at every cond(G) from 3e3 to 3e8, through the device loop
    and through the reference's host driver."""
    import cvxopt_amd.solvers as gs
    cvx = ref_cvxopt
    c, G, h = _illcond_lp(40, 120, eps, 1)
    M = lambda a: cvx.matrix(np.asfortranarray(np.atleast_2d(a.T).T if a.ndim == 1 else a))
    ref = cvx.solvers.conelp(M(c), M(G), M(h), kktsolver='qr')
    got = gs.conelp(M(c), M(G), M(h), kktsolver='qr', device_loop=device_loop)
    assert ref['status'] == 'optimal'
    assert got['status'] == ref['status'] and got['iterations'] == ref['iterations']
    for k in ('primal objective', 'dual objective'):
        assert abs(got[k] - ref[k]) <= 1e-8 * abs(ref[k]), (k, got[k], ref[k])
    record("round6_kkt_qr_eps%g_%s" % (eps, "dev" if device_loop else "host"), pobj=got['primal objective'],
           pobj_ref=ref['primal objective'], gap=got['gap'], gap_ref=ref['gap'])


@pytest.mark.parametrize("device_loop", [True, False])
def test_kkt_qr_mapping_beyond_the_references_own_qr(ref_cvxopt, device_loop):
    """cond(G) = 3e9: the reference's 'qr' runs into its iteration limit (primal infeasibility 1e32), its pivoted 'ldl' still solves
    the problem in 8 iterations -- and so does the backend's 'qr' (shifted CholeskyQR3 + refinement), with that objective"""
    import cvxopt_amd.solvers as gs
    cvx = ref_cvxopt
    c, G, h = _illcond_lp(40, 120, 1e-9, 1)
    M = lambda a: cvx.matrix(np.asfortranarray(np.atleast_2d(a.T).T if a.ndim == 1 else a))
    ref = cvx.solvers.conelp(M(c), M(G), M(h), kktsolver='ldl')
    got = gs.conelp(M(c), M(G), M(h), kktsolver='qr', device_loop=device_loop)
    assert ref['status'] == 'optimal' and got['status'] == 'optimal' and got['iterations'] == ref['iterations']
    assert abs(got['primal objective'] - ref['primal objective']) <= 1e-8 * abs(ref['primal objective'])


def test_sparse_dense_root_through_the_wide_solves():
    """The sparse engine's dense root (one big supernode without rows below it, factored by the dense tile kernel) takes the 512-row
    solves when its width is a multiple of 128 from 1024 up: a 32 x 32 x 65 grid, whose nested-dissection root is the 32 x 32 middle
    plane (1024 columns).  Same solution with the wide solves on and off, and against a sparse LU of the same matrix."""
    import scipy.sparse.linalg as spla
    from test_gpu_sparse import FakeSp
    nx, ny, nz = 32, 32, 65
    ex = lambda k: sp.diags([-np.ones(k - 1), 2 * np.ones(k), -np.ones(k - 1)], [-1, 0, 1])
    P = (sp.kron(sp.kron(sp.eye(nz), sp.eye(ny)), ex(nx)) + sp.kron(sp.kron(sp.eye(nz), ex(ny)), sp.eye(nx)) +
         sp.kron(sp.kron(ex(nz), sp.eye(ny)), sp.eye(nx)) + 1e-2 * sp.eye(nx * ny * nz)).tocsc()
    n = P.shape[0]
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()
    dims = {'l': 2 * n, 'q': [], 's': []}
    W = synth.random_scaling(dims, seed=2, spread=1.0)
    rng = np.random.default_rng(0)
    bx, bz = rng.standard_normal(n), rng.standard_normal(2 * n)
    sols = {}
    for wide in (1, 0):
        _capi.set_knob("MI355KKT_TRSV_WIDE", wide)
        f = kkt.kkt_chol2(FakeSp(G), dims, np.zeros((0, n)))
        try:
            x, y, z = bx.copy(), np.zeros(0), bz.copy()
            f(W, FakeSp(sp.tril(P)))(x, y, z)
            assert f.engine._mode == "sparse"
            sols[wide] = (x, z)
        finally:
            f.engine.close()
            _capi.set_knob("MI355KKT_TRSV_WIDE", None)
    di = W['di']
    S = (P + G.T @ sp.diags(di * di) @ G).tocsc()
    xs = spla.splu(S).solve(bx + G.T @ (di * di * bz))
    for wide in (1, 0):
        assert relerr(sols[wide][0], xs) < 1e-10
    assert relerr(sols[1][0], sols[0][0]) < 1e-11 and relerr(sols[1][1], sols[0][1]) < 1e-11

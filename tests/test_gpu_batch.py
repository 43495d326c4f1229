"""Batched mode on the GPU (BASELINE configs[4], scaled down): batched HIP factor/solve vs the per-problem
CPU back end, and the lock-step coneqp_batch vs individual reference runs."""
import numpy as np
import pytest

from cvxopt_amd import synth
from cvxopt_amd.batch import BatchKkt, coneqp_batch, pack_problems
from batch_helpers import NumpyBatchKkt
from helpers import load_golden, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,n,m", [(1, 64, 128), (5, 100, 230), (16, 256, 512), (3, 130, 61), (64, 512, 1024), (6, 300, 450),
                                   (2, 1100, 1300)])
def test_batched_factor_solve_matches_cpu(B, n, m):
    probs = [synth.dense_qp(n, m, seed=100 + i) for i in range(B)]
    P, q, Gt, h = pack_problems(probs)
    rng = np.random.default_rng(B + n)
    di = 10.0 ** rng.uniform(-1.5, 1.5, (B, m))
    g, c = BatchKkt(Gt, P), NumpyBatchKkt(Gt, P)
    assert np.all(g.factor(di) == 0) and np.all(c.factor(di) == 0)
    x, z = rng.standard_normal((B, n)), rng.standard_normal((B, m))
    xg, zg, xc, zc = x.copy(), z.copy(), x.copy(), z.copy()
    g.solve(xg, zg)
    c.solve(xc, zc)
    for b in range(B):
        assert relerr(xg[b], xc[b]) < 1e-8 and relerr(zg[b], zc[b]) < 1e-8, b
    g.close()


def test_batched_factor_reports_per_problem_failures():
    B, n, m = 4, 40, 20                       # rank(G) = 20 < 40 and P = 0 for problems 1 and 3
    probs = [synth.dense_qp(n, m, seed=i) for i in range(B)]
    P, q, Gt, h = pack_problems(probs)
    P[1] = 0.0
    P[3] = 0.0
    g = BatchKkt(Gt, P)
    info = g.factor(np.ones((B, m)))
    assert info[0] == 0 and info[2] == 0 and info[1] > 0 and info[3] > 0
    ref = NumpyBatchKkt(Gt, P).factor(np.ones((B, m)))
    # S is exactly rank 20: the first non-positive pivot is a rounding-level quantity (21 or 22)
    assert abs(int(info[1]) - int(ref[1])) <= 2 and abs(int(info[3]) - int(ref[3])) <= 2 and info[1] > 20
    g.close()


def test_batched_factor_reports_per_problem_failures_several_panels():
    """n = 270 (three 128-column panels, the last one narrow): a failing problem reports its first non-positive pivot and
    does not disturb the others"""
    B, n, m = 3, 270, 100                     # rank(G) = 100 < 270 and P = 0 for problem 1
    probs = [synth.dense_qp(n, m, seed=i) for i in range(B)]
    P, q, Gt, h = pack_problems(probs)
    P[1] = 0.0
    g = BatchKkt(Gt, P)
    rng = np.random.default_rng(0)
    info = g.factor(np.ones((B, m)))
    assert info[0] == 0 and info[2] == 0 and 100 < info[1] <= 103
    x, z = rng.standard_normal((B, n)), rng.standard_normal((B, m))
    xg, zg = x.copy(), z.copy()
    g.solve(xg, zg)
    c = NumpyBatchKkt(Gt[[0, 2]], P[[0, 2]])
    assert np.all(c.factor(np.ones((2, m))) == 0)
    xc, zc = x[[0, 2]].copy(), z[[0, 2]].copy()
    c.solve(xc, zc)
    assert relerr(xg[0], xc[0]) < 1e-8 and relerr(xg[2], xc[1]) < 1e-8 and relerr(zg[2], zc[1]) < 1e-8
    g.close()


def test_coneqp_batch_gpu_matches_golden_reference_run():
    g = load_golden("coneqp_qp256")
    pr = synth.dense_qp(int(g['n']), int(g['m']), seed=int(g['seed']))
    P, q, Gt, h = pack_problems([pr])
    res = coneqp_batch(P, q, Gt, h)
    assert res['status'][0] == 'optimal' and res['iterations'][0] == int(g['iterations'])
    assert abs(res['primal objective'][0] - float(g['pobj'])) <= 1e-9 * abs(float(g['pobj']))
    assert relerr(res['x'][0], g['x']) < 1e-7


def test_coneqp_batch_gpu_matches_individual_reference_runs(ref_cvxopt):
    from cvxopt import matrix, solvers
    probs = [synth.dense_qp(48, 100, seed=20 + i) for i in range(12)]
    probs[5]['h'] = probs[5]['h'] * 30.0
    probs[7]['q'] = probs[7]['q'] * 1e-3
    P, q, Gt, h = pack_problems(probs)
    res = coneqp_batch(P, q, Gt, h)
    for b, pr in enumerate(probs):
        ref = solvers.coneqp(matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']), kktsolver='chol2')
        assert res['status'][b] == ref['status']
        assert res['iterations'][b] == ref['iterations'], b
        assert abs(res['primal objective'][b] - ref['primal objective']) <= 1e-9 * max(1, abs(ref['primal objective']))
        assert relerr(res['x'][b], np.array(ref['x']).ravel()) < 1e-7


def test_resident_loop_matches_lockstep_numpy_loop():
    """mi355kkt_batch_coneqp (bookkeeping on the device) vs coneqp_batch's NumPy restatement, both on the GPU KKT
    back end: same per-problem iteration counts, iterates to rounding."""
    probs = [synth.dense_qp(64, 150, seed=300 + i) for i in range(24)]
    probs[3]['h'] = probs[3]['h'] * 50.0
    probs[11]['q'] = probs[11]['q'] * 1e-4
    probs[17]['P'] = probs[17]['P'] * 1e3
    P, q, Gt, h = pack_problems(probs)
    host = coneqp_batch(P, q, Gt, h)
    dev = coneqp_batch(P, q, Gt, h, resident=True)
    assert list(dev['status']) == list(host['status'])
    assert np.array_equal(dev['iterations'], host['iterations'])
    assert dev['lockstep iterations'] == host['iterations'].max()
    for b in range(len(probs)):
        assert relerr(dev['x'][b], host['x'][b]) < 1e-9, b
        assert relerr(dev['s'][b], host['s'][b]) < 1e-7 and relerr(dev['z'][b], host['z'][b]) < 1e-7, b
    assert np.allclose(dev['primal objective'], host['primal objective'], rtol=1e-10, atol=1e-12)
    assert np.allclose(dev['dual objective'], host['dual objective'], rtol=1e-9, atol=1e-10)
    assert np.allclose(dev['gap'], host['gap'], rtol=1e-6, atol=1e-14)


def test_resident_loop_matches_individual_reference_runs(ref_cvxopt):
    from cvxopt import matrix, solvers
    probs = [synth.dense_qp(40, 90, seed=60 + i) for i in range(9)]
    probs[2]['h'] = probs[2]['h'] * 20.0
    P, q, Gt, h = pack_problems(probs)
    res = coneqp_batch(P, q, Gt, h, resident=True)
    for b, pr in enumerate(probs):
        ref = solvers.coneqp(matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']), kktsolver='chol2')
        assert res['status'][b] == ref['status']
        assert res['iterations'][b] == ref['iterations'], b
        assert abs(res['primal objective'][b] - ref['primal objective']) <= 1e-9 * max(1, abs(ref['primal objective']))
        assert abs(res['dual objective'][b] - ref['dual objective']) <= 1e-8 * max(1, abs(ref['dual objective']))
        assert relerr(res['x'][b], np.array(ref['x']).ravel()) < 1e-7
        assert relerr(res['s'][b], np.array(ref['s']).ravel()) < 1e-6
        assert relerr(res['z'][b], np.array(ref['z']).ravel()) < 1e-6


def test_resident_loop_iteration_limit_and_rank_failure():
    probs = [synth.dense_qp(32, 64, seed=i) for i in range(4)]
    P, q, Gt, h = pack_problems(probs)
    host = coneqp_batch(P, q, Gt, h, maxiters=3)
    dev = coneqp_batch(P, q, Gt, h, maxiters=3, resident=True)
    assert list(dev['status']) == list(host['status']) == ['unknown'] * 4
    assert np.array_equal(dev['iterations'], host['iterations']) and np.all(dev['iterations'] == 3)
    for b in range(4):
        assert relerr(dev['x'][b], host['x'][b]) < 1e-10
    probs = [synth.dense_qp(40, 20, seed=i) for i in range(3)]      # rank(G) < n with P = 0
    P, q, Gt, h = pack_problems(probs)
    P[1] = 0.0
    with pytest.raises(ValueError):
        coneqp_batch(P, q, Gt, h, resident=True)


def test_batchkkt_takes_problem_data_already_in_hbm():
    """torch CUDA tensors (e.g. the shard an RCCL scatter delivered) are taken by device pointer: no host round trip."""
    import torch
    probs = [synth.dense_qp(48, 100, seed=70 + i) for i in range(6)]
    P, q, Gt, h = pack_problems(probs)
    host = coneqp_batch(P, q, Gt, h, resident=True)
    Gd, Pd = torch.from_numpy(np.ascontiguousarray(Gt)).cuda(), torch.from_numpy(np.ascontiguousarray(P)).cuda()
    dev = coneqp_batch(Pd, q, Gd, h, resident=True)
    assert np.array_equal(dev['iterations'], host['iterations'])
    assert np.array_equal(dev['x'], host['x'])            # same kernels, same data: bit-identical
    with pytest.raises(TypeError):
        BatchKkt(Gd.float(), None)


def test_sharded_batch_on_rccl():
    """coneqp_batch_sharded on the real nccl (= RCCL) backend, launched like bench.py is (torch.distributed.run)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    from cvxopt_amd import _capi
    nproc = max(1, min(8, _capi.device_count()))             # every visible GPU (the round's box has one)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                          "--master-addr", "127.0.0.1", "--master-port", "29531",
                          os.path.join(here, "run_batch_sharded_nccl.py")], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "SHARDED_NCCL_OK" in out.stdout


# ---- equality constraints in the batched engine (round 2) ------------------------------------------------------------------
def _batch_eq(name):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    Gt = np.ascontiguousarray(np.transpose(g['G'], (0, 2, 1)))       # (B, n, m): the m x n blocks column-major
    At = np.ascontiguousarray(np.transpose(g['A'], (0, 2, 1)))       # (B, n, p)
    return g, Gt, At


@pytest.mark.parametrize("name,singular", [("batch_eq", False), ("batch_eq_singular", True)])
def test_batched_coneqp_with_equality_constraints_vs_reference_fixture(name, singular):
    """`mi355kkt_batch_coneqp_eq` against solvers.coneqp(P, q, G, h, A=A, b=b) run problem by problem by the real reference
    (tests/golden/make_golden_batch_eq.py); the second batch has a singular S in every problem: the S + A'A mode."""
    g, Gt, At = _batch_eq(name)
    bk = BatchKkt(Gt, np.ascontiguousarray(g['P']), At=At)
    res = bk.coneqp(g['q'], g['h'], b=g['b'])
    B = Gt.shape[0]
    assert all(s == 'optimal' for s in res['status'])
    assert np.array_equal(res['iterations'], g['iterations'])
    for k in range(B):
        assert abs(res['primal objective'][k] - g['pobj'][k]) <= 1e-8 * max(1.0, abs(g['pobj'][k])), k
        assert abs(res['dual objective'][k] - g['dobj'][k]) <= 1e-8 * max(1.0, abs(g['dobj'][k])), k
        assert relerr(res['x'][k], g['x'][k]) < 1e-6, k
        assert relerr(res['y'][k], g['y'][k]) < 1e-5, k
        assert relerr(res['z'][k], g['z'][k]) < 1e-5 and relerr(res['s'][k], g['s'][k]) < 1e-5, k
    bk.close()


def test_batched_solve_with_equality_constraints_solves_the_kkt_system():
    """factor + solve of the batched engine at the hook level: residual of [P A' G'; A 0 0; G 0 -W'W][ux; uy; uz] = [bx; by; bz]"""
    g, Gt, At = _batch_eq("batch_eq")
    B, n, m = Gt.shape
    p = At.shape[2]
    rng = np.random.default_rng(3)
    di = 10.0 ** rng.uniform(-1.0, 1.0, (B, m))
    bk = BatchKkt(Gt, np.ascontiguousarray(g['P']), At=At)
    assert np.all(bk.factor(di) == 0)
    bx, by, bz = rng.standard_normal((B, n)), rng.standard_normal((B, p)), rng.standard_normal((B, m))
    x, y, z = bx.copy(), by.copy(), bz.copy()
    bk.solve(x, z, y)
    for k in range(B):
        P, G, A = g['P'][k], g['G'][k], g['A'][k]
        uz = z[k] * di[k]                                        # returned z = W uz, W = diag(1 / di)
        r1 = P @ x[k] + A.T @ y[k] + G.T @ uz - bx[k]
        r2 = A @ x[k] - by[k]
        r3 = G @ x[k] - uz / di[k] ** 2 - bz[k]
        scale = max(np.linalg.norm(bx[k]), np.linalg.norm(bz[k]), 1.0) * np.linalg.norm(di[k]) ** 2
        assert max(np.linalg.norm(r1), np.linalg.norm(r2), np.linalg.norm(r3)) <= 1e-10 * scale, k
    bk.close()


def test_batched_rank_deficient_A_is_reported_per_problem():
    g, Gt, At = _batch_eq("batch_eq")
    B, n, m = Gt.shape
    At = At.copy()
    At[2, :, 4] = 0.0                                           # problem 2: a zero row of A -> K_2 has an exactly zero pivot
    bk = BatchKkt(Gt, np.ascontiguousarray(g['P']), At=At)
    info = bk.factor(np.ones((B, m)))
    assert info[2] == n + 5 and np.all(np.delete(info, 2) == 0)       # reported as n + pivot, like the single-problem engine
    bk.close()


# ---- second-order cones in the batched engine (round 2) ----------------------------------------------------------------------
def _batch_q(name):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    Gt = np.ascontiguousarray(np.transpose(g['G'], (0, 2, 1)))       # (B, n, cdim): the cdim x n blocks column-major
    At = np.ascontiguousarray(np.transpose(g['A'], (0, 2, 1)))       # (B, n, p)
    dims = {'l': int(g['dims_l']), 'q': [int(k) for k in g['dims_q']], 's': []}
    return g, Gt, At, dims


def test_batched_factor_solve_with_second_order_cones_vs_reference_kkt_chol():
    """`mi355kkt_batch_factor_cones` + `mi355kkt_batch_solve` against the reference's misc.kkt_chol(G, dims, A)(W, P)(x, y, z)
    problem by problem, each with its own Nesterov-Todd scaling from misc.compute_scaling
    (tests/golden/make_golden_batch_q.py): cones of 3, 5, 12 and 40 rows next to an 'l' block."""
    g, Gt, _At, dims = _batch_q("batch_q")
    bk = BatchKkt(Gt, np.ascontiguousarray(g['P']), dims=dims)
    info = bk.factor_cones(g['Wdi'], g['Wv'], g['Wbeta'])
    assert np.all(info == 0)
    x, z = np.ascontiguousarray(g['bx']).copy(), np.ascontiguousarray(g['bz']).copy()
    bk.solve(x, z)
    for k in range(Gt.shape[0]):
        assert relerr(x[k], g['ux'][k]) < 1e-10, k
        assert relerr(z[k], g['uz'][k]) < 1e-10, k
    # a second factorisation with other scalings on the same handle, then the first ones again: nothing is cached
    rng = np.random.default_rng(0)
    perm = rng.permutation(Gt.shape[0])
    bk.factor_cones(g['Wdi'][perm], g['Wv'][perm], g['Wbeta'][perm])
    bk.factor_cones(g['Wdi'], g['Wv'], g['Wbeta'])
    x2, z2 = np.ascontiguousarray(g['bx']).copy(), np.ascontiguousarray(g['bz']).copy()
    bk.solve(x2, z2)
    assert np.array_equal(x, x2) and np.array_equal(z, z2)
    bk.close()


@pytest.mark.parametrize("name", ["batch_q", "batch_q_eq"])
def test_batched_coneqp_with_second_order_cones_vs_reference_fixture(name):
    """`mi355kkt_batch_coneqp_eq` on a batch with second-order cones against solvers.coneqp(P, q, G, h, dims[, A, b]) run
    problem by problem by the real reference (default refinement 1 with 'q' blocks): the problems stop after different
    numbers of iterations (10 .. 13), so the freeze of finished problems is exercised too."""
    g, Gt, At, dims = _batch_q(name)
    p = At.shape[2]
    bk = BatchKkt(Gt, np.ascontiguousarray(g['P']), At=At if p else None, dims=dims)
    res = bk.coneqp(g['q'], g['h'], b=g['b'] if p else None)
    B = Gt.shape[0]
    assert all(s == 'optimal' for s in res['status'])
    assert np.array_equal(res['iterations'], g['iterations'])
    for k in range(B):
        assert abs(res['primal objective'][k] - g['pobj'][k]) <= 1e-8 * max(1.0, abs(g['pobj'][k])), k
        assert abs(res['dual objective'][k] - g['dobj'][k]) <= 1e-8 * max(1.0, abs(g['dobj'][k])), k
        assert relerr(res['x'][k], g['x'][k]) < 1e-6, k
        if p:
            assert relerr(res['y'][k], g['y'][k]) < 1e-5, k
        assert relerr(res['z'][k], g['z'][k]) < 1e-5 and relerr(res['s'][k], g['s'][k]) < 1e-5, k
    # the same batch again on the same handle (state reuse), and a batch of ONE problem (the single-problem control flow)
    res2 = bk.coneqp(g['q'], g['h'], b=g['b'] if p else None)
    assert np.array_equal(res2['iterations'], res['iterations']) and np.array_equal(res2['x'], res['x'])
    bk.close()
    b1 = BatchKkt(Gt[:1].copy(), np.ascontiguousarray(g['P'][:1]), At=At[:1].copy() if p else None, dims=dims)
    r1 = b1.coneqp(g['q'][:1], g['h'][:1], b=g['b'][:1] if p else None)
    assert r1['status'][0] == 'optimal' and r1['iterations'][0] == g['iterations'][0]
    assert relerr(r1['x'][0], g['x'][0]) < 1e-6
    b1.close()


def test_batched_cones_match_the_single_problem_device_loop():
    """every problem of the cone batch through `cvxopt_amd.coneqp_device` (the single-problem loop: same kernels, one
    workgroup, other factorisation kernels): same iteration counts, iterates to rounding"""
    import cvxopt_amd
    g, Gt, _At, dims = _batch_q("batch_q")
    bk = BatchKkt(Gt, np.ascontiguousarray(g['P']), dims=dims)
    res = bk.coneqp(g['q'], g['h'])
    bk.close()
    for k in range(3):
        sol = cvxopt_amd.coneqp_device(g['P'][k], g['q'][k], g['G'][k], g['h'][k], dims)
        assert sol['status'] == 'optimal' and sol['iterations'] == res['iterations'][k]
        assert relerr(np.asarray(sol['x']).ravel(), res['x'][k]) < 1e-8, k


def test_batched_cones_of_8_and_4_rows_vector_path_vs_oracle_and_single_loop():
    """cones of 8 and 4 rows on even offsets with an even cdim take the 16-byte path of the batched scaling kernel
    (the SOCP class of BASELINE configs[2]): hook level against the NumPy oracle of misc.kkt_chol with a different
    Nesterov-Todd scaling per problem, whole loop against the single-problem device loop"""
    import cvxopt_amd
    from oracle import kkt_oracle as ko
    B, n = 5, 14
    dims = {'l': 4, 'q': [8, 4, 8, 4], 's': []}
    m = 28
    rng = np.random.default_rng(11)

    def interior():
        u = np.empty(m)
        u[:4] = rng.uniform(0.5, 1.5, 4)
        o = 4
        for k in dims['q']:
            t = rng.standard_normal(k - 1)
            u[o] = np.linalg.norm(t) + rng.uniform(0.5, 1.5)
            u[o + 1:o + k] = t
            o += k
        return u
    P, q, G, h = [], [], [], []
    for _ in range(B):
        Bm = rng.standard_normal((n, n))
        Pk = Bm @ Bm.T / n + 0.1 * np.eye(n)
        Gk = rng.standard_normal((m, n))
        x0 = rng.standard_normal(n)
        P.append(Pk)
        G.append(Gk)
        q.append(-(Pk @ x0 + Gk.T @ interior()))
        h.append(Gk @ x0 + interior())
    P, q, G, h = np.array(P), np.array(q), np.array(G), np.array(h)
    Gt = np.ascontiguousarray(np.transpose(G, (0, 2, 1)))
    bk = BatchKkt(Gt, P, dims=dims)
    Ws = [synth.random_scaling(dims, seed=50 + k, spread=1.0) for k in range(B)]
    di = np.array([W['di'] for W in Ws])
    v = np.array([np.concatenate(W['v']) for W in Ws])
    beta = np.array([W['beta'] for W in Ws])
    assert np.all(bk.factor_cones(di, v, beta) == 0)
    bx, bz = rng.standard_normal((B, n)), rng.standard_normal((B, m))
    x, z = bx.copy(), bz.copy()
    bk.solve(x, z)
    for k in range(B):
        xo, yo, zo = bx[k].copy(), np.zeros(0), bz[k].copy()
        ko.KktChol(G[k], dims, np.zeros((0, n))).factor(Ws[k], P[k])(xo, yo, zo)
        assert relerr(x[k], xo) < 1e-10 and relerr(z[k], zo) < 1e-10, k
    res = bk.coneqp(q, h)
    bk.close()
    assert all(s == 'optimal' for s in res['status'])
    for k in range(B):
        sol = cvxopt_amd.coneqp_device(P[k], q[k], np.asfortranarray(G[k]), h[k], dims)
        assert sol['status'] == 'optimal' and sol['iterations'] == res['iterations'][k], k
        assert relerr(np.asarray(sol['x']).ravel(), res['x'][k]) < 1e-8, k

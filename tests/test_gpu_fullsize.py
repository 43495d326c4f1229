"""Full-size parity against committed reference fixtures (tests/golden/full_*.npz, written by
tests/golden/make_golden_full.py from the REAL reference in the build container): BASELINE configs[1..4] at the sizes the
bench uses, device-resident loops and hook-level runs alike.  Nothing here compares HIP with HIP or with hand-typed numbers.

Tolerances (the BOUND table below): same status and iteration count; objectives 1e-11 relative; x 1e-8 relative in the max norm
(1e-9 / 1e-7); the per-iteration table the reference prints (pcost, dcost to 5 significant digits; gap, pres, dres
to 1) to the printed precision; the Nesterov-Todd scaling of every factor() call of a hook-level run (||di||_2 at full
precision) to 1e-11 relative over the first ten iterations, 2e-8 afterwards.  The errors actually achieved are recorded."""
import contextlib
import io
import os
import re

import numpy as np
import pytest

import cvxopt_amd
from cvxopt_amd import kkt, synth
from helpers import record

pytestmark = pytest.mark.gpu
# Bounds (relative errors against the reference fixtures).  SURVEY 8(d) states 1e-9 for the objectives and 1e-7 for x.  Every bound
# below is AT LEAST that tight; the achieved errors of the GPU run of this file are in profiles/r03_parity_report.json (the
# tests write them, `record`) and are quoted on the right -- the bounds leave one to two orders of magnitude for a different
# summation order of a later kernel, not more.
BOUND = {
    'obj': 1e-11,                                  # objectives, configs 2 and 5: achieved 5e-15 .. 1e-14
    'c2_x': 1e-8, 'c2_z': 2e-8, 'c2_s': 2e-8,      # config 2 (cond ~1e10 near convergence): achieved x 1.5e-10, z / s 7e-10
    'w_early': 1e-11, 'w_late': 2e-8,              # ||di|| of every factor call: achieved <= 4e-13 (calls 0-9); after: 7.5e-10 (r3), 1.2e-8 (r4);
                                                   # the per-solve KKT residuals behind it are recorded (profiles/r05_parity_report.json)
    'c3_obj': 1e-11, 'c3_x': 1e-10,                # config 3 (SOCP): achieved 2e-13, 5e-15
    'c4_obj': 1e-11, 'c4_x': 1e-10, 'c4_z': 1e-10, # config 4 class (sparse): achieved 2e-16, 6e-16, 2e-15
    'c5_x': 2e-9,                                  # config 5 (512 problems): achieved max 5e-11, median 4e-12
}
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LINE = re.compile(r"^\s*(\d+):\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)(?:\s+(\S+))?\s*$")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b))))


def table_of(text):
    rows = []
    for ln in text.splitlines():
        mm = LINE.match(ln)
        if mm:
            rows.append([float(v) if v is not None else np.nan for v in mm.groups()[1:]])
    return np.array(rows)


def objerr(sol, g):
    return (abs(sol['primal objective'] - float(g['pobj'])) / max(1.0, abs(float(g['pobj']))),
            abs(sol['dual objective'] - float(g['dobj'])) / max(1.0, abs(float(g['dobj']))))


def table_err(got, ref):
    """largest relative deviation of pcost / dcost from the reference's printed values (printed with 5 significant digits)"""
    if got.shape[0] != ref.shape[0]:
        return float('inf')
    return float(np.max(np.abs(got[:, :2] - ref[:, :2]) / np.maximum(1e-300, np.abs(ref[:, :2]))))


def check_table(got, ref):
    """per-iteration trajectory against the reference's printed one (its precision: 5 digits / 1 digit)"""
    assert got.shape[0] == ref.shape[0], (got.shape, ref.shape)
    for k in range(ref.shape[0]):
        for c in (0, 1):                                              # pcost, dcost: % 8.4e
            assert abs(got[k, c] - ref[k, c]) <= 1.01e-4 * abs(ref[k, c]) + 1e-12, (k, c, got[k, c], ref[k, c])
        for c in (2, 3, 4):                                           # gap, pres, dres: % 4.0e / % 7.0e
            if ref[k, c] > 1e-13:
                assert 0.45 <= got[k, c] / ref[k, c] <= 2.2, (k, c, got[k, c], ref[k, c])


def run_with_progress(fn):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sol = fn()
    return sol, table_of(buf.getvalue())


def test_config2_device_loop_vs_reference_fixture():
    """BASELINE configs[1] (n=8192, m=16384): the device-resident coneqp loop, per-iteration values from its progress callback."""
    g = gold("full_qp8192")
    pr = synth.dense_qp(int(g['n']), int(g['m']), seed=int(g['seed']))
    sol, tab = run_with_progress(lambda: cvxopt_amd.coneqp_device(pr['P'], pr['q'], pr['G'], pr['h'], kktsolver='chol2',
                                                                  show_progress=True))
    assert sol['status'] == 'optimal' and int(g['status_optimal']) == 1
    assert sol['iterations'] == int(g['iterations'])
    ep, ed = objerr(sol, g)
    ex, ez, es = relerr(sol['x'], g['x']), relerr(sol['z'], g['z']), relerr(sol['s'], g['s'])
    record("config2_device_loop", iterations=sol['iterations'], pobj_relerr=ep, dobj_relerr=ed, x_relerr=ex, z_relerr=ez,
           s_relerr=es, table_cost_relerr=table_err(tab, g['table']))
    assert ep <= BOUND['obj'] and ed <= BOUND['obj']
    assert ex < BOUND['c2_x']
    assert ez < BOUND['c2_z'] and es < BOUND['c2_s']
    check_table(tab, g['table'])


def test_config2_lp_cone_fast_loop_vs_reference_fixture():
    """the LP-cone loop (`mi355kkt_coneqp_lp`, what bench.py's ipm_end_to_end runs) on the same problem"""
    g = gold("full_qp8192")
    pr = synth.dense_qp(int(g['n']), int(g['m']), seed=int(g['seed']))
    sol = cvxopt_amd.coneqp_lp(pr['P'], pr['q'], pr['G'], pr['h'])
    assert sol['status'] == 'optimal' and sol['iterations'] == int(g['iterations'])
    ep, ed = objerr(sol, g)
    ex = relerr(sol['x'], g['x'])
    record("config2_lp_cone_loop", iterations=sol['iterations'], pobj_relerr=ep, dobj_relerr=ed, x_relerr=ex)
    assert ep <= BOUND['obj'] and ed <= BOUND['obj']
    assert ex < BOUND['c2_x']


@pytest.mark.parametrize("solves", ["two-sweep", "one-sweep"])
def test_config2_hook_level_vs_reference_fixture(ref_cvxopt, knobs, solves):
    """the reference's own coneqp driver with the GPU factory installed behind kktsolver='chol2': same trajectory, and
    the scaling W it hands to every factor() call matches the CPU run's at full precision.  The relative residual of EVERY solve
    against the 3 x 3 KKT system it stands for (computed here on the host from the solve's inputs and outputs) is recorded for the
    last three iterations -- next to the late-iteration drift of W it explains.  Both forms of the triangular
    solves run: two pipelined sweeps with a global refinement step (the default, blas2.hip trsv_pair_kernel) and the round-3
    one-sweep kernel with per-block refinement (test knob MI355KKT_TRSV_PAIR=0)."""
    from cvxopt import matrix, solvers, misc, blas
    if solves == "one-sweep":
        knobs.setenv("MI355KKT_TRSV_PAIR", "0")
    g = gold("full_qp8192")
    n, m = int(g['n']), int(g['m'])
    pr = synth.dense_qp(n, m, seed=int(g['seed']))
    digests, residuals = [], []
    Pn, Gn = pr['P'], pr['G']
    kkt.install(misc)
    orig = misc.kkt_chol2
    try:
        def wrapped(*a, **k):
            fac = orig(*a, **k)

            def factor(W, *rest):
                digests.append(float(blas.nrm2(W['di'])))
                solve = fac(W, *rest)
                d = np.array(W['d']).ravel()
                it = len(digests) - 1

                def checked(x, y, z):
                    bx, bz = np.array(x).ravel().copy(), np.array(z).ravel().copy()
                    solve(x, y, z)
                    ux, wz = np.array(x).ravel(), np.array(z).ravel()      # z := W uz  (misc.py:1563)
                    uz = wz / d
                    r1 = Pn @ ux + Gn.T @ uz - bx                          # P ux + G' uz = bx
                    r3 = Gn @ ux - d * wz - bz                             # G ux - W'W uz = bz
                    residuals.append((it, float(np.sqrt(r1 @ r1 + r3 @ r3) / np.sqrt(bx @ bx + bz @ bz))))
                return checked
            return factor
        misc.kkt_chol2 = wrapped
        old = solvers.options.get('show_progress')
        solvers.options['show_progress'] = True
        try:
            sol, tab = run_with_progress(lambda: solvers.coneqp(matrix(pr['P']), matrix(pr['q']), matrix(pr['G']),
                                                                matrix(pr['h']), kktsolver='chol2'))
        finally:
            solvers.options['show_progress'] = old
    finally:
        kkt.uninstall()
    assert sol['status'] == 'optimal' and sol['iterations'] == int(g['iterations'])
    ep, ed = objerr(sol, g)
    ex = relerr(np.array(sol['x']).ravel(), g['x'])
    ref_d = g['w_digest'][:, 0]
    assert len(digests) == len(ref_d)
    derr = [abs(a - b) / b for a, b in zip(digests, ref_d)]
    last3 = [r for (it, r) in residuals if it >= len(digests) - 3]
    record("config2_hook_level" + ("" if solves == "two-sweep" else "_one_sweep"), iterations=sol['iterations'], pobj_relerr=ep,
           dobj_relerr=ed, x_relerr=ex, w_digest_relerr_per_factor_call=derr, table_cost_relerr=table_err(tab, g['table']),
           kkt_residual_per_solve_last3_iterations=last3, kkt_residual_per_solve_max=max(r for _, r in residuals))
    assert ep <= BOUND['obj'] and ed <= BOUND['obj']
    assert ex < BOUND['c2_x']
    check_table(tab, g['table'])
    # (rounding-level differences of the KKT solves are amplified along the central path: ||di|| grows to 1e4 .. 1e5 and the
    #  last iterations' systems have condition numbers around 1e10)
    for k, e in enumerate(derr):
        assert e <= (BOUND['w_early'] if k < 10 else BOUND['w_late']), (k, e)


def test_config3_device_loop_vs_reference_fixture():
    """BASELINE configs[2] (SOCP, n=2048, 1024 cones of dimension 8): device-resident conelp against the CPU reference run"""
    g = gold("full_socp2048")
    pr = synth.socp(n=int(g['n']), ncones=int(g['N']), r=int(g['r']), seed=int(g['seed']))
    sol, tab = run_with_progress(lambda: cvxopt_amd.conelp_device(pr['c'], pr['G'], pr['h'], pr['dims'], show_progress=True))
    assert sol['status'] == 'optimal' and sol['iterations'] == int(g['iterations'])
    ep, ed = objerr(sol, g)
    ex = relerr(sol['x'], g['x'])
    record("config3_socp_device_loop", iterations=sol['iterations'], pobj_relerr=ep, dobj_relerr=ed, x_relerr=ex,
           table_cost_relerr=table_err(tab[:, :5], g['table'][:, :5]))
    assert ep <= BOUND['c3_obj'] and ed <= BOUND['c3_obj']
    assert ex < BOUND['c3_x']
    check_table(tab[:, :5], g['table'][:, :5])


def test_config4_sparse_device_loop_vs_reference_fixture():
    """BASELINE configs[3] class (46^3 Laplacian box-QP, n = 97 336): supernodal engine + device-resident loop against the
    reference's sparse kkt_chol2 branch (CHOLMOD replaced by the SuperLU shim: solutions are ordering independent)"""
    import scipy.sparse as sp
    path = os.path.join(GOLD, "full_sparse46.npz")
    if not os.path.exists(path):
        pytest.skip("full_sparse46 fixture not generated")
    g = np.load(path, allow_pickle=False)
    k = int(g['k'])
    n = k ** 3
    P = synth.grid_laplacian(k)
    q = np.random.default_rng(int(g['seed'])).standard_normal(n)
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()

    class Sp(object):
        def __init__(self, A):
            A = sp.csc_matrix(A)
            A.sort_indices()
            self.size = A.shape
            self.CCS = (A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64))
    sol = cvxopt_amd.coneqp_lp(Sp(sp.tril(P)), q, Sp(G), np.ones(2 * n))
    assert sol['status'] == 'optimal' and sol['iterations'] == int(g['iterations'])
    ep, ed = objerr(sol, g)
    ex, ez = relerr(sol['x'], g['x']), relerr(sol['z'][::97], g['z_sample'])
    record("config4_sparse_device_loop", iterations=sol['iterations'], pobj_relerr=ep, dobj_relerr=ed, x_relerr=ex, z_sample_relerr=ez)
    assert ep <= BOUND['c4_obj'] and ed <= BOUND['c4_obj']
    assert ex < BOUND['c4_x']
    assert ez < BOUND['c4_z']


def test_config4_elasticity_stand_in_vs_reference_fixture():
    """The irregular stand-in of BASELINE configs[3] (3 degrees of freedom per node of a 3000-node tetrahedral mesh, n = 9000, ~48
    entries per row): the supernodal engine + device-resident loop against the reference's sparse kkt_chol2 branch run on the
    same problem in the build container (tests/golden/make_golden_full.py elasticity; CHOLMOD replaced by the SuperLU shim)."""
    import scipy.sparse as sp
    g = gold("full_elasticity9000")
    P = synth.tet_mesh_elasticity(int(g['nodes']), seed=int(g['seed']))
    n = P.shape[0]
    assert n == int(g['n'])
    q = np.random.default_rng(int(g['seed'])).standard_normal(n)
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()

    class Sp(object):
        def __init__(self, A):
            A = sp.csc_matrix(A)
            A.sort_indices()
            self.size = A.shape
            self.CCS = (A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64))
    sol = cvxopt_amd.coneqp_lp(Sp(sp.tril(P)), q, Sp(G), np.ones(2 * n))
    assert sol['status'] == 'optimal' and int(g['status_optimal']) == 1
    assert sol['iterations'] == int(g['iterations'])
    ep, ed = objerr(sol, g)
    ex, ez = relerr(sol['x'], g['x']), relerr(sol['z'], g['z'])
    record("config4_elasticity_device_loop", iterations=sol['iterations'], pobj_relerr=ep, dobj_relerr=ed, x_relerr=ex, z_relerr=ez)
    assert ep <= BOUND['c4_obj'] and ed <= BOUND['c4_obj']
    assert ex < 1e-9 and ez < 1e-8, (ex, ez)       # (x: SURVEY 8(d) asks 1e-7; z of nearly inactive constraints is ~1e-9 of the scale)


def test_config5_batch_sample_vs_reference_fixture():
    """BASELINE configs[4]: the first 64 problems of the batch (n=512, m=1024, seed = index) in the batched device loop
    against 64 individual CPU reference solves"""
    from cvxopt_amd.batch import BatchKkt, pack_problems
    g = gold("full_batch64")
    B, n, m = int(g['B']), int(g['n']), int(g['m'])
    P, q, Gt, h = pack_problems([synth.dense_qp(n, m, seed=i) for i in range(B)])
    kk = BatchKkt(Gt, P)
    try:
        res = kk.coneqp(q, h)
    finally:
        kk.close()
    assert np.all(res['status'] == 'optimal')
    assert np.array_equal(res['iterations'], g['iterations'])
    ep = float(np.max(np.abs(res['primal objective'] - g['pobj']) / np.maximum(1.0, np.abs(g['pobj']))))
    ed = float(np.max(np.abs(res['dual objective'] - g['dobj']) / np.maximum(1.0, np.abs(g['dobj']))))
    ex = [relerr(res['x'][b], g['x'][b]) for b in range(B)]
    record("config5_batch64", problems=B, pobj_relerr_max=ep, dobj_relerr_max=ed, x_relerr_max=max(ex), x_relerr_median=float(np.median(ex)))
    assert ep < BOUND['obj'] and ed < BOUND['obj']
    assert max(ex) < BOUND['c5_x']


def test_config5_batch512_vs_reference_fixture():
    """BASELINE configs[4], one GPU's share of the 4096 problems: 512 problems (n=512, m=1024, seed = index) in ONE batched device
    loop against 512 individual CPU reference solves: iteration counts, both objectives, ||x|| and every 8th entry of x"""
    from cvxopt_amd.batch import BatchKkt, pack_problems
    path = os.path.join(GOLD, "full_batch512.npz")
    if not os.path.exists(path):
        pytest.skip("full_batch512 fixture not generated")
    g = np.load(path, allow_pickle=False)
    B, n, m = int(g['B']), int(g['n']), int(g['m'])
    P, q, Gt, h = pack_problems([synth.dense_qp(n, m, seed=i) for i in range(B)])
    kk = BatchKkt(Gt, P)
    try:
        res = kk.coneqp(q, h)
    finally:
        kk.close()
    assert np.all(res['status'] == 'optimal')
    assert np.array_equal(res['iterations'], g['iterations'])
    ep = float(np.max(np.abs(res['primal objective'] - g['pobj']) / np.maximum(1.0, np.abs(g['pobj']))))
    ed = float(np.max(np.abs(res['dual objective'] - g['dobj']) / np.maximum(1.0, np.abs(g['dobj']))))
    ex = [relerr(res['x'][b][::8], g['x_sample'][b]) for b in range(B)]
    en = float(np.max(np.abs(np.linalg.norm(res['x'], axis=1) - g['x_norm']) / g['x_norm']))
    record("config5_batch512", problems=B, iterations_min=int(res['iterations'].min()), iterations_max=int(res['iterations'].max()),
           pobj_relerr_max=ep, dobj_relerr_max=ed, x_sample_relerr_max=max(ex), x_sample_relerr_median=float(np.median(ex)),
           x_norm_relerr_max=en)
    assert ep < BOUND['obj'] and ed < BOUND['obj']
    assert max(ex) < BOUND['c5_x'] and en < BOUND['c5_x']

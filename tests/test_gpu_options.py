"""Solver OPTIONS are inputs (VERDICT r3 "missing" 1): the device-resident loops behind cvxopt_amd.solvers.coneqp / conelp with
options['use_correction'], ['refinement'], ['maxiters'], the tolerances and initvals = {} against (i) tests/golden/options.npz
(the real reference, tests/golden/make_golden_options.py) and (ii) the live reference on the box (oracle/_ref)."""
import os
import sys

import numpy as np
import pytest

from helpers import load_golden, relerr

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gen(ref_cvxopt):
    import make_golden_options as g          # (importing it loads the reference; the fixtures themselves are read from the .npz)
    return g


def _solve(gen, kind, opts, initvals, batch=False):
    from cvxopt import matrix
    import cvxopt_amd.solvers as gs
    P, q, G, h, dims, A, b = gen.qp_problem(kind)
    o = dict(opts)
    o['show_progress'] = False
    kw = {}
    if A is not None:
        kw = {'A': matrix(A), 'b': matrix(b)}
    return gs.coneqp(matrix(P), matrix(q), matrix(G), matrix(h), dims, initvals=initvals, options=o, **kw)


def test_coneqp_options_match_the_reference_fixtures(gen):
    g = load_golden("options")
    for name, kind, opts, initvals in gen.CASES:
        sol = _solve(gen, kind, opts, initvals)
        assert sol['status'] == str(g[name + '_status']), name
        assert sol['iterations'] == int(g[name + '_iterations']), (name, sol['iterations'], int(g[name + '_iterations']))
        ref_p = float(g[name + '_pobj'])
        assert abs(sol['primal objective'] - ref_p) <= 1e-9 * max(1.0, abs(ref_p)), name
        assert abs(sol['dual objective'] - float(g[name + '_dobj'])) <= 1e-9 * max(1.0, abs(float(g[name + '_dobj']))), name
        # an unconverged iterate (maxiters = 3) is an iterate of the same trajectory: compare it just as tightly
        assert relerr(np.array(sol['x']).ravel(), g[name + '_x']) < 1e-7, name
        assert relerr(np.array(sol['s']).ravel(), g[name + '_s']) < 1e-6, name
        assert relerr(np.array(sol['z']).ravel(), g[name + '_z']) < 1e-6, name
        if g[name + '_y'].size:
            assert relerr(np.array(sol['y']).ravel(), g[name + '_y']) < 1e-6, name


def test_use_correction_changes_the_trajectory_like_the_reference(gen):
    """the option is not ignored: without the Mehrotra correction the reference needs more iterations -- and so do we"""
    g = load_golden("options")
    a = _solve(gen, 'lp_cone', {}, None)
    b = _solve(gen, 'lp_cone', {'use_correction': False}, None)
    assert a['iterations'] == int(g['default_iterations']) and b['iterations'] == int(g['nocorr_iterations'])
    assert b['iterations'] > a['iterations']


def test_conelp_options_match_the_reference_fixtures(gen):
    from cvxopt import matrix
    from cvxopt_amd import synth
    import cvxopt_amd.solvers as gs
    g = load_golden("options")
    pr = synth.socp(20, 5, 4, seed=21, ml=8)
    for name, opts in (('lp_default', {}), ('lp_refine2', {'refinement': 2}), ('lp_refine0', {'refinement': 0}),
                       ('lp_maxit4', {'maxiters': 4})):
        o = dict(opts)
        o['show_progress'] = False
        sol = gs.conelp(matrix(pr['c']), matrix(pr['G']), matrix(pr['h']), pr['dims'], options=o)
        assert sol['status'] == str(g[name + '_status']) and sol['iterations'] == int(g[name + '_iterations']), name
        assert abs(sol['primal objective'] - float(g[name + '_pobj'])) <= 1e-9 * max(1.0, abs(float(g[name + '_pobj']))), name
        assert relerr(np.array(sol['x']).ravel(), g[name + '_x']) < 1e-7, name


def test_batched_loop_honours_use_correction(gen):
    """mi355kkt_batch_set_option: every problem of a batch follows the reference's trajectory without the correction"""
    from cvxopt import matrix, solvers
    from cvxopt_amd import batch, synth
    probs = [synth.dense_qp(24, 50, seed=70 + k) for k in range(5)]
    P, q, Gt, h = batch.pack_problems(probs)
    for corr in (True, False):
        got = batch.coneqp_batch(P, q, Gt, h, resident=True, use_correction=corr)
        twin = batch.coneqp_batch(P, q, Gt, h, resident=False, use_correction=corr)
        for k, pr in enumerate(probs):
            ref = solvers.coneqp(matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']),
                                 options={'show_progress': False, 'use_correction': corr})
            assert got['status'][k] == ref['status'] and int(got['iterations'][k]) == ref['iterations'], (corr, k)
            assert int(twin['iterations'][k]) == ref['iterations'], (corr, k)
            assert relerr(got['x'][k], np.array(ref['x']).ravel()) < 1e-7


def test_ldl_refinement_is_a_handle_option_not_an_environment_switch(monkeypatch):
    """the refinement steps of the ldl flavours' solve() are per handle (mi355kkt_set_option)"""
    from cvxopt_amd import kkt, synth
    from oracle import kkt_oracle as ko
    n, m = 60, 150
    pr = synth.dense_qp(n, m, seed=3)
    W = synth.random_scaling(pr['dims'], seed=1, spread=3.0)
    rng = np.random.default_rng(0)
    bx, bz = rng.standard_normal(n), rng.standard_normal(m)
    res = {}
    monkeypatch.setenv("MI355KKT_LDL_REFINE", "0")               # (round 3's process-wide switch: must be inert now)
    for steps in (0, 2):
        f = kkt.kkt_ldl(pr['G'], pr['dims'], np.zeros((0, n)))
        f.engine.set_option("ldl_refinement", steps)
        x, y, z = bx.copy(), np.zeros(0), bz.copy()
        f(W, pr['P'])(x, y, z)
        res[steps] = ko.kkt_residual(pr['P'], np.zeros((0, n)), pr['G'], W, pr['dims'], bx, np.zeros(0), bz, x, y, z)
        f.engine.close()
    f = kkt.kkt_ldl(pr['G'], pr['dims'], np.zeros((0, n)))      # default: two steps
    x, y, z = bx.copy(), np.zeros(0), bz.copy()
    f(W, pr['P'])(x, y, z)
    default = ko.kkt_residual(pr['P'], np.zeros((0, n)), pr['G'], W, pr['dims'], bx, np.zeros(0), bz, x, y, z)
    with pytest.raises(ValueError):
        f.engine.set_option("no_such_option", 1)
    f.engine.close()
    assert res[2] <= res[0] and default <= 10 * res[2] + 1e-15 and res[2] < 1e-12

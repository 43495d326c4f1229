"""Round 5: the two-sweep triangular solves with a dedicated poller wave (csrc/blas2.hip trsv_pair_kernel<.., POLLER>: the working
waves' polls sat behind the loads of their next strip) behind the hook's solve() -- against the oracle (pinned to the reference's
kkt_chol2 / kkt_chol, tests/test_oracle.py), bit for bit against the round-4 form of the same kernel (test knob
MI355KKT_TRSV_AHEAD=0; the arithmetic and its order are unchanged) and under repetition; with equality constraints the factor of S
keeps its 128 x 128 inverses while the small Schur complement K is factored."""
import numpy as np
import pytest

from cvxopt_amd import kkt, synth
from helpers import record, relerr
from oracle import kkt_oracle as ko

pytestmark = pytest.mark.gpu


def _solve_all(f, W, P, rhs, p=0):
    out = []
    s = f(W, P)
    for bx, by, bz in rhs:
        x, y, z = bx.copy(), by.copy(), bz.copy()
        s(x, y, z)
        out.append((x, y, z))
    return out


@pytest.mark.parametrize("n,p", [(512, 0), (768, 0), (1024, 0), (1280, 3), (2048, 0), (2048, 40), (4096, 0)])
def test_poller_wave_changes_nothing_but_time(n, p, knobs):
    """n = 512 / 768 are below the tile Cholesky's threshold (no 128 x 128 inverses: one-sweep kernel), 1280 is above; p > 0: the
    reduced solve runs both triangular solves twice and must keep the two-sweep kernel (S's inverses survive the factorisation of K)"""
    m = 2 * n
    pr = synth.dense_qp(n, m, seed=n, p=p)
    G, P, dims = pr['G'], pr['P'], pr['dims']
    A = pr.get('A', np.zeros((0, n)))
    rng = np.random.default_rng(n + 1)
    rhs = [(rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)) for _ in range(3)]
    W = synth.random_scaling(dims, seed=5, spread=1.5)
    f = kkt.kkt_chol2(G, dims, A)
    try:
        new = _solve_all(f, W, P, rhs)
        again = _solve_all(f, W, P, rhs)
        knobs.setenv("MI355KKT_TRSV_AHEAD", "0")
        old = _solve_all(f, W, P, rhs)
    finally:
        f.engine.close()
    for a, b, c in zip(new, again, old):
        for u, v, w in zip(a, b, c):
            assert np.array_equal(u, v)            # deterministic: same factor, same right-hand sides
            assert np.array_equal(u, w)            # the poller changes who polls, not what is computed
    oracle = ko.KktChol2(G, dims, A).factor(W, P)
    worst, worst_res = 0.0, 0.0
    for (bx, by, bz), (x, y, z) in zip(rhs, new):
        xo, yo, zo = bx.copy(), by.copy(), bz.copy()
        oracle(xo, yo, zo)
        worst = max(worst, relerr(x, xo), relerr(z, zo), relerr(y, yo) if p else 0.0)
        worst_res = max(worst_res, ko.kkt_residual(P, A, G, W, dims, bx, by, bz, x, y, z))
    record("round5_trsv_poller_%d_%d" % (n, p), x_vs_oracle=worst, kkt_residual=worst_res)
    assert worst < 1e-9 and worst_res < 1e-12, (worst, worst_res)


def test_solves_with_second_order_cones_and_equalities():
    """config-3 shape (second-order cones) plus a few equality constraints"""
    n, p = 1024, 9
    pr = synth.socp(n=n, ncones=256, r=8, seed=3, ml=64)
    dims = pr['dims']
    m = dims['l'] + sum(dims['q'])
    rng = np.random.default_rng(7)
    A = np.asfortranarray(rng.standard_normal((p, n)))
    W = synth.random_scaling(dims, seed=2, spread=1.0)
    f = kkt.kkt_chol(pr['G'], dims, A)
    bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    x, y, z = bx.copy(), by.copy(), bz.copy()
    try:
        f(W)(x, y, z)
    finally:
        f.engine.close()
    xo, yo, zo = bx.copy(), by.copy(), bz.copy()
    ko.KktChol(pr['G'], dims, A).factor(W, None)(xo, yo, zo)
    assert relerr(x, xo) < 1e-9 and relerr(y, yo) < 1e-9 and relerr(z, zo) < 1e-9

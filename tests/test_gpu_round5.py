"""Round 5: the wide triangular solves (csrc/trsv_wide.hip: 256-row hops, eight workgroups per block row, two sweeps, explicit
256 x 256 diagonal-block inverses formed by pair_inverse_kernel) behind the hook's solve() -- against the oracle (pinned to the
reference's kkt_chol2 / kkt_chol, tests/test_oracle.py), against the round-4 kernels on the same factor (test knob
MI355KKT_TRSV_WIDE=0), through the KKT residual, and bit for bit under repetition."""
import numpy as np
import pytest

from cvxopt_amd import kkt, synth
from helpers import record, relerr
from oracle import kkt_oracle as ko

pytestmark = pytest.mark.gpu


def _solve_all(f, W, P, rhs):
    out = []
    s = f(W, P)
    for bx, bz in rhs:
        x, y, z = bx.copy(), np.zeros(0), bz.copy()
        s(x, y, z)
        out.append((x, z))
    return out


@pytest.mark.parametrize("n", [512, 768, 1024, 1280, 2048, 4096])
def test_wide_triangular_solves_match_the_oracle_and_the_round4_kernels(n, knobs):
    """n = 768 and 1280 are not multiples of 256 and below / above the tile kernel's threshold: they take the round-4 paths and
    must be unaffected; the others run trsv_wide_kernel (n >= 1024: the tile Cholesky leaves the 128 x 128 inverses)."""
    m = 2 * n
    pr = synth.dense_qp(n, m, seed=n)
    G, P, dims, A = pr['G'], pr['P'], pr['dims'], np.zeros((0, n))
    rng = np.random.default_rng(n + 1)
    rhs = [(rng.standard_normal(n), rng.standard_normal(m)) for _ in range(3)]
    W = synth.random_scaling(dims, seed=5, spread=1.5)
    f = kkt.kkt_chol2(G, dims, A)
    try:
        wide = _solve_all(f, W, P, rhs)
        again = _solve_all(f, W, P, rhs)
        knobs.setenv("MI355KKT_TRSV_WIDE", "0")
        old = _solve_all(f, W, P, rhs)
    finally:
        f.engine.close()
    for (x, z), (x2, z2) in zip(wide, again):
        assert np.array_equal(x, x2) and np.array_equal(z, z2)          # deterministic: same factor, same right-hand sides
    oracle = ko.KktChol2(G, dims, A).factor(W, P)
    worst_o, worst_r, worst_res = 0.0, 0.0, 0.0
    for (bx, bz), (x, z), (xr, zr) in zip(rhs, wide, old):
        xo, yo, zo = bx.copy(), np.zeros(0), bz.copy()
        oracle(xo, yo, zo)
        worst_o = max(worst_o, relerr(x, xo), relerr(z, zo))
        worst_r = max(worst_r, relerr(x, xr), relerr(z, zr))
        res = ko.kkt_residual(P, A, G, W, dims, bx, np.zeros(0), bz, x, np.zeros(0), z)
        res_old = ko.kkt_residual(P, A, G, W, dims, bx, np.zeros(0), bz, xr, np.zeros(0), zr)
        worst_res = max(worst_res, res)
        assert res <= max(1e-12, 4.0 * res_old), (res, res_old)         # as small a residual as the kernels it replaces
    record("round5_trsv_wide_%d" % n, x_vs_oracle=worst_o, x_vs_round4_kernels=worst_r, kkt_residual=worst_res)
    assert worst_o < 1e-9 and worst_r < 1e-10, (worst_o, worst_r)


def test_wide_solves_with_second_order_cones_and_equalities():
    """config-3 shape (n = 2048, second-order cones) plus a few equality constraints: the reduced solve runs both triangular
    solves twice (x and the Schur-complement correction)"""
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

"""Every 's'-block operation of the device loops (csrc/cone_ops_s.h) run ON THE DEVICE by the two teams the loops use -- the
1024-thread workgroup (LDS-resident lane-group Jacobi) and a single wave (blocks of order <= 16, one per wave) -- against the
host twin of the same source, which tests/test_sdp_ops_cpu.py checks against the reference's misc functions."""
import numpy as np
import pytest

from cvxopt_amd import _capi

pytestmark = pytest.mark.gpu
p = lambda a: a.ctypes.data if a is not None else None
F = lambda a: np.asfortranarray(a, dtype=float)


def spd(rng, m):
    a = rng.standard_normal((m, m))
    return F(a @ a.T / m + 0.5 * np.eye(m))


def sym(rng, m):
    a = rng.standard_normal((m, m))
    return F(0.5 * (a + a.T))


def run(op, m, arg, x, y=None, r=None, rti=None, lam=None, team=None):
    L = _capi.lib()
    a = [None if v is None else v.copy(order='F') for v in (x, y, r, rti, lam)]
    if team is None:
        rc = L.mi355kkt_test_sdp_op_host(op, m, arg, *[p(v) for v in a])
    else:
        rc = L.mi355kkt_test_sdp_op_device(op, m, arg, team, *[p(v) for v in a])
    return rc, a


@pytest.mark.parametrize("m,team", [(m, t) for m in (1, 2, 3, 4, 7, 12, 16) for t in (0, 1)] + [(m, 0) for m in (33, 100, 120, 150)])
def test_block_operations_on_the_device_match_the_host_twin(m, team):
    rng = np.random.default_rng(m)
    s, z, x, y = spd(rng, m), spd(rng, m), sym(rng, m), sym(rng, m)
    lam0 = rng.random(m) + 0.2
    _, (_, _, r, rti, lam) = run(6, m, 0, s, z, F(np.zeros((m, m))), F(np.zeros((m, m))), np.zeros(m))
    tol = 1e-13 * m
    for op, arg, args in ((0, 0, (x, None, r, rti, None)), (0, 1, (x, None, r, rti, None)), (0, 2, (x, None, r, rti, None)),
                          (0, 3, (x, None, r, rti, None)), (1, 0, (x, y, None, None, None)), (2, 0, (x, None, None, None, lam0)),
                          (2, 1, (x, None, None, None, lam0)), (3, 0, (x, None, None, None, lam0)),
                          (3, 1, (x, None, None, None, lam0)), (8, 0, (s, None, None, None, None))):
        rh, ah = run(op, m, arg, *args)
        rd, ad = run(op, m, arg, *args, team=team)
        assert rh == rd == 0
        assert np.abs(ah[0] - ad[0]).max() <= tol * max(1.0, np.abs(ah[0]).max()), (op, arg)
    # smallest eigenvalue, eigendecomposition (vectors up to signs: through the reconstruction)
    _, ah = run(4, m, 0, x, lam=np.zeros(m))
    _, ad = run(4, m, 0, x, lam=np.zeros(m), team=team)
    assert abs(ah[4][0] - ad[4][0]) <= 1e-13 * m * np.linalg.norm(x)
    _, ad = run(5, m, 0, x, lam=np.zeros(m), team=team)
    assert np.abs(np.linalg.eigvalsh(x) - ad[4]).max() <= 1e-13 * m * np.linalg.norm(x)
    assert np.abs(ad[0] @ np.diag(ad[4]) @ ad[0].T - x).max() <= 1e-12 * m * np.linalg.norm(x)
    assert np.abs(ad[0].T @ ad[0] - np.eye(m)).max() <= 1e-13 * m
    # compute_scaling / update_scaling through their defining identities and the host twin's invariants
    rd, (_, _, r2, rti2, lam2) = run(6, m, 0, s, z, F(np.zeros((m, m))), F(np.zeros((m, m))), np.zeros(m), team=team)
    assert rd == 0
    assert np.abs(r2.T @ z @ r2 - np.diag(lam2)).max() <= 1e-12 * m * lam2.max()
    assert np.abs(rti2.T @ r2 - np.eye(m)).max() <= 1e-11 * m
    assert np.allclose(lam2, lam, rtol=1e-11, atol=0)
    Ls, Lz = F(np.linalg.cholesky(spd(rng, m))), F(np.linalg.cholesky(spd(rng, m)))
    _, (_, _, r3, rti3, lam3) = run(7, m, 0, Ls, Lz, r, rti, lam)
    _, (_, _, r4, rti4, lam4) = run(7, m, 0, Ls, Lz, r, rti, lam, team=team)
    assert np.allclose(lam3, lam4, rtol=1e-10, atol=0)
    assert np.abs(r3 @ r3.T - r4 @ r4.T).max() <= 1e-10 * np.linalg.norm(r3) ** 2
    assert np.abs(rti4.T @ r4 - np.eye(m)).max() <= 1e-10 * m


def test_potrf_failure_is_reported_by_both_teams():
    rng = np.random.default_rng(1)
    a = spd(rng, 9)
    a[4, 4] = -1.0
    for team in (0, 1):
        rc, _ = run(8, 9, 0, a, team=team)
        assert rc == 5

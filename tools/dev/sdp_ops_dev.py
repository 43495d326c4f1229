"""Developer probe: every 's'-block operation on the device with the workgroup team and the wave team against the host twin."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from cvxopt_amd import _capi
L = _capi.lib()
p = lambda a: a.ctypes.data if a is not None else None
F = lambda a: np.asfortranarray(a, dtype=float)


def spd(rng, m):
    a = rng.standard_normal((m, m))
    return F(a @ a.T / m + 0.5 * np.eye(m))


def sym(rng, m):
    a = rng.standard_normal((m, m))
    return F(0.5 * (a + a.T))


def run(fn, op, m, arg, x, y=None, r=None, rti=None, lam=None, team=None):
    a = [None if v is None else v.copy(order='F') for v in (x, y, r, rti, lam)]
    if team is None:
        rc = L.mi355kkt_test_sdp_op_host(op, m, arg, *[p(v) for v in a])
    else:
        rc = L.mi355kkt_test_sdp_op_device(op, m, arg, team, *[p(v) for v in a])
    return rc, a


for m in (2, 3, 4, 7, 12, 16, 40):
    rng = np.random.default_rng(m)
    s, z, x, y = spd(rng, m), spd(rng, m), sym(rng, m), sym(rng, m)
    lam0 = rng.random(m) + 0.2
    _, (_, _, r, rti, lam) = run(None, 6, m, 0, s, z, F(np.zeros((m, m))), F(np.zeros((m, m))), np.zeros(m))
    for team in ((0, 1) if m <= 16 else (0,)):
        msgs = []
        for op, arg, args in ((0, 0, (x, None, r, rti, None)), (0, 3, (x, None, r, rti, None)), (1, 0, (x, y, None, None, None)),
                              (2, 1, (x, None, None, None, lam0)), (3, 0, (x, None, None, None, lam0)),
                              (4, 0, (x, None, None, None, np.zeros(m))), (5, 0, (x, None, None, None, np.zeros(m))),
                              (8, 0, (s, None, None, None, None))):
            rh, ah = run(None, op, m, arg, *args)
            rd, ad = run(None, op, m, arg, *args, team=team)
            if op == 5:      # eigenvectors up to sign: compare reconstruction and eigenvalues
                err = max(np.abs(ah[4] - ad[4]).max(), np.abs(ad[0] @ np.diag(ad[4]) @ ad[0].T - x).max())
            elif op == 4:
                err = abs(ah[4][0] - ad[4][0])
            else:
                err = np.abs(ah[0] - ad[0]).max()
            msgs.append("op%d:%.0e" % (op, err) + ("" if rh == rd else " rc %d/%d" % (rh, rd)))
        # compute_scaling / update_scaling through their defining identities
        rd, (_, _, r2, rti2, lam2) = run(None, 6, m, 0, s, z, F(np.zeros((m, m))), F(np.zeros((m, m))), np.zeros(m), team=team)
        e6 = max(np.abs(r2.T @ z @ r2 - np.diag(lam2)).max(), np.abs(rti2.T @ r2 - np.eye(m)).max(), np.abs(np.sort(lam2) - np.sort(lam)).max())
        Ls, Lz = F(np.linalg.cholesky(spd(rng, m))), F(np.linalg.cholesky(spd(rng, m)))
        _, (_, _, r3, rti3, lam3) = run(None, 7, m, 0, Ls, Lz, r, rti, lam)
        _, (_, _, r4, rti4, lam4) = run(None, 7, m, 0, Ls, Lz, r, rti, lam, team=team)
        e7 = max(np.abs(r3 @ r3.T - r4 @ r4.T).max(), np.abs(np.sort(lam3) - np.sort(lam4)).max())
        print("m %2d team %s: %s cs:%.0e us:%.0e" % (m, "wave" if team else "wg  ", " ".join(msgs), e6, e7), flush=True)

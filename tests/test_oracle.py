"""Pins the oracle (oracle/kkt_oracle.py): (i) against the committed golden vectors produced by the real
reference (tests/golden/make_golden.py), always; (ii) against the live reference in oracle/_ref when present;
(iii) against known answers the reference documents / tests hold for this path."""
import numpy as np
import pytest

from oracle import kkt_oracle as ko
from cvxopt_amd import synth
from helpers import load_golden, dims_of, w_of, relerr


@pytest.mark.parametrize("case", ["scale0", "scale1", "scale2", "scale3"])
def test_scale_matches_reference_golden(case):
    rec = load_golden(case)
    dims = dims_of(rec)
    W = w_of(rec, dims)
    for tr in "NT":
        for inv in "NI":
            x = rec['x'].copy(order='F')
            ko.scale(x, W, dims, trans=tr, inverse=inv)
            ref = rec['out_%s%s' % (tr, inv)]
            # compare what the reference defines: l, q parts fully; 's' parts on the lower triangle
            mask = np.ones(x.shape[0], dtype=bool)
            ind = dims['l'] + sum(dims['q'])
            for nk in dims['s']:
                M = np.tril(np.ones((nk, nk), dtype=bool)).ravel(order='F')
                mask[ind:ind + nk * nk] = M
                ind += nk * nk
            assert relerr(x[mask], ref[mask]) < 1e-13, (case, tr, inv)


def test_scale_inverse_round_trip():
    dims = {'l': 7, 'q': [5, 9], 's': [4]}
    W = synth.random_scaling(dims, seed=3, spread=1.0)
    rng = np.random.default_rng(0)
    x = rng.standard_normal(ko.cdim(dims))
    X = x[7 + 14:].reshape(4, 4, order='F')
    X[:] = X + X.T
    for tr in "NT":
        y = ko.scale(x.copy(), W, dims, trans=tr, inverse='N')
        y = ko._symmetrize_s(y, dims)
        back = ko.scale(y, W, dims, trans=tr, inverse='I')
        back = ko._symmetrize_s(back, dims)
        assert relerr(back, x) < 1e-10


def test_pack_unpack_round_trip_and_inner_product():
    dims = {'l': 2, 'q': [3], 's': [3, 2]}
    rng = np.random.default_rng(1)
    def symvec():
        u = rng.standard_normal(ko.cdim(dims))
        ind = 5
        for nk in dims['s']:
            X = u[ind:ind + nk * nk].reshape(nk, nk, order='F'); X[:] = X + X.T; ind += nk * nk
        return u
    u, v = symvec(), symvec()
    pu, pv = ko.pack(u, dims), ko.pack(v, dims)
    assert pu.shape[0] == ko.cdim_packed(dims)
    assert abs(pu @ pv - u @ v) < 1e-12 * abs(u @ v)        # the sqrt(2) scaling preserves inner products
    back = ko.unpack(pu, dims)
    assert relerr(np.tril(back[5:14].reshape(3, 3, order='F')), np.tril(u[5:14].reshape(3, 3, order='F'))) < 1e-15


KIND_ORACLE = {'chol2': ko.KktChol2, 'chol': ko.KktChol, 'ldl': ko.KktLdl}


@pytest.mark.parametrize("case", ["kkt_lp_p0", "kkt_lp_p5", "kkt_soc", "kkt_soc_many", "kkt_sdp"])
def test_kktsolvers_match_reference_golden(case):
    rec = load_golden(case)
    dims = dims_of(rec)
    W = w_of(rec, dims)
    G, A, H = rec['G'], rec['A'], rec['H']
    for kind, cls in KIND_ORACLE.items():
        if ('x_' + kind) not in rec:
            continue
        x, y, z = rec['bx'].copy(), rec['by'].copy(), rec['bz'].copy()
        cls(G, dims, A).factor(W, H)(x, y, z)
        assert relerr(x, rec['x_' + kind]) < 1e-10, (case, kind)
        assert relerr(y, rec['y_' + kind]) < 1e-10, (case, kind)
        zr = rec['z_' + kind]
        if dims['s']:                                  # reference leaves strict-upper 's' entries unspecified
            z, zr = ko.pack(z, dims), ko.pack(ko._symmetrize_s(zr, dims), dims)
            z = ko.pack(ko._symmetrize_s(ko.unpack(z, dims), dims), dims)
        assert relerr(z, zr) < 1e-10, (case, kind)
    x, y, z = rec['bx'].copy(), rec['by'].copy(), rec['bz'].copy()
    ko.KktLdl(G, dims, A, kktreg=1e-3).factor(W, H)(x, y, z)
    assert relerr(x, rec['x_ldlreg']) < 1e-10 and relerr(y, rec['y_ldlreg']) < 1e-10


def test_all_reference_kktsolvers_agree_in_golden():
    """The reference's own four factorisations give the same answer (SURVEY.md 8(c) probe)."""
    rec = load_golden("kkt_lp_p5")
    for kind in ("chol", "ldl", "ldl2"):
        assert relerr(rec['x_' + kind], rec['x_chol2']) < 1e-10
        assert relerr(rec['z_' + kind], rec['z_chol2']) < 1e-10


def test_oracle_residual_is_small_on_golden_solution():
    rec = load_golden("kkt_soc")
    dims = dims_of(rec)
    W = w_of(rec, dims)
    res = ko.kkt_residual(rec['H'], rec['A'], rec['G'], W, dims, rec['bx'], rec['by'], rec['bz'],
                          rec['x_ldl'], rec['y_ldl'], rec['z_ldl'])
    assert res < 1e-12


def test_chol2_singular_fallback_and_errors():
    n, m, p = 30, 20, 12
    rng = np.random.default_rng(3)
    G, A = rng.standard_normal((m, n)), rng.standard_normal((p, n))
    dims = {'l': m, 'q': [], 's': []}
    W = synth.random_scaling(dims, seed=1, spread=0.5)
    o = ko.KktChol2(G, dims, A)
    bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
    x, y, z = bx.copy(), by.copy(), bz.copy()
    o.factor(W, None)(x, y, z)
    assert o.singular
    assert ko.kkt_residual(None, A, G, W, dims, bx, by, bz, x, y, z) < 1e-10
    with pytest.raises(ArithmeticError):
        ko.KktChol2(G, dims, np.zeros((0, n))).factor(W, None)
    with pytest.raises(ValueError):
        ko.KktChol2(np.zeros((5, 3)), {'l': 2, 'q': [3], 's': []}, np.zeros((0, 3)))


# ---- live reference (build container, or the prebuilt oracle/_ref on the GPU box) ------------------------
def test_oracle_vs_live_reference_kktsolvers(ref_cvxopt):
    from cvxopt import matrix, spmatrix, misc
    rng = np.random.default_rng(12)
    for dims, n, p in [({'l': 50, 'q': [], 's': []}, 30, 0), ({'l': 10, 'q': [6, 6], 's': []}, 14, 4)]:
        m = ko.cdim(dims)
        G, A = np.asfortranarray(rng.standard_normal((m, n))), np.asfortranarray(rng.standard_normal((p, n)))
        B = rng.standard_normal((n, n)); H = np.asfortranarray(B @ B.T + np.eye(n))
        Wn = synth.random_scaling(dims, seed=n, spread=1.5)
        W = {'d': matrix(Wn['d']), 'di': matrix(Wn['di']), 'v': [matrix(v) for v in Wn['v']], 'beta': list(Wn['beta']),
             'r': [], 'rti': []}
        bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), rng.standard_normal(m)
        Ac = matrix(A) if p else spmatrix([], [], [], (0, n))
        for kind, cls in (('chol', ko.KktChol), ('ldl', ko.KktLdl)) + ((('chol2', ko.KktChol2),) if not dims['q'] else ()):
            xr, yr, zr = matrix(bx), (matrix(by) if p else matrix(0.0, (0, 1))), matrix(bz)
            getattr(misc, 'kkt_' + kind)(matrix(G), dims, Ac)(W, matrix(H))(xr, yr, zr)
            x, y, z = bx.copy(), by.copy(), bz.copy()
            cls(G, dims, A).factor(Wn, H)(x, y, z)
            assert relerr(x, np.array(xr).ravel()) < 1e-10 and relerr(z, np.array(zr).ravel()) < 1e-10
        assert ko.w_from_cvxopt(W)['beta'] == list(Wn['beta'])


def test_oracle_as_kktsolver_reproduces_golden_coneqp(ref_cvxopt):
    """Plugging the oracle into the reference driver reproduces the reference's own iteration count and
    objectives (this is the parity bar the HIP path is held to on the GPU)."""
    from cvxopt import matrix, solvers, spmatrix
    for name in ("coneqp_qp64", "coneqp_qp96_p8"):
        g = load_golden(name)
        n, m, p = int(g['n']), int(g['m']), int(g['p'])
        pr = synth.dense_qp(n, m, seed=int(g['seed']), p=p)
        A = pr.get('A', np.zeros((0, n)))
        orc = ko.KktChol2(pr['G'], pr['dims'], A)

        def kktsolver(W):
            f = orc.factor(ko.w_from_cvxopt(W), pr['P'])

            def solve(x, y, z):
                xv, yv, zv = (np.array(u).ravel() for u in (x, y, z))
                f(xv, yv, zv)
                x[:], z[:] = matrix(xv), matrix(zv)
                if p:
                    y[:] = matrix(yv)
            return solve
        kw = dict(A=matrix(pr['A']), b=matrix(pr['b'])) if p else {}
        sol = solvers.coneqp(matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']), kktsolver=kktsolver, **kw)
        assert sol['status'] == 'optimal' and int(g['status_optimal']) == 1
        assert sol['iterations'] == int(g['iterations'])
        assert abs(sol['primal objective'] - float(g['pobj'])) <= 1e-9 * max(1, abs(float(g['pobj'])))
        assert relerr(np.array(sol['x']).ravel(), g['x']) < 1e-7


def test_golden_qp256_is_the_survey_probe_value():
    g = load_golden("coneqp_qp256")
    assert int(g['iterations']) == 10
    assert abs(float(g['pobj']) - 5.032917338763e+01) < 1e-9
    assert abs(float(g['dobj']) - 5.032914076815e+01) < 1e-9


def test_reference_documented_known_answers(ref_cvxopt):
    """Known answers the reference's tests hold: tests/test_examples.py:31-34 (lp x = [1, 1]) and :27-29
    (coneqp x ~ [0.72558319, 0.61806264, 0.30253528]); problem data restated from the doc transcripts
    doc/source/coneprog.rst (lp :640-670, coneqp :560-600)."""
    from cvxopt import matrix, solvers
    c = matrix([-4., -5.])
    G = matrix([[2., 1., -1., 0.], [1., 2., 0., -1.]])
    h = matrix([3., 3., 0., 0.])
    sol = solvers.lp(c, G, h)
    assert abs(sol['x'][0] - 1.0) < 1e-5 and abs(sol['x'][1] - 1.0) < 1e-5
    A = matrix([[.3, -.4, -.2, -.4, 1.3], [.6, 1.2, -1.7, .3, -.3], [-.3, .0, .6, -1.2, -2.0]])
    b = matrix([1.5, .0, -1.2, -.7, .0])
    m, n = A.size
    I = matrix(0.0, (n, n)); I[::n + 1] = 1.0
    G = matrix([-I, matrix(0.0, (1, n)), I])
    h = matrix(n * [0.0] + [1.0] + n * [0.0])
    dims = {'l': n, 'q': [n + 1], 's': []}
    x = solvers.coneqp(A.T * A, -A.T * b, G, h, dims)['x']
    assert relerr(np.array(x).ravel(), [0.72558319, 0.61806264, 0.30253528]) < 1e-5

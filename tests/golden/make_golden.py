"""Generates tests/golden/*.npz by running the REAL reference (cvxopt built from /root/reference by
oracle/build_ref.sh) in the build container.  Committed together with its outputs so that the GPU box
(where /root/reference does not exist) and any later reader can re-check the oracle and the HIP path
against reference-produced numbers.

    python tests/golden/make_golden.py        # rewrites the fixtures (deterministic, seeded)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import refloader            # noqa: E402
from cvxopt_amd import synth            # noqa: E402

cvx = refloader.load()
from cvxopt import matrix, spmatrix, solvers, misc, misc_solvers   # noqa: E402

solvers.options['show_progress'] = False


def w_to_cvx(W):
    return {'d': matrix(W['d']), 'di': matrix(W['di']), 'v': [matrix(v) for v in W['v']],
            'beta': list(W['beta']), 'r': [matrix(r) for r in W['r']], 'rti': [matrix(r) for r in W['rti']]}


def flat_w(W):
    out = {'W_d': W['d'], 'W_di': W['di'], 'W_beta': np.array(W['beta'])}
    out['W_v'] = np.concatenate(W['v']) if W['v'] else np.zeros(0)
    out['W_r'] = np.concatenate([r.ravel(order='F') for r in W['r']]) if W['r'] else np.zeros(0)
    out['W_rti'] = np.concatenate([r.ravel(order='F') for r in W['rti']]) if W['rti'] else np.zeros(0)
    return out


def cdim(dims):
    return dims['l'] + sum(dims['q']) + sum(k * k for k in dims['s'])


def sym_s_parts(u, dims):
    """make the 's' parts of a cone vector symmetric matrices (as the solvers hand them to the hook)"""
    ind = dims['l'] + sum(dims['q'])
    for nk in dims['s']:
        X = u[ind:ind + nk * nk].reshape(nk, nk, order='F')
        X[:] = 0.5 * (X + X.T)
        ind += nk * nk
    return u


def scale_cases():
    cases = {}
    for ci, dims in enumerate([{'l': 5, 'q': [], 's': []}, {'l': 3, 'q': [4, 2, 6], 's': []},
                               {'l': 2, 'q': [3], 's': [3, 2]}, {'l': 0, 'q': [8] * 4, 's': []}]):
        W = synth.random_scaling(dims, seed=10 + ci, spread=1.0)
        rng = np.random.default_rng(ci)
        x = np.asfortranarray(rng.standard_normal((cdim(dims), 3)))
        for c in range(3):
            sym_s_parts(x[:, c], dims)
        rec = {'dims_l': dims['l'], 'dims_q': np.array(dims['q'], dtype=int), 'dims_s': np.array(dims['s'], dtype=int),
               'x': x}
        rec.update(flat_w(W))
        for tr in 'NT':
            for inv in 'NI':
                y = matrix(x)
                misc_solvers.scale(y, w_to_cvx(W), trans=tr, inverse=inv)
                rec['out_%s%s' % (tr, inv)] = np.array(y)
        cases['scale%d' % ci] = rec
    return cases


def kkt_cases():
    cases = {}
    specs = [
        ('lp_p0', {'l': 40, 'q': [], 's': []}, 24, 0, ['chol2', 'chol', 'ldl', 'ldl2']),
        ('lp_p5', {'l': 30, 'q': [], 's': []}, 20, 5, ['chol2', 'chol', 'ldl', 'ldl2']),
        ('soc', {'l': 6, 'q': [5, 3, 8], 's': []}, 12, 3, ['chol', 'ldl', 'ldl2']),
        ('soc_many', {'l': 0, 'q': [4] * 10, 's': []}, 16, 0, ['chol', 'ldl']),
        ('sdp', {'l': 3, 'q': [4], 's': [3, 4]}, 10, 2, ['chol', 'ldl', 'ldl2']),
    ]
    for name, dims, n, p, kinds in specs:
        rng = np.random.default_rng(sum(name.encode()))
        m = cdim(dims)
        G = np.asfortranarray(rng.standard_normal((m, n)))
        # columns of G must be symmetric in the 's' parts
        for c in range(n):
            sym_s_parts(G[:, c], dims)
        A = np.asfortranarray(rng.standard_normal((p, n)))
        B = rng.standard_normal((n, n))
        H = np.asfortranarray(B @ B.T / n + 0.1 * np.eye(n))
        W = synth.random_scaling(dims, seed=len(name), spread=1.0)
        bx, by, bz = rng.standard_normal(n), rng.standard_normal(p), sym_s_parts(rng.standard_normal(m), dims)
        rec = {'dims_l': dims['l'], 'dims_q': np.array(dims['q'], dtype=int), 'dims_s': np.array(dims['s'], dtype=int),
               'G': G, 'A': A, 'H': H, 'bx': bx, 'by': by, 'bz': bz}
        rec.update(flat_w(W))
        for kind in kinds:
            fac = getattr(misc, 'kkt_' + kind)(matrix(G), dims, matrix(A) if p else spmatrix([], [], [], (0, n)))
            x, y, z = matrix(bx), matrix(by) if p else matrix(0.0, (0, 1)), matrix(bz)
            fac(w_to_cvx(W), matrix(H))(x, y, z)
            rec['x_' + kind], rec['y_' + kind], rec['z_' + kind] = (np.array(x).ravel(), np.array(y).ravel(),
                                                                     np.array(z).ravel())
        # kkt_ldl with kktreg
        fac = misc.kkt_ldl(matrix(G), dims, matrix(A) if p else spmatrix([], [], [], (0, n)), kktreg=1e-3)
        x, y, z = matrix(bx), matrix(by) if p else matrix(0.0, (0, 1)), matrix(bz)
        fac(w_to_cvx(W), matrix(H))(x, y, z)
        rec['x_ldlreg'], rec['y_ldlreg'], rec['z_ldlreg'] = np.array(x).ravel(), np.array(y).ravel(), np.array(z).ravel()
        cases['kkt_' + name] = rec
    return cases


def solver_cases():
    """End-to-end drivers: the numbers every kktsolver must reproduce (status, iterations, objectives, x)."""
    cases = {}
    for name, (n, m, p) in {'qp64': (64, 160, 0), 'qp96_p8': (96, 200, 8), 'qp256': (256, 512, 0)}.items():
        pr = synth.dense_qp(n, m, seed=0 if name == 'qp256' else 2, p=p)
        kw = dict(A=matrix(pr['A']), b=matrix(pr['b'])) if p else {}
        sol = solvers.coneqp(matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']), kktsolver='chol2', **kw)
        cases['coneqp_' + name] = {'n': n, 'm': m, 'p': p, 'seed': 0 if name == 'qp256' else 2,
                                   'iterations': sol['iterations'], 'pobj': sol['primal objective'],
                                   'dobj': sol['dual objective'], 'x': np.array(sol['x']).ravel(),
                                   'z': np.array(sol['z']).ravel(), 'status_optimal': int(sol['status'] == 'optimal')}
    for name, (n, N, r, ml) in {'socp_small': (20, 6, 4, 3), 'socp_mid': (64, 32, 8, 0)}.items():
        pr = synth.socp(n, N, r, seed=1, ml=ml)
        sol = solvers.conelp(matrix(pr['c']), matrix(pr['G']), matrix(pr['h']), pr['dims'], kktsolver='chol')
        cases['conelp_' + name] = {'n': n, 'N': N, 'r': r, 'ml': ml, 'seed': 1, 'iterations': sol['iterations'],
                                   'pobj': sol['primal objective'], 'dobj': sol['dual objective'],
                                   'x': np.array(sol['x']).ravel(), 'status_optimal': int(sol['status'] == 'optimal')}
    return cases


if __name__ == "__main__":
    allc = {}
    allc.update(scale_cases())
    allc.update(kkt_cases())
    allc.update(solver_cases())
    for name, rec in allc.items():
        np.savez_compressed(os.path.join(HERE, name + '.npz'), **rec)
        print("wrote", name, sorted(rec.keys())[:6], '...')

"""Device-resident conelp / coneqp loops with 's' (positive semidefinite) cones against committed fixtures produced by the
REAL reference (tests/golden/make_golden_sdp.py -> tests/golden/sdp_*.npz): the SDP of the reference's documentation
(examples/doc/chap8/sdp.py), the max-cut relaxation of examples/doc/chap8/mcsdp.py, mixed 'l' + 'q' + 's' cone programs
with equality constraints, a cone QP, an infeasible program.

Tolerances: same status and iteration count; objectives 1e-8 relative; x, s, z 1e-6 relative in the max norm (the
Nesterov-Todd scaling of an 's' block is unique only up to a signed permutation of its columns, which cancels in every
quantity of the original space, so these compare directly); the per-iteration table the reference prints to its printed
precision."""
import contextlib
import io
import os
import re

import numpy as np
import pytest

import cvxopt_amd
from cvxopt_amd import kkt

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LINE = re.compile(r"^\s*(\d+):\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)\s+(\S+)(?:\s+(\S+))?\s*$")


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def dims_of(g):
    return {'l': int(g['dims_l']), 'q': [int(k) for k in g['dims_q']], 's': [int(k) for k in g['dims_s']]}


def relerr(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b))))


def table_of(text):
    rows = []
    for ln in text.splitlines():
        mm = LINE.match(ln)
        if mm:
            rows.append([float(v) if v is not None else np.nan for v in mm.groups()[1:]])
    return np.array(rows)


def check_table(got, ref):
    assert got.shape[0] == ref.shape[0], (got.shape, ref.shape)
    for k in range(ref.shape[0]):
        for c in (0, 1):
            assert abs(got[k, c] - ref[k, c]) <= 1.01e-4 * abs(ref[k, c]) + 1e-9, (k, c, got[k, c], ref[k, c])
        for c in (2, 3, 4):
            if ref[k, c] > 1e-12:
                assert 0.45 <= got[k, c] / ref[k, c] <= 2.2, (k, c, got[k, c], ref[k, c])


def run_with_progress(fn):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sol = fn()
    return sol, table_of(buf.getvalue())


def lower_of(v, dims):
    """the entries the reference defines: everything of the 'l' / 'q' parts, the lower triangles of the 's' blocks"""
    out = [v[:dims['l'] + sum(dims['q'])]]
    ind = dims['l'] + sum(dims['q'])
    for m in dims['s']:
        out.append(np.tril(v[ind:ind + m * m].reshape(m, m, order='F')).ravel())
        ind += m * m
    return np.concatenate(out)


def poison_upper(M, dims, rng):
    """garbage in the strict upper triangles of the 's' blocks of every column: the reference never reads them"""
    M = np.array(M, dtype=float, order='F', copy=True)
    cols = M.reshape(M.shape[0], -1)
    ind = dims['l'] + sum(dims['q'])
    for m in dims['s']:
        iu = np.triu_indices(m, 1)
        for j in range(cols.shape[1]):
            blk = cols[ind:ind + m * m, j].reshape(m, m, order='F').copy()
            blk[iu] = 1e3 * rng.standard_normal(len(iu[0]))
            cols[ind:ind + m * m, j] = blk.ravel(order='F')
        ind += m * m
    return M


def check_solution(sol, g, dims, tol=1e-6):
    assert sol['status'] == str(g['status'])
    assert sol['iterations'] == int(g['iterations'])
    for k, key in (('primal objective', 'pobj'), ('dual objective', 'dobj'), ('gap', 'gap')):
        ref = float(g[key])
        if key == 'gap':
            assert abs(sol[k] - ref) <= 1e-4 * abs(ref) + 1e-12
        else:
            assert abs(sol[k] - ref) <= 1e-8 * max(1.0, abs(ref)), k
    assert relerr(sol['x'], g['x']) < tol
    if g['y'].size:
        assert relerr(sol['y'], g['y']) < 10 * tol
    assert relerr(lower_of(sol['s'], dims), lower_of(g['s'], dims)) < tol
    assert relerr(lower_of(sol['z'], dims), lower_of(g['z'], dims)) < tol
    # the loops keep (and return) both triangles, like the reference after its final misc.symm (coneprog.py:955-959)
    ind = dims['l'] + sum(dims['q'])
    for m in dims['s']:
        for v in (sol['s'], sol['z']):
            blk = v[ind:ind + m * m].reshape(m, m, order='F')
            assert np.array_equal(blk, blk.T)
        ind += m * m


@pytest.mark.parametrize("name", ["sdp_doc", "sdp_mc20", "sdp_mc60", "sdp_mixed"])
@pytest.mark.parametrize("kktsolver", ["chol", "ldl"])
def test_conelp_device_loop_with_s_cones_vs_reference_fixture(name, kktsolver):
    g = gold(name)
    dims = dims_of(g)
    A, b = (g['A'], g['b']) if g['A'].shape[0] else (None, None)
    sol, tab = run_with_progress(lambda: cvxopt_amd.conelp_device(g['c'], np.asfortranarray(g['G']), g['h'], dims, A, b,
                                                                  kktsolver=kktsolver, show_progress=True))
    check_solution(sol, g, dims)
    check_table(tab, g['table'])


def test_block_beyond_the_lds_resident_jacobi():
    """order 150: the eigendecompositions / the SVD of the scaling update run out of global memory (the LDS-resident Jacobi
    holds blocks up to 142 / 101), the KKT assembly takes the FP64-MFMA congruence path"""
    g = gold("sdp_mc150")
    dims = dims_of(g)
    sol, tab = run_with_progress(lambda: cvxopt_amd.conelp_device(g['c'], np.asfortranarray(g['G']), g['h'], dims,
                                                                  show_progress=True))
    check_solution(sol, g, dims)
    check_table(tab, g['table'])


def test_only_the_lower_triangles_of_G_and_h_count():
    g = gold("sdp_mixed")
    dims = dims_of(g)
    rng = np.random.default_rng(0)
    G = poison_upper(g['G'], dims, rng)
    h = poison_upper(g['h'].reshape(-1, 1), dims, rng).ravel()
    sol = cvxopt_amd.conelp_device(g['c'], G, h, dims, g['A'], g['b'])
    check_solution(sol, g, dims)


def test_coneqp_device_loop_with_s_cones_vs_reference_fixture():
    g = gold("sdp_qp")
    dims = dims_of(g)
    sol, tab = run_with_progress(lambda: cvxopt_amd.coneqp_device(g['P'], g['q'], np.asfortranarray(g['G']), g['h'], dims,
                                                                  g['A'], g['b'], show_progress=True))
    check_solution(sol, g, dims)
    check_table(tab, g['table'])


def test_primal_infeasible_sdp_certificate():
    g = gold("sdp_pinf")
    dims = dims_of(g)
    sol = cvxopt_amd.conelp_device(g['c'], np.asfortranarray(g['G']), g['h'], dims)
    assert sol['status'] == 'primal infeasible' == str(g['status'])
    assert sol['iterations'] == int(g['iterations'])
    assert sol['x'] is None and sol['s'] is None
    assert relerr(lower_of(sol['z'], dims), lower_of(g['z'], dims)) < 1e-6
    assert sol['dual objective'] == 1.0


def test_solvers_sdp_runs_the_device_loop_and_matches_the_documented_answer(ref_cvxopt):
    """cvxopt_amd.solvers.sdp on the data of examples/doc/chap8/sdp.py: the documented x (doc/source/coneprog.rst) and the
    reference fixture; zs come back as matrices."""
    from cvxopt import matrix
    from cvxopt_amd import solvers
    g = gold("sdp_doc")
    G = g['G']
    sol = solvers.sdp(matrix(g['c']), Gs=[matrix(G[:4, :]), matrix(G[4:, :])],
                      hs=[matrix(g['h'][:4].reshape(2, 2, order='F')), matrix(g['h'][4:].reshape(3, 3, order='F'))])
    assert sol['status'] == 'optimal' and sol['iterations'] == int(g['iterations'])
    x = np.array(sol['x']).ravel()
    assert np.allclose(x, [-0.367, 1.90, -0.888], atol=2e-3)             # as printed in the documentation
    assert relerr(x, g['x']) < 1e-6
    assert sol['zs'][0].size == (2, 2) and sol['zs'][1].size == (3, 3)
    z1 = np.array(sol['zs'][1])
    assert relerr(np.tril(z1), np.tril(g['z'][4:].reshape(3, 3, order='F'))) < 1e-6


def test_many_small_and_one_larger_block():
    """a program the fixtures do not cover in shape: 12 blocks of order 2..4 plus one of order 40, checked through the
    optimality conditions of the returned point (feasibility, complementarity, zero duality gap) instead of a fixture"""
    rng = np.random.default_rng(7)
    dims = {'l': 3, 'q': [], 's': [2, 3, 4] * 4 + [40]}
    n = 12
    cols = []
    for _ in range(n):
        parts = [rng.standard_normal(dims['l'])]
        for m in dims['s']:
            a = rng.standard_normal((m, m))
            parts.append((0.5 * (a + a.T)).ravel(order='F'))
        cols.append(np.concatenate(parts))
    G = np.asfortranarray(np.array(cols).T)

    def interior():
        parts = [rng.random(dims['l']) + 0.5]
        for m in dims['s']:
            a = rng.standard_normal((m, m))
            parts.append((a @ a.T / m + 0.5 * np.eye(m)).ravel(order='F'))
        return np.concatenate(parts)
    x0 = rng.standard_normal(n)
    h = G @ x0 + interior()
    c = -(G.T @ interior())
    sol = cvxopt_amd.conelp_device(c, G, h, dims)
    assert sol['status'] == 'optimal'
    x, s, z = sol['x'], sol['s'], sol['z']
    assert np.linalg.norm(G @ x + s - h) <= 1e-6 * max(1.0, np.linalg.norm(h))
    assert np.linalg.norm(G.T @ z + c) <= 1e-6 * max(1.0, np.linalg.norm(c))
    assert abs(c @ x + h @ z) <= 1e-5 * max(1.0, abs(c @ x))
    assert sol['primal slack'] > -1e-8 and sol['dual slack'] > -1e-8


@pytest.mark.parametrize("nb,mk,extra", [(100, 3, []), (40, 5, [30]), (70, 12, [])])
def test_many_blocks_one_per_wave(nb, mk, extra):
    """programs with many small blocks: blocks of order <= 16 are handled one per WAVE, 16 at a time (cone_ops.h, ParWave), the
    larger ones by the whole workgroup; checked through the optimality conditions of the returned point"""
    rng = np.random.default_rng(nb)
    dims = {'l': 2, 'q': [], 's': [mk] * nb + extra}
    n = 14
    cols = []
    for _ in range(n):
        parts = [rng.standard_normal(dims['l'])]
        for m in dims['s']:
            a = rng.standard_normal((m, m))
            parts.append((0.5 * (a + a.T)).ravel(order='F'))
        cols.append(np.concatenate(parts))
    G = np.asfortranarray(np.array(cols).T)

    def interior():
        parts = [rng.random(dims['l']) + 0.5]
        for m in dims['s']:
            a = rng.standard_normal((m, m))
            parts.append((a @ a.T / m + 0.5 * np.eye(m)).ravel(order='F'))
        return np.concatenate(parts)
    x0 = rng.standard_normal(n)
    h = G @ x0 + interior()
    c = -(G.T @ interior())
    for rep in range(3):                                    # repeated: the wave teams run concurrently
        sol = cvxopt_amd.conelp_device(c, G, h, dims)
        assert sol['status'] == 'optimal'
        x, s, z = sol['x'], sol['s'], sol['z']
        assert np.linalg.norm(G @ x + s - h) <= 1e-6 * max(1.0, np.linalg.norm(h))
        assert np.linalg.norm(G.T @ z + c) <= 1e-6 * max(1.0, np.linalg.norm(c))
        assert abs(c @ x + h @ z) <= 1e-5 * max(1.0, abs(c @ x))
        assert sol['primal slack'] > -1e-8 and sol['dual slack'] > -1e-8
        if rep == 0:
            first = x.copy()
        else:
            assert np.array_equal(first, x)                  # fixed-order arithmetic: bit-identical from run to run

"""CPU execution of the sparse engine's symbolic plan (no GPU): the host analysis of csrc/sparse_chol.hip + csrc/ordering.cpp
hands the device kernels a plan -- permutation, supernodes and their row lists, storage offsets of panels and update
matrices (two layouts: small fronts / full frontal matrices for the dense MFMA kernels), extend-add index maps and the
fixed-order assembly lists of S = H + G'D^2G.  Here the same plan is executed with NumPy, mirroring what the kernels do
with it (sp_assemble_kernel, the extend-add of sp_front_kernel / sp_extend_add_vb_kernel, a partial Cholesky per front,
supernodal substitution), and the result is checked against the matrix itself: L L' = P S P' and S x = b.  This pins
the plan for every ordering and for patterns the GPU tests do not reach."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sla
import scipy.sparse as sp

from test_ordering_cpu import delaunay, grid, preferential_attachment, random_graph


class Plan(object):
    def __init__(self, G, H):
        from cvxopt_amd import _capi
        L = _capi.lib()
        G = sp.csc_matrix(G); G.sort_indices()
        H = sp.csc_matrix(sp.tril(H)); H.sort_indices()
        self.G, self.H = G, H
        m, n = G.shape
        a = [np.ascontiguousarray(v, dtype=np.int64) for v in (G.indptr, G.indices, H.indptr, H.indices)]
        ptr = [v.ctypes.data_as(_capi.c_i64_p) for v in a]
        need = L.mi355kkt_test_symbolic_plan(n, m, ptr[0], ptr[1], ptr[2], ptr[3], None, 0)
        assert need > 0, need
        out = np.zeros(need, dtype=np.int64)
        got = L.mi355kkt_test_symbolic_plan(n, m, ptr[0], ptr[1], ptr[2], ptr[3], out.ctypes.data_as(_capi.c_i64_p), need)
        assert got == need
        (self.n, self.ns, self.nlevels, self.store, nt, nc, self.method, nrows, nch, nrel, nvb, nheavy) = [int(v) for v in out[:12]]
        pos = [16]

        def take(k):
            v = out[pos[0]:pos[0] + k]
            pos[0] += k
            return v
        ns = self.ns
        self.perm = take(self.n)
        self.sn_first, self.sn_rowptr = take(ns + 1), take(ns + 1)
        self.sn_rows = take(nrows)
        self.panel_off, self.upd_off, self.upd_ld, self.big, self.sn_level = take(ns + 1), take(ns), take(ns), take(ns), take(ns)
        self.child_ptr, self.child_list = take(ns + 1), take(nch)
        self.relmap_off, self.relmap = take(ns + 1), take(nrel)
        self.asm_slot, self.asm_ptr = take(nt), take(nt + 1)
        self.asm_a, self.asm_b, self.asm_r = take(nc), take(nc), take(nc)
        nl = self.nlevels
        self.level_ptr, self.level_sn, self.level_nsmall = take(nl + 1), take(ns), take(nl)
        self.vb_ptr, self.vb = take(nl + 1), take(5 * nvb).reshape(nvb, 5)
        self.heavy_ptr, self.heavy = take(nl + 1), take(nheavy)
        assert pos[0] == need

    def S(self, di):
        H = self.H + sp.tril(self.H, -1).T
        return (H + self.G.T @ sp.diags(di ** 2) @ self.G).tocsc()

    def check_structure(self):
        n, ns = self.n, self.ns
        assert sorted(self.perm.tolist()) == list(range(n))
        assert self.sn_first[0] == 0 and self.sn_first[ns] == n and np.all(np.diff(self.sn_first) > 0)
        assert np.all(np.diff(self.sn_first) <= 8192)
        ends = []
        for s in range(ns):
            f, l = self.sn_first[s], self.sn_first[s + 1]
            rows = self.sn_rows[self.sn_rowptr[s]:self.sn_rowptr[s + 1]]
            w, h = l - f, len(rows)
            assert np.array_equal(rows[:w], np.arange(f, l)) and np.all(np.diff(rows) > 0)      # own columns, then sorted below-rows
            if self.big[s]:      # full h x h frontal matrix, update matrix = its trailing block
                assert self.upd_off[s] == self.panel_off[s] + w + w * h and self.upd_ld[s] == h
                ends.append((self.panel_off[s], self.panel_off[s] + h * h))
            else:                # h x w panel followed by the (h-w)^2 update matrix
                assert self.upd_off[s] == self.panel_off[s] + h * w and self.upd_ld[s] == h - w
                ends.append((self.panel_off[s], self.panel_off[s] + h * w + (h - w) ** 2))
            for c in self.child_list[self.child_ptr[s]:self.child_ptr[s + 1]]:
                assert c < s and self.sn_level[c] < self.sn_level[s]
        ends.sort()
        assert all(a[1] <= b[0] for a, b in zip(ends, ends[1:])) and (not ends or ends[-1][1] <= self.store)   # no overlap
        assert np.all(self.panel_off[:ns] % 2 == 0)                                                    # 16-byte aligned fronts
        assert self.nlevels == (self.sn_level.max() + 1 if ns else 0)
        # the level schedule: every supernode once, at its level; small fronts first (one batched launch), then the big
        # ones in the order of the descriptor list the level-batched dense kernels read; the heavy list of the solves
        assert sorted(self.level_sn.tolist()) == list(range(ns))
        hgt = np.diff(self.sn_rowptr)
        wid = np.diff(self.sn_first)
        for l in range(self.nlevels):
            sl = self.level_sn[self.level_ptr[l]:self.level_ptr[l + 1]]
            assert len(sl) > 0 and np.all(self.sn_level[sl] == l)
            k = int(self.level_nsmall[l])
            assert not self.big[sl[:k]].any() and self.big[sl[k:]].all()
            vb = self.vb[self.vb_ptr[l]:self.vb_ptr[l + 1]]
            assert np.array_equal(vb[:, 4], sl[k:])
            assert np.array_equal(vb[:, 0], self.panel_off[sl[k:]]) and np.array_equal(vb[:, 1], hgt[sl[k:]])
            assert np.array_equal(vb[:, 2], wid[sl[k:]]) and np.array_equal(vb[:, 3], self.sn_first[sl[k:]])
            heavy = self.heavy[self.heavy_ptr[l]:self.heavy_ptr[l + 1]]
            expect = [int(s_) for s_ in sl if (hgt[s_] - wid[s_]) * wid[s_] > 32768 or (wid[s_] > 128 and hgt[s_] > wid[s_])]       # 128 = sp_wide_threshold()
            assert heavy.tolist() == expect

    def factor(self, di):
        """executes the plan; returns the dense L (permuted order)"""
        n, ns = self.n, self.ns
        gv, hv = self.G.data, self.H.data
        store = np.zeros(self.store)
        cnt = np.diff(self.asm_ptr)
        vals = np.where(self.asm_b < 0, hv[np.minimum(self.asm_a, len(hv) - 1)] if len(hv) else 0.0,
                        gv[np.minimum(self.asm_a, len(gv) - 1)] * gv[np.maximum(self.asm_b, 0)] * di[self.asm_r] ** 2)
        assert len(np.unique(self.asm_slot)) == len(self.asm_slot)                  # one target per structural nonzero
        np.add.at(store, np.repeat(self.asm_slot, cnt), vals)
        L = np.zeros((n, n))
        U = {}
        for s in range(ns):
            f, l = int(self.sn_first[s]), int(self.sn_first[s + 1])
            rows = self.sn_rows[self.sn_rowptr[s]:self.sn_rowptr[s + 1]]
            w, h = l - f, len(rows)
            F = np.zeros((h, h))
            F[:, :w] = store[self.panel_off[s]:self.panel_off[s] + h * w].reshape(w, h).T
            for c in self.child_list[self.child_ptr[s]:self.child_ptr[s + 1]]:
                Uc = U.pop(int(c))
                rm = self.relmap[self.relmap_off[c]:self.relmap_off[c + 1]]
                assert len(rm) == Uc.shape[0] and np.all(np.diff(rm) > 0) and np.all(rm < h)
                crow = self.sn_rows[self.sn_rowptr[c]:self.sn_rowptr[c + 1]][self.sn_first[c + 1] - self.sn_first[c]:]
                assert np.array_equal(rows[rm], crow)                               # the map lands on the same global rows
                F[np.ix_(rm, rm)] += np.tril(Uc)
            A11 = np.tril(F[:w, :w]) + np.tril(F[:w, :w], -1).T
            L11 = np.linalg.cholesky(A11)
            L21 = sla.solve_triangular(L11, F[w:, :w].T, lower=True).T if h > w else np.zeros((0, w))
            U[s] = np.tril(F[w:, w:]) - np.tril(L21 @ L21.T)
            L[f:l, f:l] = L11
            L[rows[w:], f:l] = L21
        assert all(v.shape[0] == 0 for v in U.values())                             # only roots keep an (empty) update matrix
        return L

    def solve(self, L, b):
        y = sla.solve_triangular(L, b[self.perm], lower=True)
        y = sla.solve_triangular(L.T, y, lower=False)
        x = np.empty_like(y)
        x[self.perm] = y
        return x


def lap3(k):
    e = np.ones(k)
    T = sp.diags([-e[:-1], 2 * e, -e[:-1]], [-1, 0, 1])
    I = sp.eye(k)
    return (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T) + 0.01 * sp.eye(k ** 3)).tocsc()


def box(n):
    return sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()


def coupled_G(n, m, seed):
    """general G: rows couple a few variables each (S gets fill from G'D^2G that H does not have)"""
    return (sp.random(m, n, density=3.0 / n, random_state=seed, format='csc')
            + sp.vstack([sp.eye(n), sp.csc_matrix((m - n, n))])).tocsc()


PROBLEMS = {
    'grid 30x23, box': lambda: (box(690), grid(30, 23)),
    'laplace 11^3, box (big fronts)': lambda: (box(1331), lap3(11)),
    'mesh, box': lambda: (box(1500), delaunay(1500, seed=4)),
    'hubs, box': lambda: (box(900), preferential_attachment(900, seed=1)),
    'random graph, box': lambda: (box(700), random_graph(700, 2, seed=2)),
    'coupled G, grid H': lambda: (coupled_G(600, 900, 5), grid(30, 20)),
    'coupled G, no H': lambda: (coupled_G(400, 700, 6), sp.csc_matrix((400, 400))),
    'diagonal H, box': lambda: (box(50), sp.eye(50).tocsc()),
    'one variable': lambda: (sp.csc_matrix(np.array([[2.0], [-1.0]])), sp.csc_matrix(np.array([[3.0]]))),
}
ORDERINGS = {
    'auto': {}, 'amd': {'MI355KKT_ORDERING': 'amd'}, 'nd': {'MI355KKT_ORDERING': 'nd'},
    'nd, minimum-degree leaves': {'MI355KKT_ORDERING': 'nd', 'MI355KKT_ND_LEAF_AMD': '1'},
    'nd, multilevel only': {'MI355KKT_ORDERING': 'nd', 'MI355KKT_ND_MODE': '2'},
    'nd, level sets only, no refinement': {'MI355KKT_ORDERING': 'nd', 'MI355KKT_ND_MODE': '1', 'MI355KKT_ND_NOREFINE': '1'},
}


@pytest.mark.parametrize("ordering", list(ORDERINGS))
@pytest.mark.parametrize("problem", list(PROBLEMS))
def test_plan_executes_to_the_cholesky_factor(problem, ordering, knobs):
    for k, v in ORDERINGS[ordering].items():
        knobs.setenv(k, v)
    G, H = PROBLEMS[problem]()
    plan = Plan(G, H)
    plan.check_structure()
    rng = np.random.default_rng(len(problem))
    di = 10.0 ** rng.uniform(-1, 1, G.shape[0])
    S = plan.S(di).toarray()
    L = plan.factor(di)
    Sp = S[np.ix_(plan.perm, plan.perm)]
    assert np.linalg.norm(L @ L.T - Sp) <= 1e-12 * np.linalg.norm(Sp)
    b = rng.standard_normal(plan.n)
    x = plan.solve(L, b)
    assert np.linalg.norm(S @ x - b) <= 1e-10 * np.linalg.norm(b)


def test_big_front_threshold_switches_the_storage_layout(knobs):
    """every front as a full frontal matrix / every front as panel + update matrix: same factor"""
    G, H = PROBLEMS['laplace 11^3, box (big fronts)']()
    di = np.ones(G.shape[0])
    out = []
    for flops, hmin in ((0, 1), (1e30, 1 << 30)):
        knobs.setenv('MI355KKT_SPARSE_BIG_FLOPS', repr(flops))
        knobs.setenv('MI355KKT_SPARSE_BIG_H', str(hmin))
        plan = Plan(G, H)
        plan.check_structure()
        wide = np.diff(plan.sn_first) > 128            # a wide supernode is a big front whatever the thresholds say (round 4)
        assert bool(plan.big.all()) if flops == 0 else np.array_equal(plan.big.astype(bool), wide)
        out.append((plan.perm.copy(), plan.factor(di)))
    assert np.array_equal(out[0][0], out[1][0]) and np.allclose(out[0][1], out[1][1], rtol=0, atol=1e-13)


@pytest.mark.parametrize("seed", range(4))
def test_random_patterns_orderings_and_thresholds(seed, knobs):
    """seeded fuzz: random H / G patterns (empty rows, dense-ish rows, disconnected parts), a random ordering variant and
    random big-front thresholds per case"""
    rng = np.random.default_rng(100 + seed)
    done = 0
    while done < 8:
        n = int(rng.integers(1, 220))
        m = int(rng.integers(1, 2 * n + 2))
        dens = float(rng.choice([0.0, 0.5 / n, 2.0 / n, 8.0 / n, 0.3]))
        H = sp.random(n, n, density=min(1.0, dens), random_state=int(rng.integers(1 << 30)), format='csc')
        H = H + H.T
        H = (H + sp.diags(np.asarray(abs(H).sum(1)).ravel() + float(rng.choice([0.0, 1.0])))).tocsc()
        G = sp.random(m, n, density=min(1.0, float(rng.choice([1.0 / n, 3.0 / n, 0.2]))),
                      random_state=int(rng.integers(1 << 30)), format='csc')
        G = sp.vstack([G, sp.eye(n)]).tocsc()
        name = list(ORDERINGS)[int(rng.integers(len(ORDERINGS)))]
        for k in ('MI355KKT_ORDERING', 'MI355KKT_ND_LEAF_AMD', 'MI355KKT_ND_MODE', 'MI355KKT_ND_NOREFINE',
                  'MI355KKT_SPARSE_BIG_FLOPS', 'MI355KKT_SPARSE_BIG_H'):
            knobs.delenv(k, raising=False)
        for k, v in ORDERINGS[name].items():
            knobs.setenv(k, v)
        if rng.random() < 0.4:
            knobs.setenv('MI355KKT_SPARSE_BIG_FLOPS', repr(float(rng.choice([0, 1e3, 1e30]))))
            knobs.setenv('MI355KKT_SPARSE_BIG_H', str(int(rng.choice([1, 8, 48]))))
        di = 10.0 ** rng.uniform(-1, 1, G.shape[0])
        plan = Plan(G, H)
        plan.check_structure()
        S = plan.S(di).toarray()
        L = plan.factor(di)
        Sp = S[np.ix_(plan.perm, plan.perm)]
        assert np.linalg.norm(L @ L.T - Sp) <= 1e-11 * np.linalg.norm(Sp), (n, m, name)
        done += 1


def test_structure_of_a_larger_plan_with_heavy_supernodes():
    """structure only (no NumPy factorisation): a 3-D grid whose top separators give supernodes with large off-diagonal
    panels -- the ones the solves hand to the multi-workgroup kernels"""
    n = 22 ** 3
    plan = Plan(box(n), lap3(22))
    plan.check_structure()
    assert len(plan.heavy) > 0 and plan.big.any() and not plan.big.all()
    assert plan.nlevels < 40


@pytest.mark.parametrize("maxw", ["8", "256", None])
def test_supernode_width_cap_changes_the_partition_not_the_factor(maxw, knobs):
    """$MI355KKT_SN_MAXW: chains of narrow pieces (8), the round-1 cap (256), the default (8192: the 16^3 grid's root separator
    of 256 columns and its children become single wide supernodes, more than the 128 columns of sp_wide_threshold()): the
    executed plan is the same Cholesky factor, with fewer levels the wider the supernodes may be"""
    knobs.setenv('MI355KKT_ORDERING', 'nd')
    if maxw is not None:
        knobs.setenv('MI355KKT_SN_MAXW', maxw)
    G, H = box(4096), lap3(16)
    plan = Plan(G, H)
    plan.check_structure()
    wid = np.diff(plan.sn_first)
    di = np.ones(G.shape[0])
    L = plan.factor(di)
    Sp = plan.S(di).toarray()[np.ix_(plan.perm, plan.perm)]
    assert np.linalg.norm(L @ L.T - Sp) <= 1e-12 * np.linalg.norm(Sp)
    if maxw is None:
        assert 128 < wid.max() <= 8192                        # the root separator in one piece
    else:
        assert wid.max() <= int(maxw)
    test_supernode_width_cap_changes_the_partition_not_the_factor.levels[maxw] = plan.nlevels
    lv = test_supernode_width_cap_changes_the_partition_not_the_factor.levels
    if len(lv) == 3:
        assert lv["8"] > lv["256"] >= lv[None]


test_supernode_width_cap_changes_the_partition_not_the_factor.levels = {}

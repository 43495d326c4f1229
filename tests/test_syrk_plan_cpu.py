"""CPU test of the static SYRK schedule (host logic of csrc/gemm_f64.hip, `make_syrk_items`): every lower-triangular tile
is produced exactly once -- either by one full-K segment or by split pieces whose k-ranges partition [0, K) on BK
boundaries and whose slab slots are consecutive -- for the shapes the engine builds plans for; the remainder round is
stream-K (every workgroup of it owns the same number of k-steps, +-1, as a chain of segments)."""
import ctypes as C

import numpy as np
import pytest

from cvxopt_amd import _capi

TILE, BK = 128, 16


def plan(n, K, cus=256, split=True):
    L = _capi.lib()
    cap = 1 << 16
    out = np.zeros(8 * cap, dtype=np.int32)
    ns, nsp, nl = C.c_int(), C.c_int(), C.c_int()
    cnt = L.mi355kkt_test_syrk_plan(n, K, cus, 1 if split else 0, out.ctypes.data_as(_capi.c_int_p), cap, C.byref(ns), C.byref(nsp),
                                    C.byref(nl))
    assert 0 <= cnt <= cap
    return out[:8 * cnt].reshape(cnt, 8), ns.value, nsp.value, nl.value


SHAPES = [(8192, 16384, 256, True), (2048, 8192, 256, True), (512, 1024, 256, False), (300, 77, 256, True), (1, 5, 256, True),
          (4096, 4096, 104, True), (1000, 100000, 256, True), (129, 16, 8, True), (1024, 8192, 256, True), (4096, 8192, 256, True),
          (8192, 4096, 256, True), (2176, 8192, 256, True)]


@pytest.mark.parametrize("n,K,cus,split", SHAPES)
def test_every_lower_tile_is_covered_exactly_once(n, K, cus, split):
    items, nslabs, nsplit, nlaunch = plan(n, K, cus, split)
    nt = (n + TILE - 1) // TILE
    full, pieces = {}, {}
    for ti, tj, k0, k1, slot, first, nparts, nxt in items:
        assert 0 <= tj <= ti < nt
        assert 0 <= k0 < k1 <= K
        if slot < 0:
            assert (k0, k1) == (0, K)
            full[(ti, tj)] = full.get((ti, tj), 0) + 1
        else:
            assert split
            pieces.setdefault((ti, tj), []).append((k0, k1, slot, first, nparts))
    assert not (set(full) & set(pieces))
    assert all(v == 1 for v in full.values())
    assert len(full) + len(pieces) == nt * (nt + 1) // 2              # the whole lower triangle, nothing else
    assert len(pieces) == nsplit
    used = set()
    for (ti, tj), ps in pieces.items():
        ps.sort()
        first, nparts = ps[0][3], ps[0][4]
        assert len(ps) == nparts and nparts >= 2
        assert ps[0][0] == 0 and ps[-1][1] == K
        for a, b in zip(ps, ps[1:]):
            assert a[1] == b[0] and a[1] % BK == 0                       # contiguous, on BK boundaries
        assert [p[2] for p in ps] == list(range(first, first + nparts))  # consecutive slab slots in k order
        assert all(p[3] == first and p[4] == nparts for p in ps)
        used.update(p[2] for p in ps)
    assert used == set(range(nslabs))


@pytest.mark.parametrize("n,K,cus,split", SHAPES)
def test_chains_reach_every_segment_once_and_balance_the_remainder_round(n, K, cus, split):
    items, nslabs, nsplit, nlaunch = plan(n, K, cus, split)
    assert 0 < nlaunch <= len(items)
    seen = np.zeros(len(items), dtype=int)
    loads = []
    for w in range(nlaunch):
        i, steps, hops = w, 0, 0
        while True:
            seen[i] += 1
            ti, tj, k0, k1, slot, first, nparts, nxt = items[i][:8]
            if slot >= 0:
                steps += (k1 - k0 + BK - 1) // BK
            if nxt == 0:
                break
            assert nxt - 1 >= nlaunch                                   # continuation segments are not launched themselves
            assert slot >= 0
            i, hops = nxt - 1, hops + 1
            assert hops <= 2
        if steps:
            loads.append(steps)
    assert np.all(seen == 1)
    if loads:                                                           # stream-K: equal shares of the remainder round's k-steps
        assert max(loads) - min(loads) <= 1
        assert len(loads) <= 2 * cus


def test_headline_shape_fills_the_last_round():
    """config 2: 2080 tiles on 512 slots -> 4 full rounds + 32 tiles split 16 ways = one more full round of 512 pieces."""
    items, nslabs, nsplit, nlaunch = plan(8192, 16384)
    assert nsplit == 32 and nslabs == 512 and len(items) == 2048 + 512 == nlaunch
    assert np.all(items[:2048, 4] < 0) and np.all(items[2048:, 4] >= 0)
    assert set((items[2048:, 3] - items[2048:, 2]).tolist()) == {1024}


def test_socp_shape_is_balanced_over_all_slots():
    """config 3 (n = 2048, K = 8192): 136 tiles of 512 k-steps on 512 slots -> 136 k-steps per workgroup, two segments at most."""
    items, nslabs, nsplit, nlaunch = plan(2048, 8192)
    assert nlaunch == 512 and nsplit == 136
    assert 512 < len(items) <= 512 + 136


@pytest.mark.parametrize("n,K,cus", [(300, 77, 256), (640, 2048, 8), (1000, 3000, 4), (129, 16, 8), (513, 1111, 16), (2048, 512, 256)])
def test_numpy_execution_of_the_plan_gives_the_scaled_syrk(n, K, cus):
    """the plan executed the way the kernel executes it -- every launched workgroup walks its chain of segments, a segment adds its
    k range of its tile to C (final items) or writes a slab, the reducer sums a split tile's slabs in slot order -- is
    S = P + G' diag(di)^2 G on the lower triangle"""
    rng = np.random.default_rng(n + K)
    G = rng.standard_normal((K, n))
    di = rng.uniform(0.5, 2.0, K)
    P = rng.standard_normal((n, n))
    P = P + P.T
    items, nslabs, nsplit, nlaunch = plan(n, K, cus)
    Gs = G * di[:, None]
    C = np.full((n, n), np.nan)
    slabs = {}
    written = set()
    for w in range(nlaunch):
        i = w
        while True:
            ti, tj, k0, k1, slot, first, nparts, nxt = (int(v) for v in items[i])
            I = slice(ti * TILE, min(n, (ti + 1) * TILE))
            J = slice(tj * TILE, min(n, (tj + 1) * TILE))
            part = Gs[k0:k1, I].T @ Gs[k0:k1, J]
            if slot < 0:
                assert (ti, tj) not in written
                written.add((ti, tj))
                C[I, J] = P[I, J] + part
            else:
                assert slot not in slabs
                slabs[slot] = (ti, tj, first, nparts, part)
            if nxt == 0:
                break
            i = nxt - 1
    assert sorted(slabs) == list(range(nslabs))
    for slot, (ti, tj, first, nparts, part) in sorted(slabs.items()):
        if slot != first:
            continue
        I = slice(ti * TILE, min(n, (ti + 1) * TILE))
        J = slice(tj * TILE, min(n, (tj + 1) * TILE))
        acc = P[I, J].copy()
        for s_ in range(first, first + nparts):                       # fixed order, as syrk_reduce_kernel
            assert slabs[s_][:2] == (ti, tj)
            acc = acc + slabs[s_][4]
        assert (ti, tj) not in written
        written.add((ti, tj))
        C[I, J] = acc
    ref = P + Gs.T @ Gs
    low = np.tril_indices(n)
    nt = (n + TILE - 1) // TILE
    assert len(written) == nt * (nt + 1) // 2
    assert np.all(np.isfinite(C[low]))
    assert np.max(np.abs(C[low] - ref[low])) <= 1e-11 * np.max(np.abs(ref))

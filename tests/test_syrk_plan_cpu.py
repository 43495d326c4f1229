"""CPU test of the static SYRK schedule (host logic of csrc/gemm_f64.hip, `make_syrk_items`): every lower-triangular tile
is produced exactly once -- either by one full-K work item or by split pieces whose k-ranges partition [0, K) on BK
boundaries and whose slab slots are consecutive -- for the shapes the engine builds plans for."""
import ctypes as C

import numpy as np
import pytest

from cvxopt_amd import _capi

TILE, BK = 128, 16


def plan(n, K, cus=256, split=True):
    L = _capi.lib()
    cap = 1 << 16
    out = np.zeros(8 * cap, dtype=np.int32)
    ns, nsp = C.c_int(), C.c_int()
    cnt = L.mi355kkt_test_syrk_plan(n, K, cus, 1 if split else 0, out.ctypes.data_as(_capi.c_int_p), cap, C.byref(ns), C.byref(nsp))
    assert 0 <= cnt <= cap
    return out[:8 * cnt].reshape(cnt, 8), ns.value, nsp.value


@pytest.mark.parametrize("n,K,cus,split", [(8192, 16384, 256, True), (2048, 8192, 256, True), (512, 1024, 256, False),
                                           (300, 77, 256, True), (1, 5, 256, True), (4096, 4096, 104, True),
                                           (1000, 100000, 256, True), (129, 16, 8, True)])
def test_every_lower_tile_is_covered_exactly_once(n, K, cus, split):
    items, nslabs, nsplit = plan(n, K, cus, split)
    nt = (n + TILE - 1) // TILE
    full, pieces = {}, {}
    for ti, tj, k0, k1, slot, first, nparts, _ in items:
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
        assert len(ps) == nparts
        assert ps[0][0] == 0 and ps[-1][1] == K
        for a, b in zip(ps, ps[1:]):
            assert a[1] == b[0] and a[1] % BK == 0                       # contiguous, on BK boundaries
        assert [p[2] for p in ps] == list(range(first, first + nparts))  # consecutive slab slots in k order
        assert all(p[3] == first and p[4] == nparts for p in ps)
        used.update(p[2] for p in ps)
    assert used == set(range(nslabs))


def test_headline_shape_fills_the_last_round():
    """config 2: 2080 tiles on 512 slots -> 4 full rounds + 32 tiles split 16 ways = one more full round of 512 pieces."""
    items, nslabs, nsplit = plan(8192, 16384)
    assert nsplit == 32 and nslabs == 512 and len(items) == 2048 + 512
    assert np.all(items[:2048, 4] < 0) and np.all(items[2048:, 4] >= 0)
    assert set((items[2048:, 3] - items[2048:, 2]).tolist()) == {1024}

"""bench.py's launch contract without a GPU: `--gpus N` (N > 1) re-executes itself under torch.distributed.run, one rank per
"GPU", and prints ONE JSON line from rank 0; with no device visible the ranks run the gloo / NumPy dry run of the
scatter -> solve -> gather plumbing (reported as such: value null, dry_run true) — a check of the contract, not a measurement."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

from cvxopt_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(_capi.device_count() > 0, reason="with a GPU the same command is the real multi-GPU benchmark")
def test_gpus_2_self_spawns_two_ranks_and_prints_one_json_line():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 0
    assert d["dry_run"] is True and d["value"] is None           # never a number without a GPU
    assert d["scaling"] == "strong" and d["config"]["problems_per_rank"] == [3, 3] and d["config"]["all_optimal"] is True
    for key in ("metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert key in d


def test_matrix_market_door_and_the_elasticity_stand_in(tmp_path):
    """`bench.py --workload sparse --mtx FILE` reads its P with synth.read_matrix_market: a symmetric file (lower triangle stored, the
    SuiteSparse convention) comes back as the full symmetric matrix; the 3-dof stand-in is symmetric positive definite with 3 x 3
    blocks; the symbolic analysis of the library accepts its pattern (host only)."""
    import numpy as np
    import scipy.io
    import scipy.sparse as sp
    import scipy.sparse.linalg as sla
    from cvxopt_amd import synth
    K = synth.tet_mesh_elasticity(400, seed=3)
    n = K.shape[0]
    assert n == 1200 and abs(K - K.T).max() == 0.0
    assert 30 < K.nnz / n < 60                                   # ~16 neighbours x 3 + the diagonal block
    assert sla.eigsh(K, k=1, which='SA', return_eigenvectors=False)[0] > 5e-3
    path = str(tmp_path / "k.mtx")
    scipy.io.mmwrite(path, sp.tril(K), symmetry='symmetric')
    A = synth.read_matrix_market(path)
    assert abs(A - K).max() == 0.0
    B = synth.read_matrix_market(path, shift=0.5)
    assert abs((B - K).diagonal() - 0.5 * np.max(np.abs(K.diagonal()))).max() < 1e-12
    # the box-QP wrapper of the bench: S = P + G'D^2G has P's pattern; the analysis (ordering + supernodes) runs on the host
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()
    G.sort_indices()
    Kt = sp.tril(K).tocsc()
    Kt.sort_indices()
    perm = np.zeros(n, dtype=np.int32)
    nnzL, ns, nl = C.c_int64(), C.c_int(), C.c_int()
    as64 = lambda a: np.ascontiguousarray(a, dtype=np.int64).ctypes.data_as(_capi.c_i64_p)
    rc = _capi.lib().mi355kkt_op_symbolic(n, 2 * n, as64(G.indptr), as64(G.indices), as64(Kt.indptr), as64(Kt.indices),
                                          perm.ctypes.data_as(_capi.c_int_p), C.byref(nnzL), C.byref(ns), C.byref(nl))
    assert rc == 0 and sorted(perm.tolist()) == list(range(n))
    assert nnzL.value >= Kt.nnz and 1 <= ns.value <= n


def test_matrix_market_pattern_file_and_non_positive_diagonal(tmp_path):
    """a `pattern` file gets the graph Laplacian of its pattern (+ I: positive definite), and a file whose diagonal is
    not positive is refused with the remedy instead of failing deep in the factorisation."""
    import numpy as np
    import pytest
    import scipy.sparse as sp
    import scipy.sparse.linalg as sla
    from cvxopt_amd import synth
    p = tmp_path / "ring.mtx"            # a 6-cycle, lower triangle only, no values
    edges = [(2, 1), (3, 2), (4, 3), (5, 4), (6, 5), (6, 1)]
    p.write_text("%%MatrixMarket matrix coordinate pattern symmetric\n6 6 %d\n" % len(edges)
                 + "".join("%d %d\n" % e for e in edges))
    A = synth.read_matrix_market(str(p))
    assert abs(A - A.T).max() == 0.0 and np.all(A.diagonal() == 3.0)          # degree 2 + 1
    assert abs(A).sum() == 6 * 3.0 + 12 * 1.0 and (A - sp.diags(A.diagonal())).max() == 0.0     # off-diagonal entries are -1
    assert sla.eigsh(A, k=1, which='SA', return_eigenvectors=False)[0] > 0.99
    q = tmp_path / "indef.mtx"
    q.write_text("%%MatrixMarket matrix coordinate real symmetric\n3 3 4\n1 1 2.0\n2 2 -1.0\n3 3 1.0\n3 1 0.5\n")
    with pytest.raises(ValueError, match="--mtx-shift"):
        synth.read_matrix_market(str(q))
    B = synth.read_matrix_market(str(q), shift=1.0)
    assert B.diagonal().min() == 1.0

"""CPU tests of the batched mode's host logic: the lock-step IPM restatement against individual reference
coneqp runs, and the N>1 sharding path on gloo (world_size 2)."""
import os
import sys

import numpy as np
import pytest

from cvxopt_amd import synth
from cvxopt_amd.batch import coneqp_batch, pack_problems, shard_bounds
from batch_helpers import NumpyBatchKkt, numpy_local_solver
from helpers import load_golden


def make_batch(B, n, m, seed0=0):
    return [synth.dense_qp(n, m, seed=seed0 + i) for i in range(B)]


def test_shard_bounds_cover_the_batch():
    for B in (0, 1, 7, 8, 4096, 4099):
        for w in (1, 2, 3, 8):
            b = shard_bounds(B, w)
            assert b[0][0] == 0 and b[-1][1] == B and all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_lockstep_loop_matches_golden_reference_run():
    """coneqp_batch (B=1) reproduces the golden reference coneqp run: iterations, objectives, x."""
    g = load_golden("coneqp_qp64")
    pr = synth.dense_qp(int(g['n']), int(g['m']), seed=int(g['seed']))
    P, q, Gt, h = pack_problems([pr])
    res = coneqp_batch(P, q, Gt, h, kkt=NumpyBatchKkt(Gt, P))
    assert res['status'][0] == 'optimal'
    assert res['iterations'][0] == int(g['iterations'])
    assert abs(res['primal objective'][0] - float(g['pobj'])) <= 1e-9 * abs(float(g['pobj']))
    assert abs(res['dual objective'][0] - float(g['dobj'])) <= 1e-9 * abs(float(g['dobj']))
    assert np.max(np.abs(res['x'][0] - g['x'])) <= 1e-7 * np.max(np.abs(g['x']))
    assert np.max(np.abs(res['z'][0] - g['z'])) <= 1e-6 * max(1.0, np.max(np.abs(g['z'])))


def test_lockstep_batch_matches_individual_reference_runs(ref_cvxopt):
    """Problems converge at different iteration counts; each must match its own solvers.coneqp run."""
    from cvxopt import matrix, solvers
    probs = make_batch(6, 24, 50, seed0=10) 
    probs[3]['h'] = probs[3]['h'] * 50.0           # different scales -> different iteration counts
    probs[4]['q'] = probs[4]['q'] * 1e-3
    P, q, Gt, h = pack_problems(probs)
    res = coneqp_batch(P, q, Gt, h, kkt=NumpyBatchKkt(Gt, P))
    its = []
    for b, pr in enumerate(probs):
        ref = solvers.coneqp(matrix(pr['P']), matrix(pr['q']), matrix(pr['G']), matrix(pr['h']), kktsolver='chol2')
        its.append(ref['iterations'])
        assert res['status'][b] == ref['status'] == 'optimal'
        assert res['iterations'][b] == ref['iterations'], (b, res['iterations'][b], ref['iterations'])
        assert abs(res['primal objective'][b] - ref['primal objective']) <= 1e-9 * max(1, abs(ref['primal objective']))
        assert np.max(np.abs(res['x'][b] - np.array(ref['x']).ravel())) <= 1e-7 * max(1, np.max(np.abs(np.array(ref['x']))))
    assert len(set(its)) > 1, "test should exercise the per-problem active mask"


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from cvxopt_amd.batch import coneqp_batch_sharded
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 0:
        probs = make_batch(5, 16, 40, seed0=3)       # 5 problems over 2 ranks: shards of 3 and 2
        P, q, Gt, h = pack_problems(probs)
    else:
        P = q = Gt = h = None
    res = coneqp_batch_sharded(P, q, Gt, h, local_solver=numpy_local_solver)
    if rank == 0:
        np.savez(tmp, x=res['x'], it=res['iterations'], pobj=res['primal objective'])
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_batch_on_gloo_world_size_2(tmp_path):
    import torch.multiprocessing as mp
    tmp = str(tmp_path / "out.npz")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, tmp), nprocs=2, join=True)
    got = np.load(tmp)
    probs = make_batch(5, 16, 40, seed0=3)
    P, q, Gt, h = pack_problems(probs)
    ref = coneqp_batch(P, q, Gt, h, kkt=NumpyBatchKkt(Gt, P))
    assert np.array_equal(got['it'], ref['iterations'])
    assert np.allclose(got['x'], ref['x'], rtol=0, atol=1e-12)
    assert np.allclose(got['pobj'], ref['primal objective'], rtol=1e-13)

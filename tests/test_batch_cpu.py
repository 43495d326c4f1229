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


def _worker(rank, world, port, tmp, nprob, nsub):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from cvxopt_amd import batch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    for rep in range(2):                             # the second call reuses the cached ShardedBatch (buffers, sub-batch plan)
        if rank == 0:
            probs = make_batch(nprob, 16, 40, seed0=3 + 100 * rep)
            P, q, Gt, h = pack_problems(probs)
        else:
            P = q = Gt = h = None
        res = batch.coneqp_batch_sharded(P, q, Gt, h, local_solver=numpy_local_solver, nsub=nsub)
        if rank == 0:
            out.update({"x%d" % rep: res['x'], "it%d" % rep: res['iterations'], "pobj%d" % rep: res['primal objective'],
                        "z%d" % rep: res['z']})
        else:                                        # the other ranks get their own shard back
            lo, hi = batch.shard_bounds(nprob, world)[rank]
            assert res['x'].shape[0] == hi - lo
    assert len(batch._SHARDED_CACHE) == 1
    sb = next(iter(batch._SHARDED_CACHE.values()))
    assert set(sb.last_timings) >= {"scatter_exposed", "scatter_all", "solve", "gather_exposed", "total"}
    if rank == 0:
        np.savez(tmp, **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nprob,nsub", [(2, 5, 4), (3, 7, 4), (3, 8, 2), (2, 6, 1), (4, 5, 3), (8, 19, 2), (4, 3, 2)])
def test_sharded_batch_on_gloo(tmp_path, world, nprob, nsub):
    """unequal shards (5 over 2, 7 over 3), sub-batches that are empty on the short ranks, the un-pipelined path (nsub = 1),
    two solves through one persistent ShardedBatch: per-problem equality with the single-process solve"""
    import torch.multiprocessing as mp
    tmp = str(tmp_path / "out.npz")
    port = 29500 + (os.getpid() % 2000) + 7 * world + nprob
    mp.spawn(_worker, args=(world, port, tmp, nprob, nsub), nprocs=world, join=True)
    got = np.load(tmp)
    for rep in range(2):
        probs = make_batch(nprob, 16, 40, seed0=3 + 100 * rep)
        P, q, Gt, h = pack_problems(probs)
        ref = coneqp_batch(P, q, Gt, h, kkt=NumpyBatchKkt(Gt, P))
        assert np.array_equal(got['it%d' % rep], ref['iterations'])
        assert np.allclose(got['x%d' % rep], ref['x'], rtol=0, atol=1e-12)
        assert np.allclose(got['z%d' % rep], ref['z'], rtol=0, atol=1e-12)
        assert np.allclose(got['pobj%d' % rep], ref['primal objective'], rtol=1e-13)


def _worker_device_branch(rank, world, port, tmp, nprob, nsub):
    """the DEVICE-RESIDENT branch of ShardedBatch (what runs over RCCL with BatchKkt engines) over gloo with host tensors:
    tensor inputs on the root, persistent engines per sub-batch, tensors back (return_device=True), two solves"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist
    from cvxopt_amd import batch
    from batch_helpers import TensorEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sb = batch.ShardedBatch(nprob, 16, 40, True, nsub=nsub, engine_factory=TensorEngine)
    assert sb.on_device
    out, held = {}, None
    for rep in range(2):
        if rank == 0:
            P, q, Gt, h = pack_problems(make_batch(nprob, 16, 40, seed0=3 + 100 * rep))
            args = [torch.from_numpy(a) for a in (P, q, Gt, h)]          # "already resident": tensors, as on the RCCL path
        else:
            args = [None] * 4
        res = sb.solve(*args, return_device=True, use_correction=(rep == 0))
        assert hasattr(res['x'], "numpy")                                  # tensors come back
        if rep == 0:
            held = (res['x'], res['x'].clone())
        if rank == 0:
            out.update({"x%d" % rep: res['x'].numpy(), "it%d" % rep: res['iterations'], "z%d" % rep: res['z'].numpy()})
        else:
            lo, hi = batch.shard_bounds(nprob, world)[rank]
            assert tuple(res['x'].shape) == (hi - lo, 16)
    # the first result must not have been overwritten by the second solve (non-root ranks used to hand out views of sb.pack)
    assert torch.equal(held[0], held[1])
    # engines: one per non-empty sub-batch of this rank, created once and reused by the second solve
    creates = [e for e in TensorEngine.log if e[0] == "create"]
    solves = [e for e in TensorEngine.log if e[0] == "coneqp"]
    nonempty = sum(1 for k in range(sb.nsub) if sb._rows(k)[1] > sb._rows(k)[0])
    assert len(creates) == nonempty and len(solves) == 2 * nonempty
    assert [e[0] for e in TensorEngine.log[:3]] == ["create", "set_problem", "coneqp"][:len(TensorEngine.log)]
    sb.close()
    if rank == 0:
        np.savez(tmp, **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nprob,nsub", [(2, 5, 4), (3, 7, 2), (4, 3, 2)])
def test_sharded_batch_device_resident_branch_on_gloo(tmp_path, world, nprob, nsub):
    """the first multi-GPU run must not be the first execution of the RCCL branch's control flow"""
    import torch.multiprocessing as mp
    tmp = str(tmp_path / "out.npz")
    port = 31500 + (os.getpid() % 2000) + 7 * world + nprob
    mp.spawn(_worker_device_branch, args=(world, port, tmp, nprob, nsub), nprocs=world, join=True)
    got = np.load(tmp)
    for rep in range(2):
        P, q, Gt, h = pack_problems(make_batch(nprob, 16, 40, seed0=3 + 100 * rep))
        ref = coneqp_batch(P, q, Gt, h, kkt=NumpyBatchKkt(Gt, P), use_correction=(rep == 0))
        assert np.array_equal(got['it%d' % rep], ref['iterations'])
        assert np.allclose(got['x%d' % rep], ref['x'], rtol=0, atol=1e-12)
        assert np.allclose(got['z%d' % rep], ref['z'], rtol=0, atol=1e-12)


def test_sharded_cache_key_survives_fresh_lambdas_and_is_cleared(tmp_path):
    """a new lambda per call must not rebuild the ShardedBatch; clear_sharded_cache() closes it"""
    import torch.multiprocessing as mp
    mp.spawn(_worker_cache, args=(29400 + os.getpid() % 2000,), nprocs=1, join=True)


def _worker_cache(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from cvxopt_amd import batch
    dist.init_process_group("gloo", rank=0, world_size=1)
    P, q, Gt, h = pack_problems(make_batch(3, 8, 20, seed0=1))
    seen = []
    for _ in range(3):
        batch.coneqp_batch_sharded(P, q, Gt, h, local_solver=lambda *a, **k: numpy_local_solver(*a, **k), nsub=2)
        seen.append(id(next(iter(batch._SHARDED_CACHE.values()))))
    assert len(set(seen)) == 1 and len(batch._SHARDED_CACHE) == 1
    batch.clear_sharded_cache()
    assert not batch._SHARDED_CACHE
    dist.destroy_process_group()


def test_ipc_transport_needs_a_gpu_and_says_so(tmp_path):
    """round 6: ShardedBatch(transport="ipc") without a visible HIP device fails loudly (no CPU fallback); an unknown transport is a
    ValueError"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the ipc transport is exercised by tests/test_gpu_round6.py")
    import torch.multiprocessing as mp
    mp.spawn(_worker_ipc_refused, args=(31400 + os.getpid() % 2000,), nprocs=1, join=True)


def _worker_ipc_refused(rank, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    import torch.distributed as dist
    from cvxopt_amd import batch
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        with pytest.raises(RuntimeError, match="no HIP device is visible"):
            batch.ShardedBatch(4, 8, 20, True, transport="ipc")
        with pytest.raises(ValueError):
            batch.ShardedBatch(4, 8, 20, True, transport="carrier pigeon")
    finally:
        dist.destroy_process_group()

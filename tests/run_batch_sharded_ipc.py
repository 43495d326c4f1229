"""Launched by tests/test_gpu_round6.py through torch.distributed.run: the sharded batch with transport="ipc" -- no data-path
collective: the root exports its buffers (hipIpcGetMemHandle), every rank pulls its shard and pushes its results with
device-to-device copies.  RANKS SHARE DEVICES when there are fewer GPUs than ranks (IPC_RANKS_PER_GPU: the round's box has one
GPU, so 2..4 processes all sit on it -- RCCL refuses that, IPC does not); the group is gloo and only carries the handles and
the closing barrier.

Uneven shards (3 world + 1 problems), an EMPTY shard (world + ... see `few`), several sub-batch counts, device-resident and host
inputs on the root, two solves per persistent object (the second maps nothing new): every problem of every run equals the
single-GPU device-resident solve of the same batch on the root -- iterations exactly, x / s / z to rounding."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.distributed as dist

from cvxopt_amd import _capi, synth
from cvxopt_amd.batch import BatchKkt, ShardedBatch, coneqp_batch_sharded, pack_problems

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ngpu = _capi.device_count()
local = rank % ngpu
torch.cuda.set_device(local)
dist.init_process_group("gloo")
n, m = 32, 70


def problems(nprob, seed):
    probs = [synth.dense_qp(n, m, seed=seed + i) for i in range(nprob)]
    return pack_problems(probs)


def single_gpu(P, q, Gt, h):
    kk = BatchKkt(Gt, P, device=local)
    out = kk.coneqp(q, h)
    kk.close()
    return out


def check(res, single, what):
    if rank != 0:
        return
    assert np.array_equal(res['iterations'], single['iterations']), what
    for k in ('x', 's', 'z'):
        got = res[k].cpu().numpy() if hasattr(res[k], "cpu") else res[k]
        assert np.allclose(got, single[k], rtol=1e-12, atol=1e-13), (what, k)      # same kernels, same data: rounding only
    assert np.allclose(res['primal objective'], single['primal objective'], rtol=1e-13), what
    assert all(s == 'optimal' for s in res['status']), what


tm = {}
for nprob, seed in ((3 * world + 1, 40), (max(1, world - 1), 90)):          # uneven shards; fewer problems than ranks: empty shards
    P = q = Gt = h = single = None
    if rank == 0:
        P, q, Gt, h = problems(nprob, seed)
        single = single_gpu(P, q, Gt, h)
    # 1. the convenience wrapper with host inputs on the root
    res = coneqp_batch_sharded(P, q, Gt, h, transport="ipc")
    check(res, single, "wrapper, host inputs, %d problems" % nprob)
    # 2. persistent objects, device-resident inputs on the root, several sub-batch counts, two solves each
    for nsub in (1, 2, 4):
        sb = ShardedBatch(nprob, n, m, True, nsub=nsub, transport="ipc")
        dev_in = [None] * 4
        if rank == 0:
            dev_in = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (P, q, Gt, h)]
        for rep in range(2):
            res = sb.solve(dev_in[0], dev_in[1], dev_in[2], dev_in[3], return_device=(rep == 1))
            check(res, single, "ShardedBatch ipc nsub=%d rep=%d, %d problems" % (nsub, rep, nprob))
            if rank != 0:                         # the other ranks hold their own shard's results
                lo, hi = sb.bounds[rank]
                assert len(res['iterations']) == hi - lo
        tm = sb.last_timings
        assert set(tm) >= {"scatter_exposed", "scatter_all", "upload", "solve", "gather_exposed", "total"}
        if rank != 0 and sb.nloc > 0:
            assert len(sb._ipc_open) >= 1         # mapped, not copied through the host
        sb.close()
if rank == 0:
    print("SHARDED_IPC_OK world=%d gpus=%d timings_ms=%s" % (world, ngpu, {k: round(v, 2) for k, v in tm.items()}))
dist.barrier()
from cvxopt_amd.batch import clear_sharded_cache
clear_sharded_cache()
dist.destroy_process_group()

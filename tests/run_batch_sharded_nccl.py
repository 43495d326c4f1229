"""Launched by tests/test_gpu_batch.py through torch.distributed.run: the sharded batch path on the real
RCCL backend (one rank per visible GPU; the round's GPU box has one, the driver's scaling node eight).

For ANY world size: unequal shards (3 world + 1 problems), the pipelined scatter / solve / gather of `ShardedBatch` with several
sub-batch counts, device-resident inputs (CUDA tensors on the root) and host inputs, two solves per persistent object -- every
problem of every run must equal the single-GPU device-resident solve of the same batch on the root (iterations exactly, x / s / z
to rounding: both run the same kernels) and the NumPy lock-step twin."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.distributed as dist

from cvxopt_amd import synth
from cvxopt_amd.batch import BatchKkt, ShardedBatch, coneqp_batch, coneqp_batch_sharded, pack_problems

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
nprob, n, m = 3 * world + 1, 32, 70
P = q = Gt = h = None
single = twin = None
if rank == 0:
    probs = [synth.dense_qp(n, m, seed=40 + i) for i in range(nprob)]
    P, q, Gt, h = pack_problems(probs)
    kk = BatchKkt(Gt, P, device=local)
    single = kk.coneqp(q, h)                                  # the single-GPU point of the same curve
    kk.close()
    twin = coneqp_batch(P, q, Gt, h, device=local)            # NumPy bookkeeping around the batched factor / solve


def check(res, what):
    if rank != 0:
        return
    assert np.array_equal(res['iterations'], single['iterations']), what
    for k in ('x', 's', 'z'):
        got = res[k].cpu().numpy() if hasattr(res[k], "cpu") else res[k]
        assert np.allclose(got, single[k], rtol=1e-12, atol=1e-13), (what, k)      # same kernels, same data: rounding only
    assert np.allclose(res['primal objective'], single['primal objective'], rtol=1e-13), what
    assert np.array_equal(res['iterations'], twin['iterations']), what
    assert np.allclose(res['x'].cpu().numpy() if hasattr(res['x'], "cpu") else res['x'], twin['x'], rtol=1e-9, atol=1e-11), what
    assert all(s == 'optimal' for s in res['status']), what


# 1. the convenience wrapper with host inputs (cached ShardedBatch, default sub-batch count)
res = coneqp_batch_sharded(P, q, Gt, h)
check(res, "wrapper, host inputs")
# 2. persistent objects with device-resident inputs on the root, several sub-batch counts, two solves each
for nsub in (1, 2, 4):
    sb = ShardedBatch(nprob, n, m, True, nsub=nsub)
    dev_in = [None] * 4
    if rank == 0:
        dev_in = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (P, q, Gt, h)]
    for rep in range(2):
        res = sb.solve(dev_in[0], dev_in[1], dev_in[2], dev_in[3], return_device=(rep == 1))
        check(res, "ShardedBatch nsub=%d rep=%d" % (nsub, rep))
    tm = sb.last_timings
    assert set(tm) >= {"scatter_exposed", "scatter_all", "upload", "solve", "gather_exposed", "total"}
    sb.close()
if rank == 0:
    print("SHARDED_NCCL_OK world=%d problems=%d iterations=%s timings_ms=%s" % (
        world, nprob, single['iterations'].tolist(), {k: round(v, 2) for k, v in tm.items()}))
dist.barrier()
dist.destroy_process_group()

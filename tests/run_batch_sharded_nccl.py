"""Launched by tests/test_gpu_batch.py through torch.distributed.run: the sharded batch path on the real
RCCL backend (one rank per visible GPU; the round's GPU box has one)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.distributed as dist

from cvxopt_amd import synth
from cvxopt_amd.batch import coneqp_batch, coneqp_batch_sharded, pack_problems

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
P = q = Gt = h = None
if rank == 0:
    probs = [synth.dense_qp(32, 70, seed=40 + i) for i in range(2 * world + 1)]
    P, q, Gt, h = pack_problems(probs)
res = coneqp_batch_sharded(P, q, Gt, h)
if rank == 0:
    ref = coneqp_batch(P, q, Gt, h, device=local)
    assert np.array_equal(res['iterations'], ref['iterations'])
    assert np.allclose(res['x'], ref['x'], rtol=1e-9, atol=1e-11)      # resident loop vs NumPy loop: rounding only
    assert all(s == 'optimal' for s in res['status'])
    print("SHARDED_NCCL_OK world=%d problems=%d iterations=%s" % (world, len(res['iterations']), res['iterations'].tolist()))
dist.barrier()
dist.destroy_process_group()

#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: "KKT factor+solve ms/iter (and IPM iters/sec), dense QP n=8192".

A "step" is one pass of the hot path = what one coneqp IPM iteration asks of the kktsolver hook on a
dense LP-cone QP: 1 factor(W, P) + 2 solve(x, y, z)   (reference coneprog.py:2256, :2360; refinement 0).
Workload at N=1: BASELINE configs[1] (n=8192, m=16384, p=0).  Inputs (G, P, the scaling di, the
right-hand sides) are resident in HBM when the timed region starts; only the 4-byte `info` word crosses
PCIe per factor.  N>1: one process per GPU, every rank runs its own replica of the workload
("replicas only": a single factorisation does not shard; independent problems do), weak scaling,
value = KKT iterations/s summed over ranks.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: the FP64-MFMA scaled SYRK, timed live with
HIP events on the solver's stream) and `cpu_baseline` (the real reference kkt_chol2 + MKL from oracle/_ref,
timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz (v_mfma_f64_16x16x4_f64: 64 cyc) = AMD spec
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--m", type=int, default=16384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="dense", choices=["dense", "batch", "sparse", "socp"],
                    help="dense = BASELINE configs[1] (the headline line, default; --n 256 --m 512 gives configs[0]); "
                         "batch = configs[4] class (independent n=512 problems, sharded over ranks); sparse = configs[3] "
                         "class (3-D Laplacian box-QP); socp = configs[2] (n=2048, 1024 second-order cones of dimension 8)")
    ap.add_argument("--batch", type=int, default=512, help="problems per GPU for --workload batch")
    ap.add_argument("--grid", type=int, default=46, help="k for the k^3 Laplacian of --workload sparse")
    ap.add_argument("--cone-dim", type=int, default=8, help="--workload socp: dimension of each of the 1024 second-order cones")
    ap.add_argument("--mesh", default="grid", choices=["grid", "tet"],
                    help="--workload sparse: structured 7-point grid (default) or an unstructured tetrahedral mesh of k^3 nodes")
    ap.add_argument("--cpu-iters", type=int, default=2, help="reference CPU iterations timed (bounded sample)")
    return ap.parse_args()


def cpu_baseline(pr, W_np, n, m, iters):
    """Reference misc.kkt_chol2 (MKL) on the host: iters x (factor + 2 solves), same inputs."""
    import numpy as np
    try:
        from oracle import refloader
        cvx = refloader.load()
        from cvxopt import matrix, spmatrix, misc
        kind = "reference"
    except Exception:
        cvx = None
        kind = "port"
    rng = np.random.default_rng(1)
    ts = []
    if kind == "reference":
        P, G = matrix(pr['P']), matrix(pr['G'])
        A = spmatrix([], [], [], (0, n))
        W = {'d': matrix(W_np['d']), 'di': matrix(W_np['di']), 'v': [], 'beta': [], 'r': [], 'rti': []}
        factor = misc.kkt_chol2(G, pr['dims'], A)
        for it in range(iters + 1):                 # first call allocates; not timed (reference firstcall branch)
            t0 = time.perf_counter()
            solve = factor(W, P)
            for _ in range(2):
                x, y, z = matrix(rng.standard_normal(n)), matrix(0.0, (0, 1)), matrix(rng.standard_normal(m))
                solve(x, y, z)
            ts.append(time.perf_counter() - t0)
        ts = ts[1:]
    else:
        from oracle import kkt_oracle as ko
        o = ko.KktChol2(pr['G'], pr['dims'], np.zeros((0, n)))
        for it in range(iters):
            t0 = time.perf_counter()
            solve = o.factor(W_np, pr['P'])
            for _ in range(2):
                solve(rng.standard_normal(n), np.zeros(0), rng.standard_normal(m))
            ts.append(time.perf_counter() - t0)
    try:
        import ctypes
        mkl = ctypes.CDLL("/opt/conda/lib/libmkl_rt.so")
        cores = int(mkl.MKL_Get_Max_Threads())
    except Exception:
        cores = os.cpu_count() or 1
    ms = 1e3 * sum(ts) / len(ts)
    return {"value": round(ms, 2), "unit": "ms/iter (factor + 2 solves)", "cores": cores, "kind": kind,
            "sample": "%d KKT iterations of the same n=%d, m=%d workload, kktsolver='chol2', same W" % (len(ts), n, m),
            "iters_per_s": round(1e3 / ms, 4)}


def _dist_setup():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    return rank, world, local_rank, torch, dist


def _timed(step, args, torch, dist):
    from cvxopt_amd import _capi

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        _capi.lib().mi355kkt_device_synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def main_batch(args):
    """configs[4] class: every rank owns `--batch` independent dense QPs (n=512, m=1024); a step = the whole
    device-resident coneqp solve of the rank's shard (no collective in the data path); value = problem-IPM-iterations/s."""
    rank, world, local_rank, torch, dist = _dist_setup()
    import numpy as np
    from cvxopt_amd import synth
    from cvxopt_amd.batch import BatchKkt, pack_problems
    B, n, m = args.batch, 512, 1024
    probs = [synth.dense_qp(n, m, seed=rank * B + i) for i in range(B)]
    P, q, Gt, h = pack_problems(probs)
    k = BatchKkt(Gt, P, device=local_rank)
    res = {}

    def step():
        res['r'] = k.coneqp(q, h)
    elapsed = _timed(step, args, torch, dist)
    its = int(res['r']['iterations'].sum())
    if dist is not None:
        t = torch.tensor([its], dtype=torch.float64, device="cuda")
        dist.all_reduce(t)
        its = int(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "batched coneqp: problem-IPM-iterations/s (BASELINE configs[4] class)", "value": round(its * args.steps / elapsed, 1),
            "unit": "problem-iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d independent dense QPs per GPU, n=%d, m=%d, whole coneqp solve resident on the device"
                                   % (B, n, m), "all_optimal": bool(np.all(res['r']['status'] == 'optimal'))}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_sparse(args):
    """configs[3] class stand-in: P = 3-D 7-point Laplacian + 1e-2 I, box constraints; a step = 1 sparse
    factor + 2 solves through the hook-level engine with inputs resident; replicas over ranks."""
    rank, world, local_rank, torch, dist = _dist_setup()
    import numpy as np
    import scipy.sparse as sp
    from cvxopt_amd import kkt, synth, _capi
    kkt.options["device"] = local_rank
    k = args.grid
    if args.mesh == "tet":               # unstructured stand-in: Delaunay tetrahedralisation of k^3 random points
        P = synth.tet_mesh_laplacian(k ** 3, seed=0)
        what = "graph Laplacian of a random tetrahedral mesh with %d nodes" % (k ** 3)
    else:
        P = synth.grid_laplacian(k)
        what = "%d^3 7-point Laplacian" % k
    n = k ** 3
    G = sp.vstack([sp.eye(n), -sp.eye(n)]).tocsc()

    class Sp(object):                      # the minimal spmatrix surface cvxopt_amd reads (.size, .CCS)
        def __init__(self, A):
            A = sp.csc_matrix(A)
            A.sort_indices()
            self.size = A.shape
            self.CCS = (A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64))
    dims = {'l': 2 * n, 'q': [], 's': []}
    t = time.perf_counter()
    f = kkt.kkt_chol2(Sp(G), dims, np.zeros((0, n)))
    W = synth.random_scaling(dims, seed=rank, spread=1.0)
    f(W, Sp(sp.tril(P)))
    t_sym = time.perf_counter() - t
    eng = f.engine
    d_di = _capi.DeviceBuffer.from_array(W['di'])
    rng = np.random.default_rng(rank)
    rhs = [(_capi.DeviceBuffer.from_array(rng.standard_normal(n)), _capi.DeviceBuffer.from_array(rng.standard_normal(2 * n)))
           for _ in range(2)]
    d_y = _capi.DeviceBuffer(8)

    def step():
        eng.factor_device(di_ptr=d_di.ptr)
        for dx, dz in rhs:
            eng.solve_device(dx.ptr, d_y.ptr, dz.ptr)
        eng.sync()
    elapsed = _timed(step, args, torch, dist)
    if rank == 0:
        st = eng.sparse_stats()
        print(json.dumps({
            "metric": "sparse KKT factor+solve ms/iter (BASELINE configs[3] class)", "value": round(world * args.steps / elapsed, 3),
            "unit": "KKT iterations/s (1 factor + 2 solves each)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "box-QP on the %s (n=%d, m=%d), supernodal multifrontal engine; stand-in for "
                                   "the ssget-1288 class (no network)" % (what, n, 2 * n), "replicas": world,
                       "ordering": {1: "nested dissection", 2: "approximate minimum degree"}.get(st.get("ordering"), "?"),
                       "nnzL": st["nnzL"], "supernodes": st["supernodes"], "levels": st["levels"], "flops_estimate": st["flops"],
                       "symbolic_plus_first_factor_s": round(t_sym, 3)}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_socp(args):
    """configs[2]: SOCP, n=2048, 1024 cones of dimension 8 (cdim 8192); a step = what one conelp iteration asks of the
    kktsolver hook with second-order cones: 1 factor(W) + 5 solves (refinement 1: (x1,y1,z1) + 2 right-hand sides x 2)."""
    rank, world, local_rank, torch, dist = _dist_setup()
    import numpy as np
    from cvxopt_amd import kkt, synth, _capi
    import cvxopt_amd
    kkt.options["device"] = local_rank
    n, ncones, r = 2048, 1024, args.cone_dim      # dimension 8 is the survey's choice; 4 and 64 are the side cases
    pr = synth.socp(n=n, ncones=ncones, r=r, seed=rank)
    cdim = ncones * r
    W = synth.random_scaling(pr['dims'], seed=100 + rank, spread=1.0)
    f = kkt.kkt_chol(pr['G'], pr['dims'], np.zeros((0, n)))
    eng = f.engine
    f(W)                                                   # first call: uploads, allocations
    rng = np.random.default_rng(rank)
    v = np.concatenate([np.asarray(vk, dtype=float).ravel() for vk in W['v']])
    d_v = _capi.DeviceBuffer.from_array(v)
    d_beta = _capi.DeviceBuffer.from_array(np.array([float(b) for b in W['beta']]))
    rhs = [(_capi.DeviceBuffer.from_array(rng.standard_normal(n)), _capi.DeviceBuffer.from_array(rng.standard_normal(cdim)))
           for _ in range(5)]
    d_y = _capi.DeviceBuffer(8)

    def step():
        eng.factor_device(v_ptr=d_v.ptr, beta_ptr=d_beta.ptr)
        for dx, dz in rhs:
            eng.solve_device(dx.ptr, d_y.ptr, dz.ptr)
        eng.sync()
    elapsed = _timed(step, args, torch, dist)
    tm = eng.timings()
    e2e = None
    if rank == 0:
        for _ in range(2):
            t1 = time.perf_counter()
            sol = cvxopt_amd.conelp_device(pr['c'], pr['G'], pr['h'], pr['dims'])
            t1 = time.perf_counter() - t1
        e2e = {"solver": "device-resident conelp loop (mi355kkt_conelp)", "status": sol['status'], "iterations": sol['iterations'],
               "seconds": round(t1, 4)}
        print(json.dumps({
            "metric": "SOCP KKT factor+solve ms/iter (BASELINE configs[2])", "value": round(world * args.steps / elapsed, 3),
            "unit": "KKT iterations/s (1 factor + 5 solves each)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "conelp SOCP n=%d, %d second-order cones of dimension %d (cdim %d); hook = 1 factor(W) + 5 "
                                   "solve(x,y,z) per step, inputs resident in HBM" % (n, ncones, r, cdim), "replicas": world},
            "phases_ms": {k: round(v, 3) for k, v in tm.items()}, "ipm_end_to_end": e2e}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    if args.workload == "socp":
        return main_socp(args)
    if args.workload == "batch":
        return main_batch(args)
    if args.workload == "sparse":
        return main_sparse(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch                                   # plumbing only: device selection + torch.distributed
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):   # launched by torch.distributed.run
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    from cvxopt_amd import kkt, synth, _capi
    kkt.options["device"] = local_rank
    n, m = args.n, args.m

    # ---- synthetic inputs, seeded per rank; uploaded once ---------------------------
    t_gen = time.perf_counter()
    pr = synth.dense_qp(n, m, seed=rank)
    W_np = synth.random_scaling(pr['dims'], seed=100 + rank, spread=1.0)
    rng = np.random.default_rng(7 + rank)
    factor = kkt.kkt_chol2(pr['G'], pr['dims'], np.zeros((0, n)))
    eng = factor.engine
    dP = _capi.DeviceBuffer.from_array(pr['P'])
    eng.set_H_device(dP.ptr, n)
    d_di = _capi.DeviceBuffer.from_array(W_np['di'])
    rhs = [( _capi.DeviceBuffer.from_array(rng.standard_normal(n)), _capi.DeviceBuffer.from_array(rng.standard_normal(m)))
           for _ in range(2)]
    d_y = _capi.DeviceBuffer(8)
    t_gen = time.perf_counter() - t_gen

    syrk_ms = []

    def step():
        eng.factor_device(di_ptr=d_di.ptr)
        for dx, dz in rhs:
            eng.solve_device(dx.ptr, d_y.ptr, dz.ptr)
        eng.sync()
        syrk_ms.append(eng.timings()["syrk_kernel_ms"])

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        _capi.lib().mi355kkt_device_synchronize()

    for _ in range(args.warmup):
        step()
    del syrk_ms[:]
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tm = eng.timings()

    # ---- outside the timed region: the "(and IPM iters/sec)" half of the metric -- the whole coneqp solve of the same
    # problem with the interior-point loop resident on the device (mi355kkt_coneqp_lp), G and P already in HBM
    ipm = None
    if rank == 0:
        try:
            for _ in range(2):                         # first run allocates the loop's workspaces
                t1 = time.perf_counter()
                sol = eng.coneqp(pr['q'], pr['h'], keep_H=True)
                t1 = time.perf_counter() - t1
            ipm = {"solver": "device-resident coneqp loop (mi355kkt_coneqp_lp)", "status": sol['status'],
                   "iterations": sol['iterations'], "seconds": round(t1, 4),
                   "iters_per_s": round(sol['iterations'] / t1, 3),
                   "primal_objective": sol['primal objective']}
        except Exception as e:
            ipm = {"error": repr(e)}

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        kernel_ms = sum(syrk_ms) / max(1, len(syrk_ms))
        flops = float(m) * n * n                      # algorithmic flops of the lower-triangular SYRK
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        traffic = None
        pj = os.path.join(ROOT, "profiles", "pmc_syrk_latest.json")
        if os.path.exists(pj):
            try:
                d = json.load(open(pj))
                if d.get("n") == n and d.get("m") == m:
                    traffic = d.get("hbm_bytes_per_launch")
            except Exception:
                pass
        out = {
            "metric": "KKT factor+solve ms/iter (and IPM iters/sec), dense QP n=%d" % n,
            "value": round(world * args.steps / elapsed, 4),
            "unit": "KKT iterations/s (1 factor + 2 solves each; ms/iter in ms_per_step)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]: coneqp dense QP n=%d, m=%d, p=0, LP cone; kktsolver hook "
                                   "= 1 factor(W,P) + 2 solve(x,y,z) per step, inputs resident in HBM"
                                   % (0 if (n, m) == (256, 512) else 1, n, m),
                       "replicas": world, "formulation": "reduced S = P + G'D^2G, Cholesky (kkt_chol2/ldl engine)"},
            "phases_ms": {k: round(v, 3) for k, v in tm.items()},
            "ipm_end_to_end": ipm,
            "roofline": {"kernel": "syrk_tn_kernel (S = P + G' diag(di)^2 G, FP64 MFMA)", "bound": "mfma",
                         "achieved": round(achieved, 2), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "kernel_ms": round(kernel_ms, 3), "flops_per_launch": flops},
        }
        if not args.no_cpu_baseline and world == 1:          # reported at N=1 only (rank 0, host cores)
            try:
                out["cpu_baseline"] = cpu_baseline(pr, W_np, n, m, args.cpu_iters)
                out["speedup_vs_cpu"] = round(out["cpu_baseline"]["value"] / ms_per_step, 2)
            except Exception as e:                     # the baseline must never take the GPU number down
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

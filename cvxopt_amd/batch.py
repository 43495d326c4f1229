"""Batched mode: many independent dense cone QPs of one shape (BASELINE configs[4]: the LP cone).

    min 1/2 x'P_b x + q_b'x   s.t.  G_b x <= h_b  [A_b x = b_b]          b = 0 .. B-1
    ('<=' in the cone of dims = {'l': .., 'q': [..]}; `BatchKkt(..., At=, dims=)`; the NumPy twin below is LP-cone only)

There is no reference API for a batch; parity is per problem against individual
`solvers.coneqp` calls.  The KKT work (the hot path: S_b = P_b + G_b'D_b^2 G_b, Cholesky, two solves per
iteration for every problem) runs as batched HIP kernels behind `mi355kkt_batch_*`; the O(B*m) cone-vector
bookkeeping of the interior-point loop is restated here in lock-step NumPy, operation for operation
after reference src/python/coneprog.py:2044-2547 ('l' cone, p = 0, refinement 0, Mehrotra correction on)
and src/python/misc.py:284-287, :444-464 (compute_scaling / update_scaling, 'l' blocks).

Array convention: everything is packed problem-major, `Gt[b]` is G_b' (n x ml, C order) so that the raw
buffer is G_b column-major -- the layout the C ABI wants.  `pack_problems` builds it from ordinary arrays.

Multi-GPU: `coneqp_batch_sharded` partitions the batch into contiguous shards, one per rank of a
torch.distributed group (RCCL on GPUs; gloo in the CPU tests), scatters the problem data from the root,
solves the local shard and gathers the results -- there is no collective inside the IPM.
"""
import ctypes as C

import numpy as np

from . import _capi


# ---------------------------------------------------------------------------------------------------
# KKT back end: batched factor / solve on the GPU
# ---------------------------------------------------------------------------------------------------
class BatchKkt(object):
    """B independent kkt_chol2-style solvers (LP cone; optionally p equality constraints per problem, At: (B, n, p) = the
    p x n blocks A_b column-major) behind the batched C ABI.  dims = {'l': nl, 'q': [...]} (the same for every problem) adds
    second-order cones: Gt is then (B, n, cdim), cdim = nl + sum(q), and the engine works like the reference's kkt_chol
    per problem (Gs_b = W_b^-T G_b materialised)."""

    def __init__(self, Gt=None, P=None, device=0, At=None, dims=None, shape=None, p=None):
        """Gt (B, n, m) [+ P (B, n, n), At (B, n, p)]: create the handle and upload the problems.  Gt = None with shape = (B, n, m)
        [and p]: create the handle only -- a persistent engine whose problem data arrives later through `set_problem` (the
        sharded batch: one handle per rank for the lifetime of the job, the ~GB allocations happen once)."""
        self.L = _capi.lib()
        if _capi.device_count() <= 0:
            raise RuntimeError("cvxopt_amd.batch: no HIP device visible (there is no CPU fallback)")
        if Gt is None:
            if shape is None:
                raise TypeError("BatchKkt: Gt or shape = (B, n, m) is required")
            self.B, self.n, self.m = (int(v) for v in shape)
        elif hasattr(Gt, "data_ptr"):
            device = Gt.device.index if Gt.device.index is not None else device
            self.B, self.n, self.m = (int(v) for v in Gt.shape)
        else:
            Gt = np.ascontiguousarray(Gt, dtype=np.float64)
            self.B, self.n, self.m = Gt.shape
        self.device = device
        h = C.c_void_p()
        self.p = int(p) if p is not None else (0 if At is None else int(At.shape[2]))
        self.q = [int(k) for k in (dims or {}).get('q', [])]
        self.nl = int(dims['l']) if dims is not None else self.m
        if dims is not None:
            if dims.get('s'):
                raise NotImplementedError("batched mode: 's' blocks are not supported")
            if self.nl + sum(self.q) != self.m:
                raise TypeError("Gt must have shape (B, n, %d) for these dims" % (self.nl + sum(self.q)))
        if self.q:
            qa = (C.c_int * len(self.q))(*self.q)
            _capi.check(self.L.mi355kkt_batch_create_cones(C.byref(h), device, self.B, self.n, self.nl, len(self.q), qa, self.p),
                        "batch_create_cones")
        else:
            _capi.check(self.L.mi355kkt_batch_create_eq(C.byref(h), device, self.B, self.n, self.m, self.p), "batch_create")
        self.h = h
        if Gt is not None:
            self.set_problem(Gt, P, At)

    def set_problem(self, Gt, P=None, At=None):
        """(Re)load the problem data of this handle: NumPy arrays, or contiguous float64 CUDA tensors already in this GPU's HBM
        (copied device to device into the handle's own buffers).  Same shapes as at creation."""
        h = self.h
        on_device = hasattr(Gt, "data_ptr")                       # torch CUDA tensors: the data is already in HBM
        if on_device:
            if not (Gt.is_cuda and Gt.is_contiguous() and Gt.dtype.is_floating_point and Gt.element_size() == 8):
                raise TypeError("Gt must be a contiguous float64 CUDA tensor")
            if tuple(int(v) for v in Gt.shape) != (self.B, self.n, self.m):
                raise TypeError("Gt must have shape %r" % ((self.B, self.n, self.m),))
            gptr = Gt.data_ptr()
        else:
            Gt = np.ascontiguousarray(Gt, dtype=np.float64)
            if Gt.shape != (self.B, self.n, self.m):
                raise TypeError("Gt must have shape %r" % ((self.B, self.n, self.m),))
            gptr = Gt.ctypes.data
        Pp = None
        if P is not None:
            if on_device:
                if not (hasattr(P, "data_ptr") and P.is_cuda and P.is_contiguous() and P.element_size() == 8):
                    raise TypeError("P must be a contiguous float64 CUDA tensor when Gt is one")
                assert tuple(P.shape) == (self.B, self.n, self.n)
                Pp = P.data_ptr()
            else:
                P = np.ascontiguousarray(P, dtype=np.float64)        # symmetric: C order == column-major
                assert P.shape == (self.B, self.n, self.n)
                Pp = P.ctypes.data
        if on_device:
            import torch
            # the copies below run on the library's own stream: the producer of the tensors (the current torch stream -- e.g. the
            # wait on a scatter) must be done, NOT the whole device (later collectives of a pipelined scatter keep flying)
            torch.cuda.current_stream(Gt.device).synchronize()
        _capi.check(self.L.mi355kkt_batch_set_problem(h, C.c_void_p(gptr), C.c_void_p(Pp) if Pp else None,
                                                      1 if on_device else 0), "batch_set_problem")
        if self.p:
            if At is None:
                raise TypeError("At is required for a handle with p > 0")
            if on_device:
                if not (hasattr(At, "data_ptr") and At.is_cuda and At.is_contiguous() and At.element_size() == 8):
                    raise TypeError("At must be a contiguous float64 CUDA tensor when Gt is one")
                assert tuple(At.shape) == (self.B, self.n, self.p)
                aptr = At.data_ptr()
            else:
                At = np.ascontiguousarray(At, dtype=np.float64)
                assert At.shape == (self.B, self.n, self.p)
                aptr = At.ctypes.data
            _capi.check(self.L.mi355kkt_batch_set_A(h, C.c_void_p(aptr), 1 if on_device else 0), "batch_set_A")

    def factor(self, di):
        di = np.ascontiguousarray(di, dtype=np.float64)
        info = np.zeros(self.B, dtype=np.int32)
        _capi.check(self.L.mi355kkt_batch_factor(self.h, di.ctypes.data, 0, info.ctypes.data_as(_capi.c_int_p)),
                    "batch_factor")
        return info

    def factor_cones(self, di, v, beta):
        """Batches with second-order cones: di (B, nl) = W['di'], v (B, sum(q)) = the cones' W['v'][k] back to back, beta (B, nq)
        = W['beta'] of every problem."""
        B, m = self.B, self.m
        dfull = np.zeros((B, m))
        if self.nl:
            dfull[:, :self.nl] = np.asarray(di, dtype=np.float64).reshape(B, self.nl)
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(B, sum(self.q))
        beta = np.ascontiguousarray(beta, dtype=np.float64).reshape(B, len(self.q))
        info = np.zeros(B, dtype=np.int32)
        _capi.check(self.L.mi355kkt_batch_factor_cones(self.h, dfull.ctypes.data, v.ctypes.data, beta.ctypes.data, 0,
                                                       info.ctypes.data_as(_capi.c_int_p)), "batch_factor_cones")
        return info

    def solve(self, x, z, y=None):
        assert x.flags.c_contiguous and z.flags.c_contiguous and x.dtype == np.float64 and z.dtype == np.float64
        if self.p:
            assert y is not None and y.flags.c_contiguous and y.dtype == np.float64 and y.shape == (self.B, self.p)
            _capi.check(self.L.mi355kkt_batch_solve_eq(self.h, x.ctypes.data, y.ctypes.data, z.ctypes.data, 0), "batch_solve")
        else:
            _capi.check(self.L.mi355kkt_batch_solve(self.h, x.ctypes.data, z.ctypes.data, 0), "batch_solve")

    def products(self, x, z):
        """(G x, G' z, P x) for every problem, computed next to the data in HBM."""
        x = np.ascontiguousarray(x)
        z = np.ascontiguousarray(z)
        Gx, GTz, Px = np.empty((self.B, self.m)), np.empty((self.B, self.n)), np.empty((self.B, self.n))
        _capi.check(self.L.mi355kkt_batch_products(self.h, x.ctypes.data, z.ctypes.data, Gx.ctypes.data, GTz.ctypes.data,
                                                   Px.ctypes.data, 0), "batch_products")
        return Gx, GTz, Px

    def factor_ms(self):
        return float(self.L.mi355kkt_batch_last_factor_ms(self.h))

    def coneqp(self, q, h, maxiters=100, abstol=1e-7, reltol=1e-6, feastol=1e-7, b=None, use_correction=True):
        """The whole interior-point loop on the device (`mi355kkt_batch_coneqp`): same result dict as
        `coneqp_batch`; only the count of active problems crosses PCIe per iteration.  q, h may be float64 CUDA tensors
        (a sharded batch: they arrived over RCCL); the result arrays are then CUDA tensors too and nothing but the
        per-iteration word touches the host."""
        B, n, m = self.B, self.n, self.m
        on_device = hasattr(q, "data_ptr")
        ip = _capi.c_int_p
        nrun = C.c_int(0)
        _capi.check(self.L.mi355kkt_batch_set_option(self.h, b"use_correction", 1.0 if use_correction else 0.0),
                    "batch_set_option")                      # options['use_correction'] of solvers.coneqp (coneprog.py:1781)
        if on_device:
            import torch
            if not (q.is_cuda and h.is_cuda and q.dtype == torch.float64 and h.dtype == torch.float64):
                raise TypeError("q, h must be float64 CUDA tensors")
            q, h = q.contiguous(), h.contiguous()
            assert tuple(q.shape) == (B, n) and tuple(h.shape) == (B, m)
            dev = q.device
            x, s, z = (torch.zeros((B, k), dtype=torch.float64, device=dev) for k in (n, m, m))
            y = torch.zeros((B, max(1, self.p)), dtype=torch.float64, device=dev)
            status, iters = (torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2))
            pc, dc, gap = (torch.zeros(B, dtype=torch.float64, device=dev) for _ in range(3))
            if self.p:
                if not (hasattr(b, "data_ptr") and b.is_cuda and b.dtype == torch.float64 and tuple(b.shape) == (B, self.p)):
                    raise TypeError("b must be a float64 CUDA tensor of shape (B, p)")
                b = b.contiguous()
            torch.cuda.current_stream(dev).synchronize()          # the loop runs on the library's own stream (see set_problem)
            ptr = lambda t: C.c_void_p(t.data_ptr())
            rc = self.L.mi355kkt_batch_coneqp_eq(self.h, ptr(q), ptr(h), ptr(b) if self.p else None, int(maxiters), float(abstol),
                                                 float(reltol), float(feastol), ptr(x), ptr(y), ptr(s), ptr(z),
                                                 C.cast(ptr(status), ip), C.cast(ptr(iters), ip), ptr(pc), ptr(dc), ptr(gap),
                                                 C.byref(nrun))
        else:
            q = np.ascontiguousarray(q, dtype=np.float64)
            h = np.ascontiguousarray(h, dtype=np.float64)
            assert q.shape == (B, n) and h.shape == (B, m)
            x, s, z = np.zeros((B, n)), np.zeros((B, m)), np.zeros((B, m))
            y = np.zeros((B, max(1, self.p)))
            bv = None
            if self.p:
                bv = np.ascontiguousarray(b, dtype=np.float64)
                assert bv.shape == (B, self.p)
            status, iters = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
            pc, dc, gap = np.zeros(B), np.zeros(B), np.zeros(B)
            rc = self.L.mi355kkt_batch_coneqp_eq(self.h, q.ctypes.data, h.ctypes.data, bv.ctypes.data if self.p else None,
                                                 int(maxiters), float(abstol), float(reltol), float(feastol), x.ctypes.data,
                                                 y.ctypes.data, s.ctypes.data, z.ctypes.data, status.ctypes.data_as(ip),
                                                 iters.ctypes.data_as(ip), pc.ctypes.data, dc.ctypes.data, gap.ctypes.data,
                                                 C.byref(nrun))
        if rc == 1:
            raise ValueError("Rank([P; A; G]) < n (%s)" % _capi.last_error())
        _capi.check(rc, "batch_coneqp")
        y = y[:, :self.p]
        if on_device:        # status stays numeric on the device: 1 optimal, 2 / 3 unknown (see include/mi355kkt.h)
            return {'x': x, 'y': y, 's': s, 'z': z, 'status_code': status, 'iterations': iters, 'primal objective': pc,
                    'dual objective': dc, 'gap': gap, 'lockstep iterations': nrun.value}
        names = np.array(['unknown', 'optimal', 'unknown', 'unknown'], dtype=object)
        return {'x': x, 'y': y, 's': s, 'z': z, 'status': names[status], 'iterations': iters.astype(int),
                'primal objective': pc, 'dual objective': dc, 'gap': gap, 'lockstep iterations': nrun.value}

    def close(self):
        if getattr(self, "h", None):
            self.L.mi355kkt_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_problems(problems):
    """list of dicts with P (n x n), q, G (m x n), h  ->  packed arrays (P, q, Gt, h)."""
    P = np.stack([np.asarray(p['P'], dtype=float) for p in problems])
    q = np.stack([np.asarray(p['q'], dtype=float).ravel() for p in problems])
    Gt = np.stack([np.ascontiguousarray(np.asarray(p['G'], dtype=float).T) for p in problems])
    h = np.stack([np.asarray(p['h'], dtype=float).ravel() for p in problems])
    return P, q, Gt, h


# ---------------------------------------------------------------------------------------------------
# lock-step interior-point loop (host bookkeeping, device KKT)
# ---------------------------------------------------------------------------------------------------
def coneqp_batch(P, q, Gt, h, kkt=None, maxiters=100, abstol=1e-7, reltol=1e-6, feastol=1e-7, device=0,
                 resident=False, use_correction=True):
    """Solves the B problems in lock step.  Returns a dict of arrays: x (B,n), s, z (B,m), status (B,)
    ('optimal' | 'unknown'), iterations, primal objective, dual objective, gap.
    resident=True runs the bookkeeping on the device as well (BatchKkt.coneqp); the NumPy loop below is the
    readable restatement the device loop is tested against."""
    B, n = q.shape
    m = h.shape[1]
    own = kkt is None
    if own:
        kkt = BatchKkt(Gt, P, device=device)
    if resident:
        try:
            return kkt.coneqp(q, h, maxiters=maxiters, abstol=abstol, reltol=reltol, feastol=feastol,
                              use_correction=use_correction)
        finally:
            if own:
                kkt.close()
    Psym = None
    if not hasattr(kkt, "products"):      # host products (tests): only tril(P) is meaningful (coneprog.py:1475-1477)
        Psym = P if P is not None else np.zeros((B, n, n))
        Psym = np.tril(Psym) + np.transpose(np.tril(Psym, -1), (0, 2, 1))

    def Gx(x):           # (B,n) -> (B,m)
        return np.einsum('bnm,bn->bm', Gt, x)

    def GTz(z):          # (B,m) -> (B,n)
        return np.einsum('bnm,bm->bn', Gt, z)

    def dot(a, b):
        return np.einsum('bi,bi->b', a, b)

    resx0 = np.maximum(1.0, np.sqrt(dot(q, q)))                      # coneprog.py:1998-2000
    resz0 = np.maximum(1.0, np.sqrt(dot(h, h)))

    # ---- starting point: W = I (coneprog.py:2055-2106)
    info = kkt.factor(np.ones((B, m)))
    if np.any(info > 0):
        raise ValueError("Rank([P; A; G]) < n for problems %s" % np.nonzero(info > 0)[0][:8])
    x = -q.copy()
    z = h.copy()
    kkt.solve(x, z)
    s = -z
    nrms = np.sqrt(dot(s, s))
    ts = np.max(-s, axis=1)
    shift = ts >= -1e-8 * np.maximum(nrms, 1.0)
    s[shift] += (1.0 + ts[shift])[:, None]
    nrmz = np.sqrt(dot(z, z))
    tz = np.max(-z, axis=1)
    shift = tz >= -1e-8 * np.maximum(nrmz, 1.0)
    z[shift] += (1.0 + tz[shift])[:, None]

    gap = dot(s, z)
    active = np.ones(B, dtype=bool)
    out = {'x': np.zeros((B, n)), 's': np.zeros((B, m)), 'z': np.zeros((B, m)),
           'status': np.array(['unknown'] * B, dtype=object), 'iterations': np.zeros(B, dtype=int),
           'primal objective': np.zeros(B), 'dual objective': np.zeros(B), 'gap': np.zeros(B)}
    d = np.ones((B, m))
    lmbda = np.ones((B, m))

    def finish(mask, status, it, pcost, dcost):
        out['x'][mask], out['s'][mask], out['z'][mask] = x[mask], s[mask], z[mask]
        out['status'][mask] = status
        out['iterations'][mask] = it
        out['primal objective'][mask], out['dual objective'][mask], out['gap'][mask] = pcost[mask], dcost[mask], gap[mask]

    for it in range(maxiters + 1):
        # residuals and stopping test (coneprog.py:2170-2234)
        if hasattr(kkt, "products"):
            gx, gtz, px = kkt.products(x, z)
        else:
            gx, gtz, px = Gx(x), GTz(z), np.einsum('bij,bj->bi', Psym, x)
        rx = q + px
        f0 = 0.5 * (dot(x, rx) + dot(x, q))
        rx = rx + gtz
        resx = np.sqrt(dot(rx, rx))
        rz = s + gx - h
        resz = np.sqrt(dot(rz, rz))
        pcost = f0
        dcost = f0 + dot(z, rz) - gap
        with np.errstate(divide='ignore', invalid='ignore'):
            relgap = np.where(pcost < 0.0, gap / -pcost, np.where(dcost > 0.0, gap / dcost, np.inf))
        pres = resz / resz0
        dres = resx / resx0
        done = (pres <= feastol) & (dres <= feastol) & ((gap <= abstol) | (relgap <= reltol))
        if it == maxiters:
            finish(active & ~done, 'unknown', it, pcost, dcost)
            finish(active & done, 'optimal', it, pcost, dcost)
            active[:] = False
        else:
            finish(active & done, 'optimal', it, pcost, dcost)
            active &= ~done
        if not active.any():
            break

        if it == 0:                                                    # misc.py:284-287
            d = np.sqrt(s / z)
            lmbda = np.sqrt(s * z)
        lmbdasq = lmbda * lmbda                                        # misc.ssqr
        di = 1.0 / d
        di_safe = np.where(active[:, None], di, 1.0)                   # finished problems keep a benign system
        info = kkt.factor(di_safe)
        bad = active & (info > 0)
        if bad.any():                                                  # coneprog.py:2256-2275
            if it == 0:
                raise ValueError("Rank([P; A; G]) < n for problems %s" % np.nonzero(bad)[0][:8])
            finish(bad, 'unknown', it, pcost, dcost)
            active &= ~bad
            if not active.any():
                break

        mu = gap / m
        sigma = np.zeros(B)
        ws3 = np.zeros((B, m))
        for i in (0, 1):                                               # coneprog.py:2360-2456
            ds = -lmbdasq + (sigma * mu)[:, None]
            if i == 1 and use_correction:                             # coneprog.py:2377-2378
                ds = ds - ws3
            dx = -rx.copy()
            dz = -rz.copy()
            ds = ds / lmbda                                            # sinv
            dz = dz - d * ds                                           # z -= W' s
            dx = np.ascontiguousarray(dx)
            dz = np.ascontiguousarray(dz)
            kkt.solve(dx, dz)                                          # f3
            ds = ds - dz
            dsdz = dot(ds, dz)
            if i == 0:
                ws3 = ds * dz                                          # sprod
            ds = ds / lmbda                                            # scale2
            dz = dz / lmbda
            t = np.maximum(0.0, np.maximum(np.max(-ds, axis=1), np.max(-dz, axis=1)))
            with np.errstate(divide='ignore'):
                step = np.where(t == 0.0, 1.0, np.minimum(1.0, (1.0 if i == 0 else 0.99) / t))
            if i == 0:
                with np.errstate(divide='ignore', invalid='ignore'):
                    sigma = np.minimum(1.0, np.maximum(0.0, 1.0 - step + dsdz / gap * step ** 2)) ** 3
                sigma = np.where(np.isfinite(sigma), sigma, 0.0)

        upd = active[:, None]
        x = np.where(upd, x + step[:, None] * dx, x)
        ds = (1.0 + step[:, None] * ds) * lmbda                        # coneprog.py:2471-2491
        dz = (1.0 + step[:, None] * dz) * lmbda
        with np.errstate(invalid='ignore'):
            ds = np.sqrt(ds)                                           # misc.py:450-464
            dz = np.sqrt(dz)
        d_new = d * ds / dz
        l_new = ds * dz
        d = np.where(upd, d_new, d)
        lmbda = np.where(upd, l_new, lmbda)
        s = np.where(upd, d * lmbda, s)                                # coneprog.py:2525-2545
        z = np.where(upd, lmbda / d, z)
        gap = np.where(active, dot(lmbda, lmbda), gap)
    if own:
        kkt.close()
    return out


# ---------------------------------------------------------------------------------------------------
# sharding over ranks (RCCL / gloo): scatter the batch, solve locally, gather
# ---------------------------------------------------------------------------------------------------
def shard_bounds(B, world):
    """Contiguous shards; the first B % world ranks get one extra problem."""
    base, rem = divmod(B, world)
    starts = [r * base + min(r, rem) for r in range(world + 1)]
    return [(starts[r], starts[r + 1]) for r in range(world)]


class _ForeignMemory(object):
    """CUDA array interface over device memory this process did not allocate (an IPC mapping)"""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False), "version": 2,
                                         "strides": None}


def _foreign_tensor(torch, ptr, shape, dev):
    """A zero-copy view of an IPC mapping.  NO `device=` for torch.as_tensor: the tensor must stay on the device that OWNS the memory
    (the root's GPU, as torch learns from the pointer's attributes) -- asking for this rank's device would make torch copy the
    whole exported allocation across xGMI at every solve; the row ranges a rank needs are moved by the cross-device `copy_` of
    `_solve_ipc` (a peer copy of exactly those rows)."""
    if int(np.prod(shape)) == 0:
        return torch.empty(shape, dtype=torch.float64, device=dev)
    return torch.as_tensor(_ForeignMemory(ptr, shape))


_STATUS_NAMES = np.array(['unknown', 'optimal', 'unknown', 'unknown'], dtype=object)


class ShardedBatch(object):
    """Persistent per-rank state of the sharded batch solve (BASELINE configs[4]).

    Everything that costs allocation time lives as long as this object: the rank's receive buffers for its shard of (q, h, Gt, P),
    the packed result buffer, and -- on GPUs -- one `BatchKkt` engine per SUB-BATCH of the shard (the ~GB of engine state per
    512 problems is allocated once, not inside a timed step).

    A `solve()` is pipelined over `nsub` sub-batches of every rank's shard: all scatters are enqueued at once (async, in sub-batch
    order), sub-batch k is solved as soon as ITS pieces have arrived while the pieces of k+1.. are still on the wire, and its
    packed results go back with an async gather that overlaps the next solve.  Data path per sub-batch: `dist.scatter` of
    contiguous row ranges (RCCL: grouped send/recv, one xGMI link per peer) -> device-to-device copy into the engine
    (`set_problem`) -> device-resident coneqp loop -> ONE packed float64 tensor [x | s | z | pcost dcost gap status iters] ->
    `dist.gather`.  No collective inside the interior-point loop.
    """

    def __init__(self, B, n, m, hasP, group=None, root=0, nsub=4, local_solver=None, device_of_rank=None, engine_factory=None,
                 transport="rccl"):
        """engine_factory(shape=(cnt, n, m), device=index) -> an object with BatchKkt's set_problem / coneqp / close: the
        device-resident branch then runs with it whatever the backend (tests walk the RCCL branch's exact ordering -- scatter
        lists, async work handles, wait order, packed gather -- over gloo with host tensors this way); None: `BatchKkt` on RCCL.

        transport = "rccl": `dist.scatter` / `dist.gather` of the group's backend (RCCL: grouped send / recv).
        transport = "ipc": NO data-path collective.  The root exports the allocations that hold q, h, Gt, P and its result buffer
        (mi355kkt_ipc_export: hipIpcGetMemHandle); every rank maps them once (mi355kkt_ipc_open) and PULLS the rows of its shard
        with device-to-device copies on a stream of its own, sub-batch by sub-batch, and PUSHES its packed result rows straight
        into the root's result buffer -- seven independent point-to-point streams over seven xGMI links, nothing for a collective
        library to schedule or serialise.  The group only carries the handles (an object broadcast) and the closing barrier, so it
        may be gloo: the ranks then only need a visible GPU each -- or all the SAME one (how the path is tested on a one-GPU box)."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group, self.root = group, root
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        if transport not in ("rccl", "ipc"):
            raise ValueError("ShardedBatch: transport must be 'rccl' or 'ipc'")
        self.transport = transport
        if transport == "ipc" and not torch.cuda.is_available():
            raise RuntimeError("ShardedBatch(transport='ipc') moves device memory between the ranks' GPUs: no HIP device is visible "
                               "(there is no CPU fallback; transport='rccl' over a gloo group is the host-side dry run)")
        on_gpu = self.backend == "nccl" or transport == "ipc"
        self.dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
        self.B, self.n, self.m, self.hasP = int(B), int(n), int(m), bool(hasP)
        self.local_solver, self.device_of_rank = local_solver, device_of_rank
        self.bounds = shard_bounds(self.B, self.world)
        self.lo, self.hi = self.bounds[self.rank]
        self.nloc = self.hi - self.lo
        self.mx = max(b - a for a, b in self.bounds)
        self.nsub = max(1, min(int(nsub), self.mx))
        # sub-batch k covers rows [sb[k], sb[k+1]) of every rank's shard (a short shard simply has fewer rows in the last ones)
        self.sb = [k * self.mx // self.nsub for k in range(self.nsub + 1)]
        self.width = self.n + 2 * self.m + 5
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.q_l = torch.empty((self.mx, self.n), **f64)
        self.h_l = torch.empty((self.mx, self.m), **f64)
        self.G_l = torch.empty((self.mx, self.n, self.m), **f64)
        self.P_l = torch.empty((self.mx, self.n, self.n), **f64) if self.hasP else None
        self.pack = torch.zeros((self.mx, self.width), **f64)
        self.gathered = None
        if self.rank == root and transport != "ipc":
            self.gathered = [[torch.empty((self.sb[k + 1] - self.sb[k], self.width), **f64) for _ in range(self.world)]
                             for k in range(self.nsub)]
        self.engine_factory = engine_factory
        self.on_device = engine_factory is not None or (on_gpu and local_solver is None)
        self.engines = [None] * self.nsub          # BatchKkt per sub-batch, created at the first solve and kept
        self.last_timings = {}
        if transport == "ipc":
            self._ipc_init()

    def close(self):
        for e in self.engines:
            if e is not None:
                e.close()
        self.engines = [None] * self.nsub
        if getattr(self, "_ipc_open", None):
            if self.dev.type == "cuda":
                self.torch.cuda.synchronize(self.dev)
            for base in self._ipc_open.values():
                _capi.lib().mi355kkt_ipc_close(C.c_void_p(base))
            self._ipc_open = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- transport = "ipc" -------------------------------------------------------------------------------------------------
    def _ipc_init(self):
        torch = self.torch
        self._ipc_open = {}                       # handle bytes -> base address of the mapping in this process
        self._ipc_seen = set()                    # handles the current solve has used (mappings of allocations the root no longer
                                                  # exports are closed at the end of a solve once more than a handful have piled up)
        self.pull_stream = torch.cuda.Stream(self.dev)
        self.push_stream = torch.cuda.Stream(self.dev)
        self.result = None                        # root: (B, width), problem order; exported once
        self._result_ref = None                   # others: the root's result buffer, mapped
        if self.rank == self.root:
            self.result = torch.zeros((self.B, self.width), dtype=torch.float64, device=self.dev)

    def _ipc_export(self, t):
        """(handle bytes, byte offset, shape) of a contiguous float64 CUDA tensor of this process"""
        hd = (C.c_ubyte * 64)()
        off, size = C.c_int64(), C.c_int64()
        _capi.check(_capi.lib().mi355kkt_ipc_export(C.c_void_p(t.data_ptr()), hd, C.byref(off), C.byref(size)), "ipc_export")
        return bytes(hd), int(off.value), tuple(int(v) for v in t.shape)

    def _ipc_view(self, desc):
        """a tensor of this process over memory another process exported (mapped on first sight of its allocation)"""
        hd, off, shape = desc
        base = self._ipc_open.get(hd)
        self._ipc_seen.add(hd)
        if base is None:
            out = C.c_void_p()
            buf = (C.c_ubyte * 64).from_buffer_copy(hd)
            _capi.check(_capi.lib().mi355kkt_ipc_open(buf, C.byref(out)), "ipc_open")
            base = self._ipc_open[hd] = int(out.value)
        return _foreign_tensor(self.torch, base + off, shape, self.dev)

    def _solve_ipc(self, P, q, Gt, h, return_device, opts):
        import time
        torch, dist = self.torch, self.dist
        n, m, root = self.n, self.m, self.root
        t0 = time.perf_counter()
        descs = [None]
        if self.rank == root:
            fulls = [self._as_tensor(q), self._as_tensor(h), self._as_tensor(Gt), self._as_tensor(P) if self.hasP else None]
            fulls = [None if t is None else t.contiguous() for t in fulls]
            torch.cuda.current_stream(self.dev).synchronize()       # the data is in HBM before anybody is told where it is
            descs = [[None if t is None else self._ipc_export(t) for t in fulls] + [self._ipc_export(self.result)]]
        dist.broadcast_object_list(descs, src=root, group=self.group)
        if self.rank == root:
            srcs, result = fulls, self.result
        else:
            d = descs[0]
            srcs = [None if e is None else self._ipc_view(e) for e in d[:4]]
            result = self._ipc_view(d[4])
        # every pull is enqueued now, in the order it will be consumed; an event per sub-batch
        dsts = [self.q_l, self.h_l, self.G_l, self.P_l]
        pulled = []
        with torch.cuda.stream(self.pull_stream):
            for k in range(self.nsub):
                a, b = self._rows(k)
                if b > a:
                    for src, dst in zip(srcs, dsts):
                        if src is not None:
                            dst[a:b].copy_(src[self.lo + a:self.lo + b], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.pull_stream)
                pulled.append(ev)
        tm = {"scatter_exposed": 0.0, "scatter_all": 0.0, "upload": 0.0, "solve": 0.0, "pack": 0.0, "gather_exposed": 0.0}
        lockstep = 0
        cur = torch.cuda.current_stream(self.dev)
        for k in range(self.nsub):
            pulled[k].synchronize()
            t1 = time.perf_counter()
            if k == 0:
                tm["scatter_exposed"] = 1e3 * (t1 - t0)
            tm["scatter_all"] = 1e3 * (t1 - t0)
            a, b = self._rows(k)
            cnt = b - a
            if cnt <= 0:
                continue
            if self.engines[k] is None:
                make = self.engine_factory or BatchKkt
                self.engines[k] = make(shape=(cnt, n, m), device=self.dev.index)
            eng = self.engines[k]
            t2 = time.perf_counter()
            eng.set_problem(self.G_l[a:b], self.P_l[a:b] if self.hasP else None)
            t3 = time.perf_counter()
            res = eng.coneqp(self.q_l[a:b], self.h_l[a:b], **opts)
            t3b = time.perf_counter()
            tm["upload"] += 1e3 * (t3 - t2)
            tm["solve"] += 1e3 * (t3b - t3)
            pk = self.pack[a:b]
            pk[:, :n] = res['x']
            pk[:, n:n + m] = res['s']
            pk[:, n + m:n + 2 * m] = res['z']
            pk[:, n + 2 * m] = res['primal objective']
            pk[:, n + 2 * m + 1] = res['dual objective']
            pk[:, n + 2 * m + 2] = res['gap']
            pk[:, n + 2 * m + 3] = res['status_code'].to(torch.float64)
            pk[:, n + 2 * m + 4] = res['iterations'].to(torch.float64)
            lockstep = max(lockstep, int(res['lockstep iterations']))
            # this sub-batch's rows go into the root's result buffer (problem order) while the next one is solved
            done = torch.cuda.Event()
            done.record(cur)
            self.push_stream.wait_event(done)
            with torch.cuda.stream(self.push_stream):
                result[self.lo + a:self.lo + b].copy_(pk, non_blocking=True)
            tm["pack"] += 1e3 * (time.perf_counter() - t3b)
        t4 = time.perf_counter()
        self.push_stream.synchronize()
        self.pull_stream.synchronize()
        dist.barrier(group=self.group)            # control plane only: every shard is in the root's buffer, nobody reads the inputs any more
        if len(self._ipc_open) > 8:               # a root that exports fresh allocations at every solve: drop the mappings of the old ones
            for hd in [h_ for h_ in self._ipc_open if h_ not in self._ipc_seen]:
                _capi.lib().mi355kkt_ipc_close(C.c_void_p(self._ipc_open.pop(hd)))
        self._ipc_seen = set()
        t5 = time.perf_counter()
        tm["gather_exposed"] = 1e3 * (t5 - t4)
        tm["total"] = 1e3 * (t5 - t0)
        self.last_timings = tm
        t = self.result if self.rank == root else self.pack[:self.nloc]
        if return_device:
            t = t.clone()                          # (persistent buffers: the next solve() overwrites them)
        return self._unpack(t, lockstep, return_device)

    def _unpack(self, t, ls, return_device):
        n, m = self.n, self.m
        sc = t[:, n + 2 * m:].cpu().numpy()                        # pcost dcost gap status iters: O(B) doubles
        if return_device:
            vec = {'x': t[:, :n], 's': t[:, n:n + m], 'z': t[:, n + m:n + 2 * m]}
        else:
            a_ = t[:, :n + 2 * m].cpu().numpy()
            vec = {'x': a_[:, :n].copy(), 's': a_[:, n:n + m].copy(), 'z': a_[:, n + m:n + 2 * m].copy()}
        vec.update({'primal objective': sc[:, 0].copy(), 'dual objective': sc[:, 1].copy(), 'gap': sc[:, 2].copy(),
                    'status': _STATUS_NAMES[sc[:, 3].astype(int)], 'iterations': sc[:, 4].astype(int),
                    'lockstep iterations': ls})
        return vec

    def _rows(self, k):
        """rows of MY shard in sub-batch k"""
        a = min(self.sb[k], self.nloc)
        return a, min(self.sb[k + 1], self.nloc)

    def _as_tensor(self, arr):
        torch = self.torch
        if hasattr(arr, "data_ptr"):
            return arr if arr.device == self.dev else arr.to(self.dev)
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)).to(self.dev)

    def _scatter_sub(self, k, full, buf):
        """rows [sb[k], sb[k+1]) of every rank's shard of `full` (root only) -> buf[sb[k]:sb[k+1]] on every rank, async"""
        torch, dist = self.torch, self.dist
        a, b = self.sb[k], self.sb[k + 1]
        out = buf[a:b]
        if self.rank != self.root:
            return dist.scatter(out, None, src=self.root, group=self.group, async_op=True)
        chunks = []
        for lo, hi in self.bounds:
            ra, rb = min(lo + a, hi), min(lo + b, hi)
            c = full[ra:rb]                      # a view: whole pieces go out without a copy
            if rb - ra < b - a:                  # dist.scatter needs equal sizes: only the short tail pieces are padded
                c = torch.cat([c, torch.zeros((b - a - (rb - ra),) + tuple(full.shape[1:]), dtype=torch.float64, device=self.dev)])
            chunks.append(c.contiguous())
        return dist.scatter(out, chunks, src=self.root, group=self.group, async_op=True)

    def _wait(self, works):
        for w in works:
            w.wait()
        if self.dev.type == "cuda":              # host waits for THESE collectives only (later ones keep flying)
            self.torch.cuda.current_stream(self.dev).synchronize()

    def solve(self, P, q, Gt, h, return_device=False, **opts):
        """P, q, Gt, h are only read on the root (other ranks may pass None).  The root returns the FULL gathered result dict,
        the other ranks their local shard's.  `self.last_timings` (ms, this rank): scatter_exposed (until the first sub-batch
        was complete here), scatter_all (until the last one was), upload, solve (sum over sub-batches), pack (results into the packed
        gather rows), gather_exposed (after the last solve), total."""
        import time
        if self.transport == "ipc":
            if not (self.on_device and opts.get("resident", True)):
                raise NotImplementedError("ShardedBatch(transport='ipc') serves the device-resident solve only")
            return self._solve_ipc(P, q, Gt, h, return_device, {k_: v for k_, v in opts.items() if k_ != "resident"})
        torch, dist = self.torch, self.dist
        n, m, root = self.n, self.m, self.root
        t0 = time.perf_counter()
        fulls = None
        if self.rank == root:
            fulls = (self._as_tensor(q), self._as_tensor(h), self._as_tensor(Gt), self._as_tensor(P) if self.hasP else None)
        works = []
        for k in range(self.nsub):                # everything is enqueued up front, in the order it will be consumed
            wk = [self._scatter_sub(k, fulls[0] if fulls else None, self.q_l),
                  self._scatter_sub(k, fulls[1] if fulls else None, self.h_l),
                  self._scatter_sub(k, fulls[2] if fulls else None, self.G_l)]
            if self.hasP:
                wk.append(self._scatter_sub(k, fulls[3] if fulls else None, self.P_l))
            works.append(wk)
        tm = {"scatter_exposed": 0.0, "scatter_all": 0.0, "upload": 0.0, "solve": 0.0, "pack": 0.0, "gather_exposed": 0.0}
        gworks, lockstep = [], 0
        sopts = {k_: v for k_, v in opts.items() if k_ != "resident"}
        for k in range(self.nsub):
            self._wait(works[k])
            t1 = time.perf_counter()
            if k == 0:
                tm["scatter_exposed"] = 1e3 * (t1 - t0)
            tm["scatter_all"] = 1e3 * (t1 - t0)
            a, b = self._rows(k)
            cnt = b - a
            if cnt > 0:
                G_k, P_k = self.G_l[a:b], (self.P_l[a:b] if self.hasP else None)
                q_k, h_k = self.q_l[a:b], self.h_l[a:b]
                if self.on_device and opts.get("resident", True):
                    if self.engines[k] is None:
                        make = self.engine_factory or BatchKkt
                        self.engines[k] = make(shape=(cnt, n, m), device=self.dev.index)
                    eng = self.engines[k]
                    t2 = time.perf_counter()
                    eng.set_problem(G_k, P_k)
                    t3 = time.perf_counter()
                    res = eng.coneqp(q_k, h_k, **sopts)
                    t3b = time.perf_counter()
                    tm["upload"] += 1e3 * (t3 - t2)
                    tm["solve"] += 1e3 * (t3b - t3)
                    pk = self.pack[a:b]
                    pk[:, :n] = res['x']
                    pk[:, n:n + m] = res['s']
                    pk[:, n + m:n + 2 * m] = res['z']
                    pk[:, n + 2 * m] = res['primal objective']
                    pk[:, n + 2 * m + 1] = res['dual objective']
                    pk[:, n + 2 * m + 2] = res['gap']
                    pk[:, n + 2 * m + 3] = res['status_code'].to(torch.float64)
                    pk[:, n + 2 * m + 4] = res['iterations'].to(torch.float64)
                    lockstep = max(lockstep, int(res['lockstep iterations']))
                    tm["pack"] += 1e3 * (time.perf_counter() - t3b)       # (enqueue time of the packing copies on a GPU)
                else:
                    t3 = time.perf_counter()
                    Pn = P_k.cpu().numpy() if P_k is not None else None
                    qn, Gn, hn = q_k.cpu().numpy(), G_k.cpu().numpy(), h_k.cpu().numpy()
                    if self.local_solver is None:
                        device = self.device_of_rank(self.rank) if self.device_of_rank else (
                            torch.cuda.current_device() if self.backend == "nccl" else 0)
                        res = coneqp_batch(Pn, qn, Gn, hn, device=device, **sopts)
                    else:
                        res = self.local_solver(Pn, qn, Gn, hn, **sopts)
                    host = np.zeros((cnt, self.width))
                    host[:, :n], host[:, n:n + m], host[:, n + m:n + 2 * m] = res['x'], res['s'], res['z']
                    host[:, n + 2 * m], host[:, n + 2 * m + 1] = res['primal objective'], res['dual objective']
                    host[:, n + 2 * m + 2] = res['gap']
                    st_ = np.asarray(res['status'])
                    host[:, n + 2 * m + 3] = (st_ == 'optimal').astype(float) + 2.0 * (st_ != 'optimal')
                    host[:, n + 2 * m + 4] = res['iterations']
                    self.pack[a:b] = torch.from_numpy(host).to(self.dev)
                    lockstep = max(lockstep, int(res.get('lockstep iterations', 0)))
                    tm["solve"] += 1e3 * (time.perf_counter() - t3)
            # this sub-batch's results travel while the next one is solved
            sa, sbk = self.sb[k], self.sb[k + 1]
            gworks.append(dist.gather(self.pack[sa:sbk], self.gathered[k] if self.rank == root else None, dst=root,
                                      group=self.group, async_op=True))
        t4 = time.perf_counter()
        self._wait(gworks)
        t5 = time.perf_counter()
        tm["gather_exposed"] = 1e3 * (t5 - t4)
        tm["total"] = 1e3 * (t5 - t0)
        self.last_timings = tm

        def unpack(t, ls):
            sc = t[:, n + 2 * m:].cpu().numpy()                    # pcost dcost gap status iters: O(B) doubles
            if return_device:
                vec = {'x': t[:, :n], 's': t[:, n:n + m], 'z': t[:, n + m:n + 2 * m]}
            else:
                a_ = t[:, :n + 2 * m].cpu().numpy()
                vec = {'x': a_[:, :n].copy(), 's': a_[:, n:n + m].copy(), 'z': a_[:, n + m:n + 2 * m].copy()}
            vec.update({'primal objective': sc[:, 0].copy(), 'dual objective': sc[:, 1].copy(), 'gap': sc[:, 2].copy(),
                        'status': _STATUS_NAMES[sc[:, 3].astype(int)], 'iterations': sc[:, 4].astype(int),
                        'lockstep iterations': ls})
            return vec
        if self.rank != root:
            # (a copy: self.pack is persistent and the next solve() overwrites it -- a caller holding return_device views of
            #  the previous result must not see them change)
            return unpack(self.pack[:self.nloc].clone() if return_device else self.pack[:self.nloc], lockstep)
        # rank r's shard = its rows of every sub-batch, in order
        pieces = []
        for r, (lo, hi) in enumerate(self.bounds):
            cnt_r = hi - lo
            for k in range(self.nsub):
                a, b = min(self.sb[k], cnt_r), min(self.sb[k + 1], cnt_r)
                if b > a:
                    pieces.append(self.gathered[k][r][:b - a])
        full = unpack(torch.cat(pieces), lockstep)           # the root's own lock-step count (shards stop independently)
        return full


def ipc_transport_works(group=None, root=0):
    """Collective (every rank of `group` calls it): can every rank map an allocation of the root's GPU and read it?  A 4 KB
    probe through exactly the calls ShardedBatch(transport="ipc") makes -- mi355kkt_ipc_export on the root, mi355kkt_ipc_open +
    a device-to-device copy everywhere else.  Returns (ok, reasons): ok only if EVERY rank succeeded, so that all ranks choose
    the same transport (`bench.py --gpus N --transport auto`)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device())
    L = _capi.lib()
    desc, src, why, base = [None], None, "", None
    try:
        if rank == root:
            src = torch.arange(512, dtype=torch.float64, device=dev)
            torch.cuda.current_stream(dev).synchronize()
            hd = (C.c_ubyte * 64)()
            off, size = C.c_int64(), C.c_int64()
            _capi.check(L.mi355kkt_ipc_export(C.c_void_p(src.data_ptr()), hd, C.byref(off), C.byref(size)), "ipc_export")
            desc = [(bytes(hd), int(off.value))]
    except Exception as e:                     # the broadcast below must still happen on every rank
        why = "export: %r" % (e,)
    dist.broadcast_object_list(desc, src=root, group=group)
    if rank != root:
        try:
            if desc[0] is None:
                raise RuntimeError("the root could not export")
            out = C.c_void_p()
            _capi.check(L.mi355kkt_ipc_open((C.c_ubyte * 64).from_buffer_copy(desc[0][0]), C.byref(out)), "ipc_open")
            base = int(out.value)
            got = torch.empty(512, dtype=torch.float64, device=dev)
            got.copy_(_foreign_tensor(torch, base + desc[0][1], (512,), dev))
            torch.cuda.current_stream(dev).synchronize()
            if not torch.equal(got.cpu(), torch.arange(512, dtype=torch.float64)):
                raise RuntimeError("the mapped allocation does not hold the root's data")
        except Exception as e:
            why = "open / copy: %r" % (e,)
        finally:
            if base is not None:
                L.mi355kkt_ipc_close(C.c_void_p(base))
    whys = [None] * world
    dist.all_gather_object(whys, why, group=group)
    del src
    return all(not w for w in whys), whys


_SHARDED_CACHE = {}


def coneqp_batch_sharded(P, q, Gt, h, group=None, root=0, local_solver=None, device_of_rank=None, return_device=False, nsub=4,
                         transport="rccl", **opts):
    """P, q, Gt, h are only read on `root` (other ranks may pass None): NumPy arrays, or -- with RCCL -- float64 CUDA
    tensors already resident in the root's HBM (then the scatter sends views of them, nothing is staged or copied).
    Root returns the FULL gathered result dict (NumPy), the other ranks their local shard's.

    Thin wrapper around a cached `ShardedBatch` (one per group / problem shape): receive buffers and the per-rank engines are
    created on the first call and reused, so a repeated call costs the data path only -- pipelined scatter of `nsub` sub-batches
    -> local device-resident solve of each as it arrives -> async gather of its packed results.  No collective inside the
    interior-point loop; with RCCL nothing but the per-iteration "still active" word of the local loop touches a host.
    return_device=True leaves x, s, z of the result as tensors on the gathering device."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    meta = [None]
    if rank == root:
        meta = [(int(q.shape[0]), int(q.shape[1]), int(h.shape[1]), P is not None)]
    dist.broadcast_object_list(meta, src=root, group=group)
    B, n, m, hasP = meta[0]
    # keyed on the group OBJECT (kept alive by the cache entry, so its id cannot be recycled under us) and this rank in it;
    # local_solver is not part of the key (it is re-assigned below: a fresh lambda per call must not rebuild the engines)
    key = (group, rank, root, B, n, m, hasP, int(nsub), dist.get_backend(group), dist.get_world_size(group), transport)
    sb = _SHARDED_CACHE.get(key)
    if sb is not None and sb.on_device != (sb.engine_factory is not None or
                                           ((sb.backend == "nccl" or transport == "ipc") and local_solver is None)):
        sb = None                                     # device-resident <-> host-solver switch: rebuild
    if sb is None:
        for old in list(_SHARDED_CACHE.values()):    # one shape at a time: the engines hold GBs
            old.close()
        _SHARDED_CACHE.clear()
        sb = _SHARDED_CACHE[key] = ShardedBatch(B, n, m, hasP, group=group, root=root, nsub=nsub, local_solver=local_solver,
                                                device_of_rank=device_of_rank, transport=transport)
    sb.local_solver = local_solver
    return sb.solve(P, q, Gt, h, return_device=return_device, **opts)


def clear_sharded_cache():
    """closes the cached ShardedBatch (engines, buffers): call before torch.distributed.destroy_process_group() -- the cache
    entry holds the group it was built for"""
    for old in list(_SHARDED_CACHE.values()):
        old.close()
    _SHARDED_CACHE.clear()

// Device-resident bookkeeping of the reference coneqp loop for one problem with
// dims = {'l': ml, 'q': [...]}: one 256-thread workgroup; the cone-vector operations live in cone_ops.h.  Each kernel
// restates a stretch of src/python/coneprog.py, operation for operation:
//     qp_start_kernel       :2083-2106   s = -z, shifts into the cone interior, gap, resx0 / resy0 / resz0
//     qp_residual_kernel    :2170-2234   residuals, costs, stopping test, compute_scaling at iteration 0, lmbdasq, mu
//     qp_build_kernel       :2376-2399   right-hand side (dx, dy, dz, ds), saved for the refinement
//     qp_f4pre / f4post     f4_no_ir :2303-2316
//     qp_res_a / res_b      res() :1930-1961 (the refinement step of f4, :2330-2345; default 1 with 'q' blocks, :1862-1865)
//     qp_step_kernel        :2420-2456   dsdz, Mehrotra product, scale2, step to the boundary, sigma
//     qp_update_kernel      :2459-2547   iterate, scaling update (misc.py:444-464, :503-573), unscaled s, z, gap
// (The LP-cone batch variant is batch_ipm.hip.)
#include "cone_ops.h"

namespace mi355kkt {

// batched mode: workgroup b works on problem b — shift every per-problem pointer of (its copy of) the state
__device__ __forceinline__ void qp_select(QpState& S) {
    const int64_t b = blockIdx.x;
    if (S.nbatch <= 1 || b == 0) return;
    const int64_t on = b * S.n, op = b * (S.p > 0 ? S.p : 1), om = b * S.m;
    const int64_t ov = b * (S.lq > S.ml ? S.lq - S.ml : 1), ob = b * (S.nq > 0 ? S.nq : 1);
    S.q += on; S.x += on; S.dx += on; S.rx += on; S.Px += on; S.GTz += on; S.ATy += on; S.x_out += on; S.wx += on; S.wx2 += on;
    S.b += op; S.y += op; S.dy += op; S.ry += op; S.Ax += op; S.y_out += op; S.wy += op; S.wy2 += op;
    S.h += om; S.s += om; S.z += om; S.ds += om; S.dz += om; S.rz += om; S.lmbda += om; S.lmbdasq += om; S.d += om; S.di += om;
    S.ws3 += om; S.Gx += om; S.s_out += om; S.z_out += om; S.t1 += om; S.t2 += om; S.wz3 += om; S.ws += om; S.wz += om;
    S.ws2 += om; S.wz2 += om;
    S.v += ov; S.beta += ob;
    S.sc += b * QP_NSC;
    S.active += b; S.status += b; S.iters += b;
}
__device__ __forceinline__ void qp_select(const QpState& S, QpBuf& X) {
    const int64_t b = blockIdx.x;
    if (S.nbatch <= 1 || b == 0) return;
    X.x += b * S.n; X.y += b * (S.p > 0 ? S.p : 1); X.z += b * S.m; X.s += b * S.m;
}

__device__ __forceinline__ void qp_store_result(const QpState& S, int status, int it) {
    const int tid = threadIdx.x;
    for (int i = tid; i < S.n; i += blockDim.x) S.x_out[i] = S.x[i];
    for (int i = tid; i < S.p; i += blockDim.x) S.y_out[i] = S.y[i];
    for (int i = tid; i < S.m; i += blockDim.x) {
        S.s_out[i] = S.s[i];
        S.z_out[i] = S.z[i];
    }
    if (tid == 0) {
        S.status[0] = status;       // 1 optimal, 2 unknown (iteration limit), 3 unknown (singular KKT matrix)
        S.iters[0] = it;
        S.active[0] = 0;
    }
}

// W = I (coneprog.py:2054-2063)
__global__ __launch_bounds__(1024) void qp_unit_scaling_kernel(QpState S) {
    qp_select(S);
    const int tid = threadIdx.x;
    for (int i = tid; i < S.ml; i += blockDim.x) { S.d[i] = 1.0; S.di[i] = 1.0; }
    for (int i = tid; i < S.lq - S.ml; i += blockDim.x) S.v[i] = 0.0;
    __syncthreads();
    for (int k = tid; k < S.nq; k += blockDim.x) { S.v[S.qoff[k] - S.ml] = 1.0; S.beta[k] = 1.0; }
    for (int k = 0; k < S.ns; ++k) {                     // r_k = rti_k = I
        const int mk = S.sdim[k], o = S.soff[k] - S.lq;
        for (int e = tid; e < mk * mk; e += blockDim.x) S.r[o + e] = S.rti[o + e] = (e % mk == e / mk) ? 1.0 : 0.0;
    }
}

// given != 0: s and z are the caller's starting point (initvals, coneprog.py:2109-2149): no s = -z, no shift into the interior
__global__ __launch_bounds__(1024) void qp_start_kernel(QpState S, int given) {
    qp_select(S);
    __shared__ double sh[16];
    const int tid = threadIdx.x, m = S.m;
    double* sc = S.sc;
    const double q2 = lp_dot(S.q, S.q, S.n, sh), h2 = lp_dot(S.h, S.h, m, sh);
    const double b2 = S.p > 0 ? lp_dot(S.b, S.b, S.p, sh) : 0.0;
    if (!given) {
        for (int i = tid; i < m; i += blockDim.x) S.s[i] = -S.z[i];
        __syncthreads();
        const double ns = sqrt(lp_dot(S.s, S.s, m, sh));
        const double ts = cv_maxstep(S, S.s, sh);
        if (ts >= -1e-8 * fmax(ns, 1.0)) cv_add_e(S, S.s, 1.0 + ts);
        const double nz = sqrt(lp_dot(S.z, S.z, m, sh));
        const double tz = cv_maxstep(S, S.z, sh);
        if (tz >= -1e-8 * fmax(nz, 1.0)) cv_add_e(S, S.z, 1.0 + tz);
    }
    __syncthreads();
    const double g = lp_dot(S.s, S.z, m, sh);
    if (tid == 0) {
        sc[QP_RESX0] = fmax(1.0, sqrt(q2));
        sc[QP_RESY0] = fmax(1.0, sqrt(b2));
        sc[QP_RESZ0] = fmax(1.0, sqrt(h2));
        sc[QP_GAP] = g;
        S.active[0] = 1;
        S.status[0] = 0;
        S.iters[0] = 0;
    }
}

// in place: S.Px = P x, S.ATy = A' y, S.GTz = G' z, S.Ax = A x, S.Gx = G x
__global__ __launch_bounds__(1024) void qp_residual_kernel(QpState S, int it, int maxiters, double abstol, double reltol,
                                                          double feastol) {
    qp_select(S);
    __shared__ double sh[16];
    const int tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    double* sc = S.sc;
    if (S.active[0] == 0) return;
    double f0a = 0.0, f0b = 0.0, r2 = 0.0;
    for (int i = tid; i < n; i += blockDim.x) {
        const double t = S.q[i] + S.Px[i];
        f0a += S.x[i] * t;
        f0b += S.x[i] * S.q[i];
        double r = t;
        if (p > 0) r += S.ATy[i];
        r += S.GTz[i];
        S.rx[i] = r;
        r2 += r * r;
    }
    f0a = lp_block_sum(f0a, sh);
    f0b = lp_block_sum(f0b, sh);
    const double resx = sqrt(lp_block_sum(r2, sh));
    double resy = 0.0, yry = 0.0;
    if (p > 0) {
        double e2 = 0.0, d2 = 0.0;
        for (int i = tid; i < p; i += blockDim.x) {
            const double r = S.Ax[i] - S.b[i];
            S.ry[i] = r;
            e2 += r * r;
            d2 += S.y[i] * r;
        }
        resy = sqrt(lp_block_sum(e2, sh));
        yry = lp_block_sum(d2, sh);
    }
    double z2 = 0.0, zr = 0.0;
    for (int i = tid; i < m; i += blockDim.x) {
        const double r = S.s[i] + S.Gx[i] - S.h[i];
        S.rz[i] = r;
        z2 += r * r;
        zr += S.z[i] * r;
    }
    const double resz = sqrt(lp_block_sum(z2, sh));
    zr = lp_block_sum(zr, sh);
    const double gap = sc[QP_GAP];
    const double f0 = 0.5 * (f0a + f0b);
    const double pcost = f0, dcost = f0 + yry + zr - gap;
    double relgap = 1e300;
    if (pcost < 0.0) relgap = gap / -pcost;
    else if (dcost > 0.0) relgap = gap / dcost;
    const double pres = fmax(resy / sc[QP_RESY0], resz / sc[QP_RESZ0]);
    const double dres = resx / sc[QP_RESX0];
    if (tid == 0) {
        sc[QP_PCOST] = pcost;
        sc[QP_DCOST] = dcost;
        sc[QP_RELGAP] = relgap;
        sc[QP_PRES] = pres;
        sc[QP_DRES] = dres;
        sc[QP_GAP_OUT] = gap;
    }
    const bool conv = pres <= feastol && dres <= feastol && (gap <= abstol || relgap <= reltol);
    __syncthreads();
    if (conv || it == maxiters) {
        qp_store_result(S, conv ? 1 : 2, it);
        return;
    }
    if (tid == 0) atomicAdd(S.nactive, 1);
    if (it == 0) {
        cv_compute_scaling(S, S.s, S.z, S.lmbda, sh);
        __syncthreads();
    }
    for (int i = tid; i < S.ml; i += blockDim.x) S.di[i] = 1.0 / S.d[i];
    cv_ssqr(S, S.lmbdasq, S.lmbda);
    if (tid == 0) {
        sc[QP_MU] = gap / (double)(S.ml + S.nq + (S.ldim - S.lq));   // gap / (dims['l'] + len(dims['q']) + sum(dims['s']))
        sc[QP_SIGMA] = 0.0;
    }
}

// "Terminated (singular KKT matrix)" (:2256-2275)
__global__ __launch_bounds__(1024) void qp_singular_kernel(QpState S, const int* info, int it) {
    if (S.nbatch > 1) info += blockIdx.x;
    qp_select(S);
    if (!S.active[0] || info[0] <= 0) return;
    __syncthreads();
    qp_store_result(S, 3, it);
    if (threadIdx.x == 0) atomicAdd(S.nactive, -1);
}

// right-hand side (:2376-2399): ds = -lmbdasq [- ws3] + sigma mu e; (dx, dy, dz) = -(rx, ry, rz)
__global__ __launch_bounds__(1024) void qp_build_kernel(QpState S, QpBuf D, QpBuf W, int i01, int save) {
    qp_select(S, D); qp_select(S, W); qp_select(S);
    const int tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    const double sigma = (i01 == 0) ? 0.0 : S.sc[QP_SIGMA];
    const double mu = S.sc[QP_MU];
    cv_expand(S, D.s, S.lmbdasq);                   // 's' blocks: diag(lmbdasq_k) (:2386-2391)
    __syncthreads();
    for (int i = tid; i < m; i += blockDim.x) {
        double v = 0.0;
        if (i01 == 1 && S.correction) v -= S.ws3[i];    // coneprog.py:2377-2378
        v -= D.s[i];
        D.s[i] = v;
        D.z[i] = -S.rz[i];
    }
    for (int i = tid; i < n; i += blockDim.x) D.x[i] = -S.rx[i];
    for (int i = tid; i < p; i += blockDim.x) D.y[i] = -S.ry[i];
    __syncthreads();
    cv_add_e(S, D.s, sigma * mu);
    if (save) {
        __syncthreads();
        for (int i = tid; i < m; i += blockDim.x) { W.s[i] = D.s[i]; W.z[i] = D.z[i]; }
        for (int i = tid; i < n; i += blockDim.x) W.x[i] = D.x[i];
        for (int i = tid; i < p; i += blockDim.x) W.y[i] = D.y[i];
    }
}

__global__ __launch_bounds__(1024) void qp_copy_kernel(QpState S, QpBuf dst, QpBuf src) {
    qp_select(S, dst); qp_select(S, src); qp_select(S);
    const int tid = threadIdx.x;
    for (int i = tid; i < S.m; i += blockDim.x) { dst.s[i] = src.s[i]; dst.z[i] = src.z[i]; }
    for (int i = tid; i < S.n; i += blockDim.x) dst.x[i] = src.x[i];
    for (int i = tid; i < S.p; i += blockDim.x) dst.y[i] = src.y[i];
}
__global__ __launch_bounds__(1024) void qp_add_kernel(QpState S, QpBuf dst, QpBuf src) {
    qp_select(S, dst); qp_select(S, src); qp_select(S);
    const int tid = threadIdx.x;
    for (int i = tid; i < S.m; i += blockDim.x) { dst.s[i] += src.s[i]; dst.z[i] += src.z[i]; }
    for (int i = tid; i < S.n; i += blockDim.x) dst.x[i] += src.x[i];
    for (int i = tid; i < S.p; i += blockDim.x) dst.y[i] += src.y[i];
}

// f4_no_ir before the KKT solve (:2303-2309): s := lmbda o\ s; z := z - W's
__global__ __launch_bounds__(1024) void qp_f4pre_kernel(QpState S, QpBuf X) {
    qp_select(S, X); qp_select(S);
    __shared__ double sh[16];
    const int tid = threadIdx.x, m = S.m;
    cv_sinv(S, X.s, S.lmbda, sh);
    __syncthreads();
    for (int i = tid; i < m; i += blockDim.x) S.t1[i] = X.s[i];
    __syncthreads();
    cv_scale(S, S.t1, false, true, sh);             // misc.scale(ws3, W, trans = 'T')
    __syncthreads();
    for (int i = tid; i < m; i += blockDim.x) X.z[i] -= S.t1[i];
}
// ... and after it (:2316): s := s - z
__global__ __launch_bounds__(1024) void qp_f4post_kernel(QpState S, QpBuf X) {
    qp_select(S, X); qp_select(S);
    for (int i = threadIdx.x; i < S.m; i += blockDim.x) X.s[i] -= X.z[i];
}

// res() (:1930-1961), first half: wz3 = W^-1 uz (products with P, A', G', A, G launched by the host in between)
__global__ __launch_bounds__(1024) void qp_res_a_kernel(QpState S, QpBuf U) {
    qp_select(S, U); qp_select(S);
    __shared__ double sh[16];
    for (int i = threadIdx.x; i < S.m; i += blockDim.x) S.wz3[i] = U.z[i];
    __syncthreads();
    cv_scale(S, S.wz3, true, false, sh);            // misc.scale(wz3, W, inverse = 'I')
}
// second half: S.Px = P ux, S.ATy = A' uy, S.GTz = G' wz3, S.Ax = A ux, S.Gx = G ux are in place
__global__ __launch_bounds__(1024) void qp_res_b_kernel(QpState S, QpBuf U, QpBuf V) {
    qp_select(S, U); qp_select(S, V); qp_select(S);
    __shared__ double sh[16];
    const int tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    for (int i = tid; i < n; i += blockDim.x) {
        double v = V.x[i] - S.Px[i];
        if (p > 0) v -= S.ATy[i];
        v -= S.GTz[i];
        V.x[i] = v;
    }
    for (int i = tid; i < p; i += blockDim.x) V.y[i] -= S.Ax[i];
    for (int i = tid; i < m; i += blockDim.x) {
        S.t1[i] = U.s[i];                           // W' us
        S.t2[i] = U.s[i] + U.z[i];                  // lmbda o (uz + us)
    }
    __syncthreads();
    cv_scale(S, S.t1, false, true, sh);             // misc.scale(ws3, W, trans = 'T')
    cv_sprod_diag(S, S.t2, S.lmbda, sh);            // misc.sprod(ws3, lmbda, dims, diag = 'D')
    __syncthreads();
    for (int i = tid; i < m; i += blockDim.x) {
        V.z[i] = V.z[i] - S.Gx[i] - S.t1[i];
        V.s[i] -= S.t2[i];
    }
}

__global__ __launch_bounds__(1024) void qp_step_kernel(QpState S, QpBuf D, int i01) {
    qp_select(S, D); qp_select(S);
    __shared__ double sh[16];
    const int tid = threadIdx.x, m = S.m;
    double* sc = S.sc;
    const double dsdz = lp_dot(D.s, D.z, m, sh);
    if (i01 == 0) {
        for (int i = tid; i < m; i += blockDim.x) S.ws3[i] = D.s[i];
        __syncthreads();
        cv_sprod(S, S.ws3, D.z, sh);
    }
    __syncthreads();
    cv_scale2(S, S.lmbda, D.s, false, sh);
    cv_scale2(S, S.lmbda, D.z, false, sh);
    __syncthreads();
    // i01 == 1: also the eigenvalue decomposition of the 's' blocks of ds, dz (:2427-2437)
    const double ts = (i01 == 0) ? cv_maxstep(S, D.s, sh) : cv_maxstep_sigma(S, D.s, S.sigs, sh);
    const double tz = (i01 == 0) ? cv_maxstep(S, D.z, sh) : cv_maxstep_sigma(S, D.z, S.sigz, sh);
    if (tid == 0) {
        const double t = fmax(0.0, fmax(ts, tz));
        const double step = (t == 0.0) ? 1.0 : fmin(1.0, (i01 == 0 ? 1.0 : 0.99) / t);
        sc[QP_STEP] = step;
        if (i01 == 0) {
            double sg = 1.0 - step + dsdz / sc[QP_GAP] * step * step;
            sg = fmin(1.0, fmax(0.0, sg));
            sc[QP_SIGMA] = sg * sg * sg;
        }
    }
}

__global__ __launch_bounds__(1024) void qp_update_kernel(QpState S, QpBuf D) {
    qp_select(S, D); qp_select(S);
    __shared__ double sh[16];
    const int tid = threadIdx.x;
    if (!S.active[0]) return;
    const double step = S.sc[QP_STEP];
    for (int i = tid; i < S.n; i += blockDim.x) S.x[i] += step * D.x[i];
    for (int i = tid; i < S.p; i += blockDim.x) S.y[i] += step * D.y[i];
    // ('s' blocks: ds, dz hold the eigenvectors Qs, Qz; they become the factors Ls, Lz of the updated variables in the
    // current scaling, :2459-2501)
    for (int i = tid; i < S.lq; i += blockDim.x) {
        D.s[i] *= step;
        D.z[i] *= step;
    }
    __syncthreads();
    cv_add_e(S, D.s, 1.0, false);
    cv_add_e(S, D.z, 1.0, false);
    __syncthreads();
    cv_scale2(S, S.lmbda, D.s, true, sh);
    cv_scale2(S, S.lmbda, D.z, true, sh);
    cv_s_factors(S, S.lmbda, D.s, S.sigs, step);
    cv_s_factors(S, S.lmbda, D.z, S.sigz, step);
    __syncthreads();
    cv_update_scaling(S, S.lmbda, D.s, D.z, sh);
    __syncthreads();
    cv_expand(S, S.s, S.lmbda);
    cv_expand(S, S.z, S.lmbda);
    __syncthreads();
    cv_scale(S, S.s, false, true, sh);
    cv_scale(S, S.z, true, false, sh);
    const double g = lp_dot(S.lmbda, S.lmbda, S.ldim, sh);
    if (tid == 0) S.sc[QP_GAP] = g;
}

// upper triangles of the 's' blocks of a KKT-solve result := lower triangles
__global__ __launch_bounds__(1024) void qp_symm_kernel(QpState S, double* z) { cv_symm(S, z); }

// one workgroup of S.nthreads threads per problem; kernels that run the Jacobi iteration get the dynamic LDS staging area (> 64 KB
// needs the attribute once per kernel)
#define QP1(kernel, ...) hipLaunchKernelGGL(kernel, dim3(S.nbatch > 1 ? S.nbatch : 1), dim3(S.nthreads), 0, st, __VA_ARGS__)
#define QP1J(kernel, ...)                                                                                              \
    do {                                                                                                               \
        static bool attr_done = false;                                                                                 \
        if (S.lds_doubles > 0 && !attr_done) {                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024 - 512);                                                               \
            attr_done = true;                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL(kernel, dim3(S.nbatch > 1 ? S.nbatch : 1), dim3(S.nthreads), sizeof(double) * S.lds_doubles, st,     \
                           __VA_ARGS__);         \
    } while (0)
void qp_launch_symm(const QpState& S, double* z, hipStream_t st) { if (S.ns > 0) QP1(qp_symm_kernel, S, z); }
void qp_launch_unit_scaling(const QpState& S, hipStream_t st) { QP1(qp_unit_scaling_kernel, S); }
void qp_launch_start(const QpState& S, hipStream_t st, int given) { QP1J(qp_start_kernel, S, given); }
void qp_launch_residual(const QpState& S, int it, int maxiters, double abstol, double reltol, double feastol, hipStream_t st) {
    QP1J(qp_residual_kernel, S, it, maxiters, abstol, reltol, feastol);
}
void qp_launch_singular(const QpState& S, const int* d_info, int it, hipStream_t st) { QP1(qp_singular_kernel, S, d_info, it); }
void qp_launch_build(const QpState& S, const QpBuf& D, const QpBuf& W, int i01, int save, hipStream_t st) { QP1(qp_build_kernel, S, D, W, i01, save); }
void qp_launch_copy(const QpState& S, const QpBuf& dst, const QpBuf& src, hipStream_t st) { QP1(qp_copy_kernel, S, dst, src); }
void qp_launch_add(const QpState& S, const QpBuf& dst, const QpBuf& src, hipStream_t st) { QP1(qp_add_kernel, S, dst, src); }
void qp_launch_f4pre(const QpState& S, const QpBuf& X, hipStream_t st) { QP1(qp_f4pre_kernel, S, X); }
void qp_launch_f4post(const QpState& S, const QpBuf& X, hipStream_t st) { QP1(qp_f4post_kernel, S, X); }
void qp_launch_res_a(const QpState& S, const QpBuf& U, hipStream_t st) { QP1(qp_res_a_kernel, S, U); }
void qp_launch_res_b(const QpState& S, const QpBuf& U, const QpBuf& V, hipStream_t st) { QP1(qp_res_b_kernel, S, U, V); }
void qp_launch_step(const QpState& S, const QpBuf& D, int i01, hipStream_t st) { QP1J(qp_step_kernel, S, D, i01); }
void qp_launch_update(const QpState& S, const QpBuf& D, hipStream_t st) { QP1J(qp_update_kernel, S, D); }

}  // namespace mi355kkt

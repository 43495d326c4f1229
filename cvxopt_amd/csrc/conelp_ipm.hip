// Device-resident bookkeeping of the reference conelp loop (self-dual embedding) for
// dims = {'l': ml, 'q': [...]}: one 256-thread workgroup, fixed-order block reductions; second-order cones are walked
// one cone per thread (the BASELINE SOCP class has ~1e3 cones of dimension ~1e1).  Each kernel restates a stretch of
// src/python/coneprog.py, operation for operation:
//     lp_init_primal_kernel   :699-713    s = -s after the first KKT solve, ts
//     lp_init_dual_kernel     :735-836    tz, the "initial point is optimal" test, shifts, tau = kappa = 1, gap
//     lp_residual_kernel      :860-1041   residuals, statistics, the four stopping tests, compute_scaling (misc.py:284-354),
//                                         right-hand side of the extra solve (:1064-1071), th (:1133-1135), lmbdasq, mu
//     lp_scale1_kernel        :1072-1074  (x1, y1, z1) *= dgi
//     lp_build_kernel         :1259-1296  right-hand side (dx, dy, dz, dtau, ds, dkappa), saved for the refinement
//     lp_f6pre_kernel         f6_no_ir :1158-1174      lp_f6post_kernel   f6_no_ir :1187-1203
//     lp_res_a/_b_kernel      res() :596-634 (5x5 residual of the refinement step, :1220-1235);  lp_add_kernel
//     lp_step_kernel          :1299-1331  Mehrotra products, scale2, step to the boundary, sigma
//     lp_update_kernel        :1335-1432 + misc.py:444-464, :503-573   iterate, scaling, tau / kappa update
// and the cone-vector operations sprod, sinv, ssqr, scale2, max_step, scale (src/C/misc_solvers.c:634, :775, :256, :1052,
// :85; misc.py:945) for the 'l' and 'q' blocks.
#include "cone_ops.h"
#include <algorithm>

namespace mi355kkt {

// status codes: 1 optimal, 2 unknown (iteration limit), 3 unknown (singular KKT matrix), 4 primal infeasible,
// 5 dual infeasible
__device__ __forceinline__ void lp_store_result(const LpState& S, int status, int it, double xs, double ys, double ss, double zs) {
    const int tid = threadIdx.x, n = S.n, m = S.m, p = S.p;
    for (int i = tid; i < n; i += blockDim.x) S.x_out[i] = S.x[i] * xs;
    for (int i = tid; i < p; i += blockDim.x) S.y_out[i] = S.y[i] * ys;
    for (int i = tid; i < m; i += blockDim.x) {
        S.s_out[i] = S.s[i] * ss;
        S.z_out[i] = S.z[i] * zs;
    }
    if (tid == 0) {
        S.status[0] = status;
        S.iters[0] = it;
        S.active[0] = 0;
    }
}

// W = I: d = di = 1, v_k = e, beta_k = 1 (coneprog.py:676-688)
__global__ __launch_bounds__(1024) void lp_unit_scaling_kernel(LpState S) {
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

// given: s is the caller's primalstart['s'] (coneprog.py:703-705): no sign flip
__global__ __launch_bounds__(1024) void lp_init_primal_kernel(LpState S, int given) {
    __shared__ double sh[16];
    if (!given)
        for (int i = threadIdx.x; i < S.m; i += blockDim.x) S.s[i] = -S.s[i];
    __syncthreads();
    const double ts = cv_maxstep(S, S.s, sh);
    if (threadIdx.x == 0) S.sc[LP_TS] = ts;
}

// have_primal / have_dual: the caller supplied primalstart / dualstart (coneprog.py:740-836): the "constructed point is optimal"
// exit and both shifts belong to the case where neither was given; with one of them only the OTHER, constructed, vector is shifted
__global__ __launch_bounds__(1024) void lp_init_dual_kernel(LpState S, double abstol, double reltol, int have_primal, int have_dual) {
    __shared__ double sh[16];
    const int tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    double* sc = S.sc;
    const double tz = cv_maxstep(S, S.z, sh);
    const double ns = sqrt(lp_dot(S.s, S.s, m, sh)), nz = sqrt(lp_dot(S.z, S.z, m, sh));
    const double g = lp_dot(S.s, S.z, m, sh), hz = lp_dot(S.h, S.z, m, sh);
    const double cx = lp_dot(S.c, S.x, n, sh);
    const double by = p > 0 ? lp_dot(S.b, S.y, p, sh) : 0.0;
    const double c2 = lp_dot(S.c, S.c, n, sh), h2 = lp_dot(S.h, S.h, m, sh);
    const double b2 = p > 0 ? lp_dot(S.b, S.b, p, sh) : 0.0;
    const double ts = sc[LP_TS];
    const double pcost = cx, dcost = -by - hz;
    double relgap = 1e300;
    if (pcost < 0.0) relgap = g / -pcost;
    else if (dcost > 0.0) relgap = g / dcost;
    const bool init_opt = !have_primal && !have_dual && (ts <= 0.0 && tz <= 0.0 && (g <= abstol || relgap <= reltol));
    if (tid == 0) {
        sc[LP_RESX0] = fmax(1.0, sqrt(c2));
        sc[LP_RESY0] = fmax(1.0, sqrt(b2));
        sc[LP_RESZ0] = fmax(1.0, sqrt(h2));
        sc[LP_TZ] = tz;
        sc[LP_GAP] = g;
        sc[LP_PCOST] = pcost;
        sc[LP_DCOST] = dcost;
        sc[LP_RELGAP] = relgap;
        sc[LP_TAU] = 1.0;
        sc[LP_KAPPA] = 1.0;
        S.active[0] = 1;
        S.status[0] = 0;
        S.iters[0] = 0;
        S.init_optimal[0] = init_opt ? 1 : 0;
    }
    if (init_opt) {                                 // coneprog.py:761-806: the constructed point is already optimal
        __syncthreads();
        lp_store_result(S, 1, 0, 1.0, 1.0, 1.0, 1.0);
        return;
    }
    if (!have_primal && ts >= -1e-8 * fmax(ns, 1.0)) cv_add_e(S, S.s, 1.0 + ts);
    if (!have_dual && tz >= -1e-8 * fmax(nz, 1.0)) cv_add_e(S, S.z, 1.0 + tz);
    __syncthreads();
    const double g2 = lp_dot(S.s, S.z, m, sh);
    if (tid == 0) sc[LP_GAP] = g2;
}

__global__ __launch_bounds__(1024) void lp_residual_kernel(LpState S, int it, int maxiters, double abstol, double reltol,
                                                          double feastol) {
    __shared__ double sh[16];
    const int tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    double* sc = S.sc;
    if (S.active[0] == 0) return;
    const double tau = sc[LP_TAU], kappa = sc[LP_KAPPA], gap = sc[LP_GAP];
    // hrx = -A'y - G'z ; rx = hrx - c tau
    double hx2 = 0.0, rx2 = 0.0, cx = 0.0;
    for (int i = tid; i < n; i += blockDim.x) {
        double hr = 0.0;
        if (p > 0) hr = -S.ATy[i];
        hr -= S.GTz[i];
        const double r = hr - S.c[i] * tau;
        S.rx[i] = r;
        hx2 += hr * hr;
        rx2 += r * r;
        cx += S.c[i] * S.x[i];
    }
    const double hresx = sqrt(lp_block_sum(hx2, sh));
    const double resx = sqrt(lp_block_sum(rx2, sh)) / tau;
    cx = lp_block_sum(cx, sh);
    double hresy = 0.0, resy = 0.0, by = 0.0;
    if (p > 0) {
        double a2 = 0.0, r2 = 0.0, d = 0.0;
        for (int i = tid; i < p; i += blockDim.x) {
            const double hr = S.Ax[i];
            const double r = hr - S.b[i] * tau;
            S.ry[i] = r;
            a2 += hr * hr;
            r2 += r * r;
            d += S.b[i] * S.y[i];
        }
        hresy = sqrt(lp_block_sum(a2, sh));
        resy = sqrt(lp_block_sum(r2, sh)) / tau;
        by = lp_block_sum(d, sh);
    }
    double hz2 = 0.0, rz2 = 0.0, hz = 0.0;
    for (int i = tid; i < m; i += blockDim.x) {
        const double hr = S.s[i] + S.Gx[i];
        const double r = hr - S.h[i] * tau;
        S.rz[i] = r;
        hz2 += hr * hr;
        rz2 += r * r;
        hz += S.h[i] * S.z[i];
    }
    const double hresz = sqrt(lp_block_sum(hz2, sh));
    const double resz = sqrt(lp_block_sum(rz2, sh)) / tau;
    hz = lp_block_sum(hz, sh);
    const double rt = kappa + cx + by + hz;
    const double pcost = cx / tau, dcost = -(by + hz) / tau;
    double relgap = 1e300;
    if (pcost < 0.0) relgap = gap / -pcost;
    else if (dcost > 0.0) relgap = gap / dcost;
    const double resx0 = sc[LP_RESX0], resy0 = sc[LP_RESY0], resz0 = sc[LP_RESZ0];
    const double pres = fmax(resy / resy0, resz / resz0), dres = resx / resx0;
    const bool has_pinf = (hz + by < 0.0), has_dinf = (cx < 0.0);
    const double pinfres = has_pinf ? hresx / resx0 / (-hz - by) : 1e300;
    const double dinfres = has_dinf ? fmax(hresy / resy0, hresz / resz0) / (-cx) : 1e300;
    if (tid == 0) {
        sc[LP_RT] = rt;
        sc[LP_PCOST] = pcost;
        sc[LP_DCOST] = dcost;
        sc[LP_RELGAP] = relgap;
        sc[LP_PRES] = pres;
        sc[LP_DRES] = dres;
        sc[LP_PINFRES] = pinfres;
        sc[LP_DINFRES] = dinfres;
        sc[LP_GAP_OUT] = gap;
    }
    const bool conv = pres <= feastol && dres <= feastol && (gap <= abstol || relgap <= reltol);
    const bool stop_opt = conv || it == maxiters;
    const bool stop_pinf = !stop_opt && has_pinf && pinfres <= feastol;
    const bool stop_dinf = !stop_opt && !stop_pinf && has_dinf && dinfres <= feastol;
    __syncthreads();
    if (stop_opt) {                                // coneprog.py:920-971
        lp_store_result(S, conv ? 1 : 2, it, 1.0 / tau, 1.0 / tau, 1.0 / tau, 1.0 / tau);
        return;
    } else if (stop_pinf) {                        // :973-995
        lp_store_result(S, 4, it, 0.0, 1.0 / (-hz - by), 0.0, 1.0 / (-hz - by));
        return;
    } else if (stop_dinf) {                        // :997-1020
        lp_store_result(S, 5, it, 1.0 / (-cx), 0.0, 1.0 / (-cx), 0.0);
        return;
    }
    if (tid == 0) atomicAdd(S.nactive, 1);
    if (it == 0) {                                 // compute_scaling (misc.py:284-354); dg, lambda_g (:1026-1041)
        cv_compute_scaling(S, S.s, S.z, S.lmbda, sh);
        if (tid == 0) {
            sc[LP_DG] = sqrt(kappa / tau);
            sc[LP_DGI] = sqrt(tau / kappa);
            sc[LP_LG] = sqrt(tau * kappa);
        }
        __syncthreads();
    }
    // scaling for the factorisation; right-hand side of the extra solve (x1, y1, z1) = (-c, b, h); th = W^-T h
    for (int i = tid; i < S.ml; i += blockDim.x) S.di[i] = 1.0 / S.d[i];
    for (int i = tid; i < n; i += blockDim.x) S.x1[i] = -S.c[i];
    for (int i = tid; i < p; i += blockDim.x) S.y1[i] = S.b[i];
    for (int i = tid; i < m; i += blockDim.x) {
        S.z1[i] = S.h[i];
        S.th[i] = S.h[i];
    }
    __syncthreads();
    cv_scale(S, S.th, true, true, sh);              // misc.scale(th, W, trans = 'T', inverse = 'I')
    cv_ssqr(S, S.lmbdasq, S.lmbda);
    const double l2 = lp_dot(S.lmbda, S.lmbda, S.ldim, sh);
    if (tid == 0) {
        const double lg = sc[LP_LG];
        const double nr = sqrt(l2 + lg * lg);       // blas.nrm2(lmbda)**2 / (1 + cdim_diag)
        sc[LP_MU] = nr * nr / (1.0 + S.ldim);
        sc[LP_SIGMA] = 0.0;
    }
}

// "Terminated (singular KKT matrix)" (:1076-1109)
__global__ __launch_bounds__(1024) void lp_singular_kernel(LpState S, const int* info, int it) {
    if (!S.active[0] || info[0] <= 0) return;
    const double tau = S.sc[LP_TAU];
    __syncthreads();
    lp_store_result(S, 3, it, 1.0 / tau, 1.0 / tau, 1.0 / tau, 1.0 / tau);
    if (threadIdx.x == 0) atomicAdd(S.nactive, -1);
}

__global__ __launch_bounds__(1024) void lp_scale1_kernel(LpState S) {
    __shared__ double sh[16];
    const int tid = threadIdx.x;
    const double dgi = S.sc[LP_DGI];
    for (int i = tid; i < S.n; i += blockDim.x) S.x1[i] *= dgi;
    for (int i = tid; i < S.p; i += blockDim.x) S.y1[i] *= dgi;
    for (int i = tid; i < S.m; i += blockDim.x) S.z1[i] *= dgi;
    __syncthreads();
    const double zz = lp_dot(S.z1, S.z1, S.m, sh);
    if (tid == 0) S.sc[LP_Z1Z1] = zz;
}

// right-hand side of the Newton system (:1259-1296), also saved in W for the refinement step
__global__ __launch_bounds__(1024) void lp_build_kernel(LpState S, LpBuf D, LpBuf W, int i01, int save) {
    const int tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    double* sc = S.sc;
    const double sigma = (i01 == 0) ? 0.0 : sc[LP_SIGMA];
    const double mu = sc[LP_MU], lg = sc[LP_LG];
    cv_expand(S, D.s, S.lmbdasq);                   // ds := lmbdasq ('s' blocks: diag(lmbdasq_k), coneprog.py:1264-1273)
    for (int i = tid; i < m; i += blockDim.x) D.z[i] = (1.0 - sigma) * S.rz[i];
    if (i01 == 1) {
        __syncthreads();
        for (int i = tid; i < m; i += blockDim.x) D.s[i] += S.ws3[i];
    }
    for (int i = tid; i < n; i += blockDim.x) D.x[i] = (1.0 - sigma) * S.rx[i];
    for (int i = tid; i < p; i += blockDim.x) D.y[i] = (1.0 - sigma) * S.ry[i];
    __syncthreads();
    if (i01 == 1) cv_add_e(S, D.s, -sigma * mu);
    if (tid == 0) {
        double dk = lg * lg;
        if (i01 == 1) dk += sc[LP_WKAPPA3] - sigma * mu;
        sc[D.ikappa] = dk;
        sc[D.itau] = (1.0 - sigma) * sc[LP_RT];
    }
    if (save) {
        __syncthreads();
        for (int i = tid; i < m; i += blockDim.x) { W.s[i] = D.s[i]; W.z[i] = D.z[i]; }
        for (int i = tid; i < n; i += blockDim.x) W.x[i] = D.x[i];
        for (int i = tid; i < p; i += blockDim.x) W.y[i] = D.y[i];
        if (tid == 0) { sc[W.itau] = sc[D.itau]; sc[W.ikappa] = sc[D.ikappa]; }
    }
}

// dst := src (all six components)
__global__ __launch_bounds__(1024) void lp_copy_kernel(LpState S, LpBuf dst, LpBuf src) {
    const int tid = threadIdx.x;
    for (int i = tid; i < S.m; i += blockDim.x) { dst.s[i] = src.s[i]; dst.z[i] = src.z[i]; }
    for (int i = tid; i < S.n; i += blockDim.x) dst.x[i] = src.x[i];
    for (int i = tid; i < S.p; i += blockDim.x) dst.y[i] = src.y[i];
    if (tid == 0) { S.sc[dst.itau] = S.sc[src.itau]; S.sc[dst.ikappa] = S.sc[src.ikappa]; }
}
// dst += src
__global__ __launch_bounds__(1024) void lp_add_kernel(LpState S, LpBuf dst, LpBuf src) {
    const int tid = threadIdx.x;
    for (int i = tid; i < S.m; i += blockDim.x) { dst.s[i] += src.s[i]; dst.z[i] += src.z[i]; }
    for (int i = tid; i < S.n; i += blockDim.x) dst.x[i] += src.x[i];
    for (int i = tid; i < S.p; i += blockDim.x) dst.y[i] += src.y[i];
    if (tid == 0) { S.sc[dst.itau] += S.sc[src.itau]; S.sc[dst.ikappa] += S.sc[src.ikappa]; }
}

// f6_no_ir, part before the KKT solve (:1158-1174): y := -y; s := -lmbda o\ s; z := -(z + W's)
__global__ __launch_bounds__(1024) void lp_f6pre_kernel(LpState S, LpBuf X) {
    __shared__ double sh[16];
    const int tid = threadIdx.x, m = S.m;
    for (int i = tid; i < S.p; i += blockDim.x) X.y[i] = -X.y[i];
    cv_sinv(S, X.s, S.lmbda, sh);
    __syncthreads();
    for (int i = tid; i < m; i += blockDim.x) {
        const double v = -X.s[i];
        X.s[i] = v;
        S.t1[i] = v;
    }
    __syncthreads();
    cv_scale(S, S.t1, false, true, sh);             // W' s
    __syncthreads();
    for (int i = tid; i < m; i += blockDim.x) X.z[i] = -(X.z[i] + S.t1[i]);
}

// f6_no_ir, part after the KKT solve (:1187-1203)
__global__ __launch_bounds__(1024) void lp_f6post_kernel(LpState S, LpBuf X) {
    __shared__ double sh[16];
    __shared__ double tsh;
    const int tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    double* sc = S.sc;
    const double cdx = lp_dot(S.c, X.x, n, sh);
    const double bdy = p > 0 ? lp_dot(S.b, X.y, p, sh) : 0.0;
    const double thz = lp_dot(S.th, X.z, m, sh);
    if (tid == 0) {
        const double lg = sc[LP_LG], dgi = sc[LP_DGI];
        const double kap = -sc[X.ikappa] / lg;      // kappa[0] := -bkappa / lmbdag
        double t = sc[X.itau] + kap / dgi;
        t = dgi * (t + cdx + bdy + thz) / (1.0 + sc[LP_Z1Z1]);
        sc[X.itau] = t;
        sc[X.ikappa] = kap - t;
        tsh = t;
    }
    __syncthreads();
    const double t = tsh;
    for (int i = tid; i < n; i += blockDim.x) X.x[i] += t * S.x1[i];
    for (int i = tid; i < p; i += blockDim.x) X.y[i] += t * S.y1[i];
    for (int i = tid; i < m; i += blockDim.x) {
        const double zz = X.z[i] + t * S.z1[i];
        X.z[i] = zz;
        X.s[i] -= zz;                               // s := s - z
    }
}

// res() (:596-634), first half: wz3 = W^-1 uz (the products with G', A', G, A are launched by the host in between)
__global__ __launch_bounds__(1024) void lp_res_a_kernel(LpState S, LpBuf U) {
    __shared__ double sh[16];
    for (int i = threadIdx.x; i < S.m; i += blockDim.x) S.wz3[i] = U.z[i];
    __syncthreads();
    cv_scale(S, S.wz3, true, false, sh);            // misc.scale(wz3, W, inverse = 'I')
}
// second half: S.GTz = G' wz3, S.ATy = A' uy, S.Gx = G ux, S.Ax = A ux are in place
__global__ __launch_bounds__(1024) void lp_res_b_kernel(LpState S, LpBuf U, LpBuf V) {
    __shared__ double sh[16];
    const int tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    double* sc = S.sc;
    const double dg = sc[LP_DG], lg = sc[LP_LG];
    const double utau = sc[U.itau], ukappa = sc[U.ikappa];
    const double cux = lp_dot(S.c, U.x, n, sh);
    const double buy = p > 0 ? lp_dot(S.b, U.y, p, sh) : 0.0;
    const double hw = lp_dot(S.h, S.wz3, m, sh);
    for (int i = tid; i < n; i += blockDim.x) {
        double v = V.x[i];
        if (p > 0) v -= S.ATy[i];
        v -= S.GTz[i];
        v -= S.c[i] * (utau / dg);
        V.x[i] = v;
    }
    for (int i = tid; i < p; i += blockDim.x) V.y[i] = V.y[i] + S.Ax[i] - S.b[i] * (utau / dg);
    for (int i = tid; i < m; i += blockDim.x) {
        S.t1[i] = U.s[i];                           // W' us
        S.t2[i] = U.s[i] + U.z[i];                  // lmbda o (uz + us)
    }
    __syncthreads();
    cv_scale(S, S.t1, false, true, sh);             // misc.scale(ws3, W, trans = 'T')
    cv_sprod_diag(S, S.t2, S.lmbda, sh);            // misc.sprod(ws3, lmbda, dims, diag = 'D')
    __syncthreads();
    for (int i = tid; i < m; i += blockDim.x) {
        V.z[i] = V.z[i] + S.Gx[i] - S.h[i] * (utau / dg) + S.t1[i];
        V.s[i] += S.t2[i];
    }
    if (tid == 0) {
        sc[V.itau] += dg * ukappa + cux + buy + hw;
        sc[V.ikappa] += lg * (utau + ukappa);
    }
}

// Mehrotra products, scale2, step to the boundary, sigma (:1299-1331)
__global__ __launch_bounds__(1024) void lp_step_kernel(LpState S, LpBuf D, int i01) {
    __shared__ double sh[16];
    const int tid = threadIdx.x, m = S.m;
    double* sc = S.sc;
    if (i01 == 0) {
        for (int i = tid; i < m; i += blockDim.x) S.ws3[i] = D.s[i];
        __syncthreads();
        cv_sprod(S, S.ws3, D.z, sh);
    }
    __syncthreads();
    cv_scale2(S, S.lmbda, D.s, false, sh);
    cv_scale2(S, S.lmbda, D.z, false, sh);
    __syncthreads();
    // i01 == 1: also the eigenvalue decomposition of the 's' blocks of ds, dz (eigenvectors in place, eigenvalues in
    // sigs, sigz), coneprog.py:1308-1318
    const double ts = (i01 == 0) ? cv_maxstep(S, D.s, sh) : cv_maxstep_sigma(S, D.s, S.sigs, sh);
    const double tz = (i01 == 0) ? cv_maxstep(S, D.z, sh) : cv_maxstep_sigma(S, D.z, S.sigz, sh);
    if (tid == 0) {
        const double lg = sc[LP_LG];
        const double dtau = sc[D.itau], dkappa = sc[D.ikappa];
        if (i01 == 0) sc[LP_WKAPPA3] = dtau * dkappa;
        const double tt = -dtau / lg, tk = -dkappa / lg;
        const double tmax = fmax(fmax(0.0, fmax(ts, tz)), fmax(tt, tk));
        const double step = (tmax == 0.0) ? 1.0 : fmin(1.0, (i01 == 0 ? 1.0 : 0.99) / tmax);
        sc[LP_TT] = tt;
        sc[LP_TK] = tk;
        sc[LP_STEP] = step;
        if (i01 == 0) {
            const double om = 1.0 - step;
            sc[LP_SIGMA] = om * om * om;            // (1 - step) ** EXPON, EXPON = 3
        }
    }
}

__global__ __launch_bounds__(1024) void lp_update_kernel(LpState S, LpBuf D) {
    __shared__ double sh[16];
    const int tid = threadIdx.x, n = S.n, p = S.p;
    if (!S.active[0]) return;
    double* sc = S.sc;
    const double step = sc[LP_STEP];
    for (int i = tid; i < n; i += blockDim.x) S.x[i] += step * D.x[i];
    for (int i = tid; i < p; i += blockDim.x) S.y[i] += step * D.y[i];
    // ds := e + step ds, dz := e + step dz; then H(lambda)^{-1/2}: the updated variables in the current scaling
    // ('s' blocks: ds, dz hold the eigenvectors Qs, Qz; they become the factors Ls, Lz of the updated variables in the
    // current scaling, coneprog.py:1348-1395)
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
    if (tid == 0) {
        const double tt = sc[LP_TT], tk = sc[LP_TK];
        const double dg = sc[LP_DG] * (sqrt(1.0 - step * tk) / sqrt(1.0 - step * tt));
        const double dgi = 1.0 / dg;
        const double lg = sc[LP_LG] * (sqrt(1.0 - step * tt) * sqrt(1.0 - step * tk));
        sc[LP_DG] = dg;
        sc[LP_DGI] = dgi;
        sc[LP_LG] = lg;
        sc[LP_KAPPA] = lg / dgi;
        sc[LP_TAU] = lg * dgi;
    }
    // unscale: s = W' lmbda, z = W^-1 lmbda
    cv_expand(S, S.s, S.lmbda);
    cv_expand(S, S.z, S.lmbda);
    __syncthreads();
    cv_scale(S, S.s, false, true, sh);
    cv_scale(S, S.z, true, false, sh);
    const double g = lp_dot(S.lmbda, S.lmbda, S.ldim, sh);
    if (tid == 0) {
        const double r = sqrt(g) / sc[LP_TAU];
        sc[LP_GAP] = r * r;
    }
}

// upper triangles of the 's' blocks of a KKT-solve result := lower triangles
__global__ __launch_bounds__(1024) void lp_symm_kernel(LpState S, double* z) { cv_symm(S, z); }

// one workgroup of S.nthreads threads; kernels that run the Jacobi iteration get the dynamic LDS staging area (> 64 KB
// needs the attribute once per kernel)
// One 's'-block operation of cone_ops_s.h on the device, with the workgroup team (1024 threads) or one wave team (64): the
// device twin of mi355kkt_debug_sdp_op_host for the GPU unit tests.  w: 3 m^2 + jw scratch; out[0]: return value.
__global__ __launch_bounds__(1024) void sdp_op_debug_kernel(int op, int m, int arg, int wave_team, int lds_doubles, double* x, double* y,
                                                            double* r, double* rti, double* lam, double* w, double* out) {
    __shared__ double sh[16];
    const size_t mm = (size_t)m * m;
    double *T1 = w, *T2 = T1 + mm, *T3 = T2 + mm, *jw = T3 + mm;
    const bool inverse = arg & 1, trans = arg & 2;
    auto run = [&](const auto& par) {
        double ret = 0.0;
        switch (op) {
            case 0: s_scale_blk(par, x, inverse ? rti : r, m, trans == inverse, T1); break;
            case 1: s_sprod_blk(par, x, y, m, T1); break;
            case 2: s_sprod_diag_blk(par, x, lam, m, inverse); break;
            case 3: s_scale2_blk(par, lam, x, m, inverse); break;
            case 4: ret = s_min_eig_blk(par, x, m, T1, T2, jw); break;
            case 5: s_eig_blk(par, x, lam, m, T1, T2, jw); break;
            case 6: ret = s_compute_scaling_blk(par, x, y, r, rti, lam, m, T1, T2, T3, jw); break;
            case 7: s_update_scaling_blk(par, x, y, r, rti, lam, m, T1, T2, jw); break;
            case 8: ret = s_potrf(par, x, m); break;
        }
        if (par.tid() == 0) out[0] = ret;
    };
    if (wave_team) {
        if (threadIdx.x < 64) run(ParWave{jw});
    } else {
        run(ParWG{sh, lds_doubles, jw});
    }
}
int sdp_op_debug_launch(int op, int m, int arg, int wave_team, double* x, double* y, double* r, double* rti, double* lam, double* w,
                        double* out, hipStream_t st) {
    const int lds_doubles = wave_team ? 0 : (int)std::min<size_t>(20416, 2 * (size_t)m * m);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sdp_op_debug_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024 - 512);
        attr_done = true;
    }
    hipLaunchKernelGGL(sdp_op_debug_kernel, dim3(1), dim3(wave_team ? 64 : 1024), sizeof(double) * lds_doubles, st, op, m, arg, wave_team,
                       lds_doubles, x, y, r, rti, lam, w, out);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

#define LP1(kernel, ...) hipLaunchKernelGGL(kernel, dim3(1), dim3(S.nthreads), 0, st, __VA_ARGS__)
#define LP1J(kernel, ...)                                                                                              \
    do {                                                                                                               \
        static bool attr_done = false;                                                                                 \
        if (S.lds_doubles > 0 && !attr_done) {                                                                         \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024 - 512);                                                               \
            attr_done = true;                                                                                          \
        }                                                                                                              \
        hipLaunchKernelGGL(kernel, dim3(1), dim3(S.nthreads), sizeof(double) * S.lds_doubles, st, __VA_ARGS__);         \
    } while (0)
void lp_launch_symm(const LpState& S, double* z, hipStream_t st) { if (S.ns > 0) LP1(lp_symm_kernel, S, z); }
void lp_launch_unit_scaling(const LpState& S, hipStream_t st) { LP1(lp_unit_scaling_kernel, S); }
void lp_launch_init_primal(const LpState& S, hipStream_t st, int given) { LP1J(lp_init_primal_kernel, S, given); }
void lp_launch_init_dual(const LpState& S, double abstol, double reltol, hipStream_t st, int have_primal, int have_dual) {
    LP1J(lp_init_dual_kernel, S, abstol, reltol, have_primal, have_dual);
}
void lp_launch_residual(const LpState& S, int it, int maxiters, double abstol, double reltol, double feastol, hipStream_t st) {
    LP1J(lp_residual_kernel, S, it, maxiters, abstol, reltol, feastol);
}
void lp_launch_singular(const LpState& S, const int* d_info, int it, hipStream_t st) { LP1(lp_singular_kernel, S, d_info, it); }
void lp_launch_scale1(const LpState& S, hipStream_t st) { LP1(lp_scale1_kernel, S); }
void lp_launch_build(const LpState& S, const LpBuf& D, const LpBuf& W, int i01, int save, hipStream_t st) { LP1(lp_build_kernel, S, D, W, i01, save); }
void lp_launch_copy(const LpState& S, const LpBuf& dst, const LpBuf& src, hipStream_t st) { LP1(lp_copy_kernel, S, dst, src); }
void lp_launch_add(const LpState& S, const LpBuf& dst, const LpBuf& src, hipStream_t st) { LP1(lp_add_kernel, S, dst, src); }
void lp_launch_f6pre(const LpState& S, const LpBuf& X, hipStream_t st) { LP1(lp_f6pre_kernel, S, X); }
void lp_launch_f6post(const LpState& S, const LpBuf& X, hipStream_t st) { LP1(lp_f6post_kernel, S, X); }
void lp_launch_res_a(const LpState& S, const LpBuf& U, hipStream_t st) { LP1(lp_res_a_kernel, S, U); }
void lp_launch_res_b(const LpState& S, const LpBuf& U, const LpBuf& V, hipStream_t st) { LP1(lp_res_b_kernel, S, U, V); }
void lp_launch_step(const LpState& S, const LpBuf& D, int i01, hipStream_t st) { LP1J(lp_step_kernel, S, D, i01); }
void lp_launch_update(const LpState& S, const LpBuf& D, hipStream_t st) { LP1J(lp_update_kernel, S, D); }

}  // namespace mi355kkt

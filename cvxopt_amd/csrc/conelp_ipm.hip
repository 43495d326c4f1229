// Device-resident bookkeeping of the reference conelp loop (self-dual embedding) for
// dims = {'l': m}: one workgroup per problem, fixed-order block reductions.  Each kernel restates a stretch of
// src/python/coneprog.py, operation for operation:
//     lp_init_primal_kernel   :699-713    s = -s after the first KKT solve, ts
//     lp_init_dual_kernel     :735-836    tz, the "initial point is optimal" test, shifts, tau = kappa = 1, gap
//     lp_residual_kernel      :860-1041   residuals, statistics, the four stopping tests, compute_scaling (misc.py:284-287),
//                                         right-hand side of the extra solve (:1064-1071), th (:1133-1135), mu
//     lp_scale1_kernel        :1072-1074  (x1, y1, z1) *= dgi
//     lp_rhs_kernel           :1259-1296 + f6_no_ir :1158-1174   right-hand sides
//     lp_post_kernel          f6_no_ir :1187-1203, :1299-1331    combination with (x1, y1, z1), step length, sigma
//     lp_update_kernel        :1335-1432 + misc.py:444-464       iterate, scaling, tau / kappa update
// Refinement is 0 for the LP cone (coneprog.py:551-554), so f6 == f6_no_ir.
#include "kkt_common.h"

namespace mi355kkt {

__device__ __forceinline__ double lp_block_sum(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ double lp_block_max(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}
__device__ __forceinline__ double lp_dot(const double* a, const double* b, int n, double* sh) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += a[i] * b[i];
    return lp_block_sum(v, sh);
}

// status codes: 1 optimal, 2 unknown (iteration limit), 3 unknown (singular KKT matrix), 4 primal infeasible,
// 5 dual infeasible
__device__ __forceinline__ void lp_store_result(const LpState& S, int b, int status, int it, double xs, double ys, double ss,
                                                double zs) {
    const int tid = threadIdx.x, n = S.n, m = S.m, p = S.p;
    for (int i = tid; i < n; i += 256) S.x_out[(int64_t)b * n + i] = S.x[(int64_t)b * n + i] * xs;
    for (int i = tid; i < p; i += 256) S.y_out[(int64_t)b * p + i] = S.y[(int64_t)b * p + i] * ys;
    for (int i = tid; i < m; i += 256) {
        S.s_out[(int64_t)b * m + i] = S.s[(int64_t)b * m + i] * ss;
        S.z_out[(int64_t)b * m + i] = S.z[(int64_t)b * m + i] * zs;
    }
    if (tid == 0) {
        S.status[b] = status;
        S.iters[b] = it;
        S.active[b] = 0;
    }
}

__global__ __launch_bounds__(256) void lp_init_primal_kernel(LpState S) {
    __shared__ double sh[4];
    const int b = blockIdx.x, tid = threadIdx.x, m = S.m;
    double* s = S.s + (int64_t)b * m;
    double ts = -1e300;
    for (int i = tid; i < m; i += 256) {
        const double v = -s[i];
        s[i] = v;
        ts = fmax(ts, -v);
    }
    ts = lp_block_max(ts, sh);
    if (tid == 0) S.sc[b * LP_NSC + LP_TS] = ts;
}

__global__ __launch_bounds__(256) void lp_init_dual_kernel(LpState S, double abstol, double reltol) {
    __shared__ double sh[4];
    const int b = blockIdx.x, tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    double* sc = S.sc + b * LP_NSC;
    double* s = S.s + (int64_t)b * m;
    double* z = S.z + (int64_t)b * m;
    const double* h = S.h + (int64_t)b * m;
    const double* c = S.c + (int64_t)b * n;
    const double* x = S.x + (int64_t)b * n;
    double tz = -1e300, ns = 0.0, nz = 0.0, g = 0.0, hz = 0.0;
    for (int i = tid; i < m; i += 256) {
        tz = fmax(tz, -z[i]);
        ns += s[i] * s[i];
        nz += z[i] * z[i];
        g += s[i] * z[i];
        hz += h[i] * z[i];
    }
    tz = lp_block_max(tz, sh);
    ns = sqrt(lp_block_sum(ns, sh));
    nz = sqrt(lp_block_sum(nz, sh));
    g = lp_block_sum(g, sh);
    hz = lp_block_sum(hz, sh);
    const double cx = lp_dot(c, x, n, sh);
    const double by = p > 0 ? lp_dot(S.b + (int64_t)b * p, S.y + (int64_t)b * p, p, sh) : 0.0;
    const double c2 = lp_dot(c, c, n, sh), h2 = lp_dot(h, h, m, sh);
    const double b2 = p > 0 ? lp_dot(S.b + (int64_t)b * p, S.b + (int64_t)b * p, p, sh) : 0.0;
    const double ts = sc[LP_TS];
    const double pcost = cx, dcost = -by - hz;
    double relgap = 1e300;
    if (pcost < 0.0) relgap = g / -pcost;
    else if (dcost > 0.0) relgap = g / dcost;
    const bool init_opt = (ts <= 0.0 && tz <= 0.0 && (g <= abstol || relgap <= reltol));
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
        S.active[b] = 1;
        S.status[b] = 0;
        S.iters[b] = 0;
        S.init_optimal[b] = init_opt ? 1 : 0;
    }
    if (init_opt) {                                 // coneprog.py:761-806: the constructed point is already optimal
        __syncthreads();
        lp_store_result(S, b, 1, 0, 1.0, 1.0, 1.0, 1.0);
        return;
    }
    const double as = (ts >= -1e-8 * fmax(ns, 1.0)) ? 1.0 + ts : 0.0;
    const double az = (tz >= -1e-8 * fmax(nz, 1.0)) ? 1.0 + tz : 0.0;
    double g2 = 0.0;
    for (int i = tid; i < m; i += 256) {
        const double si = s[i] + as, zi = z[i] + az;
        s[i] = si;
        z[i] = zi;
        g2 += si * zi;
    }
    g2 = lp_block_sum(g2, sh);
    if (tid == 0) sc[LP_GAP] = g2;
}

__global__ __launch_bounds__(256) void lp_residual_kernel(LpState S, int it, int maxiters, double abstol, double reltol,
                                                          double feastol) {
    __shared__ double sh[4];
    const int b = blockIdx.x, tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    double* sc = S.sc + b * LP_NSC;
    const bool act = S.active[b] != 0;
    const double tau = sc[LP_TAU], kappa = sc[LP_KAPPA], gap = sc[LP_GAP];
    const double* c = S.c + (int64_t)b * n;
    const double* x = S.x + (int64_t)b * n;
    const double* gtz = S.GTz + (int64_t)b * n;
    const double* aty = S.ATy + (int64_t)b * n;
    double* rx = S.rx + (int64_t)b * n;
    // hrx = -A'y - G'z ; rx = hrx - c tau
    double hx2 = 0.0, rx2 = 0.0, cx = 0.0;
    for (int i = tid; i < n; i += 256) {
        double hr = 0.0;
        if (p > 0) hr = -aty[i];
        hr -= gtz[i];
        const double r = hr - c[i] * tau;
        rx[i] = r;
        hx2 += hr * hr;
        rx2 += r * r;
        cx += c[i] * x[i];
    }
    const double hresx = sqrt(lp_block_sum(hx2, sh));
    const double resx = sqrt(lp_block_sum(rx2, sh)) / tau;
    cx = lp_block_sum(cx, sh);
    double hresy = 0.0, resy = 0.0, by = 0.0;
    if (p > 0) {
        const double* bb = S.b + (int64_t)b * p;
        const double* y = S.y + (int64_t)b * p;
        const double* ax = S.Ax + (int64_t)b * p;
        double* ry = S.ry + (int64_t)b * p;
        double a2 = 0.0, r2 = 0.0, d = 0.0;
        for (int i = tid; i < p; i += 256) {
            const double hr = ax[i];
            const double r = hr - bb[i] * tau;
            ry[i] = r;
            a2 += hr * hr;
            r2 += r * r;
            d += bb[i] * y[i];
        }
        hresy = sqrt(lp_block_sum(a2, sh));
        resy = sqrt(lp_block_sum(r2, sh)) / tau;
        by = lp_block_sum(d, sh);
    }
    const double* h = S.h + (int64_t)b * m;
    double* s = S.s + (int64_t)b * m;
    double* z = S.z + (int64_t)b * m;
    const double* gx = S.Gx + (int64_t)b * m;
    double* rz = S.rz + (int64_t)b * m;
    double hz2 = 0.0, rz2 = 0.0, hz = 0.0;
    for (int i = tid; i < m; i += 256) {
        const double hr = s[i] + gx[i];
        const double r = hr - h[i] * tau;
        rz[i] = r;
        hz2 += hr * hr;
        rz2 += r * r;
        hz += h[i] * z[i];
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
    if (act && tid == 0) {
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
    if (!act) {
        for (int i = tid; i < m; i += 256) S.di[(int64_t)b * m + i] = 1.0;
        return;
    }
    const bool conv = pres <= feastol && dres <= feastol && (gap <= abstol || relgap <= reltol);
    const bool stop_opt = conv || it == maxiters;
    const bool stop_pinf = !stop_opt && has_pinf && pinfres <= feastol;
    const bool stop_dinf = !stop_opt && !stop_pinf && has_dinf && dinfres <= feastol;
    __syncthreads();
    if (stop_opt) {                                // coneprog.py:920-971
        lp_store_result(S, b, conv ? 1 : 2, it, 1.0 / tau, 1.0 / tau, 1.0 / tau, 1.0 / tau);
    } else if (stop_pinf) {                        // :973-995
        lp_store_result(S, b, 4, it, 0.0, 1.0 / (-hz - by), 0.0, 1.0 / (-hz - by));
    } else if (stop_dinf) {                        // :997-1020
        lp_store_result(S, b, 5, it, 1.0 / (-cx), 0.0, 1.0 / (-cx), 0.0);
    } else if (tid == 0) {
        atomicAdd(S.nactive, 1);
    }
    double* d = S.d + (int64_t)b * m;
    double* lm = S.lmbda + (int64_t)b * m;
    double* di = S.di + (int64_t)b * m;
    if (stop_opt || stop_pinf || stop_dinf) {
        for (int i = tid; i < m; i += 256) di[i] = 1.0;
        return;
    }
    if (it == 0) {                                 // compute_scaling, 'l' block; dg, lambda_g (:1026-1041)
        for (int i = tid; i < m; i += 256) {
            d[i] = sqrt(s[i] / z[i]);
            lm[i] = sqrt(s[i] * z[i]);
        }
        if (tid == 0) {
            sc[LP_DG] = sqrt(kappa / tau);
            sc[LP_DGI] = sqrt(tau / kappa);
            sc[LP_LG] = sqrt(tau * kappa);
        }
        __syncthreads();
    }
    // right-hand side of the extra solve (x1, y1, z1) = (-c, b, h); th = W^-T h; mu
    double* x1 = S.x1 + (int64_t)b * n;
    for (int i = tid; i < n; i += 256) x1[i] = -c[i];
    for (int i = tid; i < p; i += 256) S.y1[(int64_t)b * p + i] = S.b[(int64_t)b * p + i];
    double l2 = 0.0;
    for (int i = tid; i < m; i += 256) {
        const double dii = 1.0 / d[i];
        di[i] = dii;
        S.z1[(int64_t)b * m + i] = h[i];
        S.th[(int64_t)b * m + i] = h[i] * dii;
        l2 += lm[i] * lm[i];
    }
    l2 = lp_block_sum(l2, sh);
    if (tid == 0) {
        const double lg = sc[LP_LG];
        const double nr = sqrt(l2 + lg * lg);       // blas.nrm2(lmbda)**2 / (1 + cdim_diag)
        sc[LP_MU] = nr * nr / (1.0 + m);
        sc[LP_SIGMA] = 0.0;
    }
}

// problems whose factorisation failed: "Terminated (singular KKT matrix)" (:1076-1109)
__global__ __launch_bounds__(256) void lp_singular_kernel(LpState S, const int* info, int it) {
    const int b = blockIdx.x;
    if (!S.active[b] || info[b] <= 0) return;
    const double tau = S.sc[b * LP_NSC + LP_TAU];
    __syncthreads();
    lp_store_result(S, b, 3, it, 1.0 / tau, 1.0 / tau, 1.0 / tau, 1.0 / tau);
    if (threadIdx.x == 0) atomicAdd(S.nactive, -1);
}

__global__ __launch_bounds__(256) void lp_scale1_kernel(LpState S) {
    __shared__ double sh[4];
    const int b = blockIdx.x, tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    double* sc = S.sc + b * LP_NSC;
    const double dgi = sc[LP_DGI];
    for (int i = tid; i < n; i += 256) S.x1[(int64_t)b * n + i] *= dgi;
    for (int i = tid; i < p; i += 256) S.y1[(int64_t)b * p + i] *= dgi;
    double zz = 0.0;
    for (int i = tid; i < m; i += 256) {
        const double v = S.z1[(int64_t)b * m + i] * dgi;
        S.z1[(int64_t)b * m + i] = v;
        zz += v * v;
    }
    zz = lp_block_sum(zz, sh);
    if (tid == 0) sc[LP_Z1Z1] = zz;
}

__global__ __launch_bounds__(256) void lp_rhs_kernel(LpState S, int i01) {
    const int b = blockIdx.x, tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    double* sc = S.sc + b * LP_NSC;
    const double sigma = (i01 == 0) ? 0.0 : sc[LP_SIGMA];
    const double mu = sc[LP_MU], lg = sc[LP_LG];
    const double* lm = S.lmbda + (int64_t)b * m;
    const double* d = S.d + (int64_t)b * m;
    const double* rz = S.rz + (int64_t)b * m;
    const double* ws3 = S.ws3 + (int64_t)b * m;
    double* ds = S.ds + (int64_t)b * m;
    double* dz = S.dz + (int64_t)b * m;
    for (int i = tid; i < m; i += 256) {
        double v = lm[i] * lm[i];                   // ds = lmbdasq (+ ws3 - sigma mu)
        if (i01 == 1) v += ws3[i] - sigma * mu;
        v = -(v / lm[i]);                           // f6_no_ir: s := -lmbda o\ s
        ds[i] = v;
        dz[i] = -((1.0 - sigma) * rz[i] + d[i] * v);   // z := -(z + W's)
    }
    for (int i = tid; i < n; i += 256) S.dx[(int64_t)b * n + i] = (1.0 - sigma) * S.rx[(int64_t)b * n + i];
    for (int i = tid; i < p; i += 256) S.dy[(int64_t)b * p + i] = -((1.0 - sigma) * S.ry[(int64_t)b * p + i]);
    if (tid == 0) {
        double dk = lg * lg;
        if (i01 == 1) dk += sc[LP_WKAPPA3] - sigma * mu;
        sc[LP_DKAPPA] = dk;
        sc[LP_DTAU] = (1.0 - sigma) * sc[LP_RT];
    }
}

__global__ __launch_bounds__(256) void lp_post_kernel(LpState S, int i01) {
    __shared__ double sh[4];
    __shared__ double tsh;
    const int b = blockIdx.x, tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    double* sc = S.sc + b * LP_NSC;
    double* dx = S.dx + (int64_t)b * n;
    double* dz = S.dz + (int64_t)b * m;
    double* ds = S.ds + (int64_t)b * m;
    const double cdx = lp_dot(S.c + (int64_t)b * n, dx, n, sh);
    const double bdy = p > 0 ? lp_dot(S.b + (int64_t)b * p, S.dy + (int64_t)b * p, p, sh) : 0.0;
    const double thz = lp_dot(S.th + (int64_t)b * m, dz, m, sh);
    if (tid == 0) {
        const double lg = sc[LP_LG], dgi = sc[LP_DGI];
        const double kap = -sc[LP_DKAPPA] / lg;     // kappa[0] := -bkappa / lmbdag
        double t = sc[LP_DTAU] + kap / dgi;
        t = dgi * (t + cdx + bdy + thz) / (1.0 + sc[LP_Z1Z1]);
        sc[LP_DTAU] = t;
        sc[LP_DKAPPA] = kap - t;
        tsh = t;
    }
    __syncthreads();
    const double t = tsh;
    for (int i = tid; i < n; i += 256) dx[i] += t * S.x1[(int64_t)b * n + i];
    for (int i = tid; i < p; i += 256) S.dy[(int64_t)b * p + i] += t * S.y1[(int64_t)b * p + i];
    const double* lm = S.lmbda + (int64_t)b * m;
    double* ws3 = S.ws3 + (int64_t)b * m;
    double tm = 0.0;
    for (int i = tid; i < m; i += 256) {
        const double zz = dz[i] + t * S.z1[(int64_t)b * m + i];
        const double ss = ds[i] - zz;               // s := s - z
        if (i01 == 0) ws3[i] = ss * zz;             // ds o dz for the Mehrotra correction
        const double sl = ss / lm[i], zl = zz / lm[i];   // scale2
        ds[i] = sl;
        dz[i] = zl;
        tm = fmax(tm, fmax(-sl, -zl));
    }
    tm = lp_block_max(tm, sh);
    if (tid == 0) {
        const double lg = sc[LP_LG];
        const double dtau = sc[LP_DTAU], dkappa = sc[LP_DKAPPA];
        if (i01 == 0) sc[LP_WKAPPA3] = dtau * dkappa;
        const double tt = -dtau / lg, tk = -dkappa / lg;
        const double tmax = fmax(fmax(0.0, tm), fmax(tt, tk));
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

__global__ __launch_bounds__(256) void lp_update_kernel(LpState S) {
    __shared__ double sh[4];
    const int b = blockIdx.x, tid = threadIdx.x, m = S.m, n = S.n, p = S.p;
    if (!S.active[b]) return;
    double* sc = S.sc + b * LP_NSC;
    const double step = sc[LP_STEP];
    for (int i = tid; i < n; i += 256) S.x[(int64_t)b * n + i] += step * S.dx[(int64_t)b * n + i];
    for (int i = tid; i < p; i += 256) S.y[(int64_t)b * p + i] += step * S.dy[(int64_t)b * p + i];
    double* lm = S.lmbda + (int64_t)b * m;
    double* d = S.d + (int64_t)b * m;
    double* s = S.s + (int64_t)b * m;
    double* z = S.z + (int64_t)b * m;
    const double* ds = S.ds + (int64_t)b * m;
    const double* dz = S.dz + (int64_t)b * m;
    double g = 0.0;
    for (int i = tid; i < m; i += 256) {
        const double l = lm[i];
        const double a = sqrt((1.0 + step * ds[i]) * l);
        const double c = sqrt((1.0 + step * dz[i]) * l);
        const double dn = d[i] * a / c;
        const double ln = a * c;
        d[i] = dn;
        lm[i] = ln;
        s[i] = dn * ln;
        z[i] = ln / dn;
        g += ln * ln;
    }
    g = lp_block_sum(g, sh);
    if (tid == 0) {
        const double tt = sc[LP_TT], tk = sc[LP_TK];
        const double dg = sc[LP_DG] * (sqrt(1.0 - step * tk) / sqrt(1.0 - step * tt));
        const double dgi = 1.0 / dg;
        const double lg = sc[LP_LG] * (sqrt(1.0 - step * tt) * sqrt(1.0 - step * tk));
        sc[LP_DG] = dg;
        sc[LP_DGI] = dgi;
        sc[LP_LG] = lg;
        const double kappa = lg / dgi, tau = lg * dgi;
        sc[LP_KAPPA] = kappa;
        sc[LP_TAU] = tau;
        const double r = sqrt(g) / tau;
        sc[LP_GAP] = r * r;
    }
}

void lp_launch_init_primal(const LpState& S, int B, hipStream_t st) { hipLaunchKernelGGL(lp_init_primal_kernel, dim3(B), dim3(256), 0, st, S); }
void lp_launch_init_dual(const LpState& S, int B, double abstol, double reltol, hipStream_t st) {
    hipLaunchKernelGGL(lp_init_dual_kernel, dim3(B), dim3(256), 0, st, S, abstol, reltol);
}
void lp_launch_residual(const LpState& S, int B, int it, int maxiters, double abstol, double reltol, double feastol,
                        hipStream_t st) {
    hipLaunchKernelGGL(lp_residual_kernel, dim3(B), dim3(256), 0, st, S, it, maxiters, abstol, reltol, feastol);
}
void lp_launch_singular(const LpState& S, int B, const int* d_info, int it, hipStream_t st) {
    hipLaunchKernelGGL(lp_singular_kernel, dim3(B), dim3(256), 0, st, S, d_info, it);
}
void lp_launch_scale1(const LpState& S, int B, hipStream_t st) { hipLaunchKernelGGL(lp_scale1_kernel, dim3(B), dim3(256), 0, st, S); }
void lp_launch_rhs(const LpState& S, int B, int i01, hipStream_t st) { hipLaunchKernelGGL(lp_rhs_kernel, dim3(B), dim3(256), 0, st, S, i01); }
void lp_launch_post(const LpState& S, int B, int i01, hipStream_t st) { hipLaunchKernelGGL(lp_post_kernel, dim3(B), dim3(256), 0, st, S, i01); }
void lp_launch_update(const LpState& S, int B, hipStream_t st) { hipLaunchKernelGGL(lp_update_kernel, dim3(B), dim3(256), 0, st, S); }

}  // namespace mi355kkt

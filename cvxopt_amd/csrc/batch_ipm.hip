// Device-resident bookkeeping of the LP-cone coneqp interior-point loop for a batch of independent problems
// (SURVEY.md 8(f) row 1).  Each kernel restates a stretch of the reference driver, operation for operation, for
// every problem of the batch (one workgroup per problem, block reductions for the dot products / maxima):
//     ipm_start_kernel      coneprog.py:2083-2106   s = -z, shifts into the cone interior, gap
//     ipm_residual_kernel   coneprog.py:2170-2234   residuals, costs, stopping test (+ misc.py:284-287 at iteration 0)
//     ipm_rhs_kernel        coneprog.py:2376-2399 + f4_no_ir :2303-2309   right-hand side of the two KKT solves
//     ipm_post_kernel       coneprog.py:2316, :2423-2456   ds, step to the boundary, sigma
//     ipm_update_kernel     coneprog.py:2459-2547 + misc.py:444-464   iterate + scaling update
// The KKT work in between is the batched factor / solve of capi.hip; nothing but one "how many problems are still
// active" word per iteration goes back to the host.
#include "kkt_common.h"

namespace mi355kkt {

__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = sh[0] + sh[1] + sh[2] + sh[3];      // fixed order: reproducible
    return r;
}
__device__ __forceinline__ double block_max(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

__global__ __launch_bounds__(256) void ipm_start_kernel(IpmState S) {
    __shared__ double sh[4];
    const int b = blockIdx.x, tid = threadIdx.x, m = S.m, n = S.n;
    double* s = S.s + (int64_t)b * m;
    double* z = S.z + (int64_t)b * m;
    const double* q = S.q + (int64_t)b * n;
    const double* h = S.h + (int64_t)b * m;
    double a = 0.0, c = 0.0;
    for (int i = tid; i < n; i += 256) a += q[i] * q[i];
    for (int i = tid; i < m; i += 256) c += h[i] * h[i];
    a = block_sum(a, sh);
    c = block_sum(c, sh);
    double e = 0.0;
    for (int i = tid; i < S.p; i += 256) e += S.b[(int64_t)b * S.p + i] * S.b[(int64_t)b * S.p + i];
    e = block_sum(e, sh);
    if (tid == 0) {
        if (S.p > 0) S.resy0[b] = fmax(1.0, sqrt(e));
        S.resx0[b] = fmax(1.0, sqrt(a));
        S.resz0[b] = fmax(1.0, sqrt(c));
        S.active[b] = 1;
        S.status[b] = 0;
        S.iters[b] = 0;
    }
    double ns = 0.0, ts = -1e300, nz = 0.0, tz = -1e300;
    for (int i = tid; i < m; i += 256) {
        const double zi = z[i], si = -zi;
        s[i] = si;
        ns += si * si;
        ts = fmax(ts, -si);
        nz += zi * zi;
        tz = fmax(tz, -zi);
    }
    ns = sqrt(block_sum(ns, sh));
    ts = block_max(ts, sh);
    nz = sqrt(block_sum(nz, sh));
    tz = block_max(tz, sh);
    const double as = (ts >= -1e-8 * fmax(ns, 1.0)) ? 1.0 + ts : 0.0;
    const double az = (tz >= -1e-8 * fmax(nz, 1.0)) ? 1.0 + tz : 0.0;
    double g = 0.0;
    for (int i = tid; i < m; i += 256) {
        const double si = s[i] + as, zi = z[i] + az;
        s[i] = si;
        z[i] = zi;
        g += si * zi;
    }
    g = block_sum(g, sh);
    if (tid == 0) S.gap[b] = g;
}

__global__ __launch_bounds__(256) void ipm_residual_kernel(IpmState S, int it, int maxiters, double abstol, double reltol,
                                                           double feastol) {
    __shared__ double sh[4];
    const int b = blockIdx.x, tid = threadIdx.x, m = S.m, n = S.n;
    const double* x = S.x + (int64_t)b * n;
    const double* q = S.q + (int64_t)b * n;
    const double* px = S.Px + (int64_t)b * n;
    const double* gtz = S.GTz + (int64_t)b * n;
    double* rx = S.rx + (int64_t)b * n;
    double* s = S.s + (int64_t)b * m;
    double* z = S.z + (int64_t)b * m;
    const double* gx = S.Gx + (int64_t)b * m;
    const double* h = S.h + (int64_t)b * m;
    double* rz = S.rz + (int64_t)b * m;
    const int p = S.p;
    const double* aty = p > 0 ? S.ATy + (int64_t)b * n : nullptr;
    double f0a = 0.0, f0b = 0.0, r2 = 0.0;
    for (int i = tid; i < n; i += 256) {
        const double t = q[i] + px[i];
        f0a += x[i] * t;
        f0b += x[i] * q[i];
        double r = t;
        if (p > 0) r += aty[i];                 // rx = P x + q + A'y + G'z in the reference's order (coneprog.py:2170-2178)
        r += gtz[i];
        rx[i] = r;
        r2 += r * r;
    }
    double resy = 0.0, yry = 0.0;
    if (p > 0) {                                // ry = A x - b (coneprog.py:2181-2183)
        const double* y = S.y + (int64_t)b * p;
        const double* ax = S.Ax + (int64_t)b * p;
        const double* bb = S.b + (int64_t)b * p;
        double* ry = S.ry + (int64_t)b * p;
        double e2 = 0.0, d2 = 0.0;
        for (int i = tid; i < p; i += 256) {
            const double r = ax[i] - bb[i];
            ry[i] = r;
            e2 += r * r;
            d2 += y[i] * r;
        }
        resy = sqrt(block_sum(e2, sh));
        yry = block_sum(d2, sh);
    }
    f0a = block_sum(f0a, sh);
    f0b = block_sum(f0b, sh);
    const double resx = sqrt(block_sum(r2, sh));
    double z2 = 0.0, zr = 0.0;
    for (int i = tid; i < m; i += 256) {
        const double r = s[i] + gx[i] - h[i];
        rz[i] = r;
        z2 += r * r;
        zr += z[i] * r;
    }
    const double resz = sqrt(block_sum(z2, sh));
    zr = block_sum(zr, sh);
    const double gap = S.gap[b];
    const double f0 = 0.5 * (f0a + f0b);
    const double pcost = f0, dcost = f0 + yry + zr - gap;
    double relgap = 1e300;
    if (pcost < 0.0) relgap = gap / -pcost;
    else if (dcost > 0.0) relgap = gap / dcost;
    double pres = resz / S.resz0[b];
    if (p > 0) pres = fmax(resy / S.resy0[b], pres);
    const double dres = resx / S.resx0[b];
    const bool conv = (pres <= feastol) && (dres <= feastol) && ((gap <= abstol) || (relgap <= reltol));
    const bool act = S.active[b] != 0;
    const bool stop = act && (conv || it == maxiters);
    if (stop) {                                   // freeze this problem's answer
        double* xo = S.x_out + (int64_t)b * n;
        double* so = S.s_out + (int64_t)b * m;
        double* zo = S.z_out + (int64_t)b * m;
        for (int i = tid; i < n; i += 256) xo[i] = x[i];
        for (int i = tid; i < m; i += 256) { so[i] = s[i]; zo[i] = z[i]; }
        for (int i = tid; i < p; i += 256) S.y_out[(int64_t)b * p + i] = S.y[(int64_t)b * p + i];
    }
    if (it == 0) {                                // misc.compute_scaling, 'l' block (misc.py:284-287)
        double* d = S.d + (int64_t)b * m;
        double* lm = S.lmbda + (int64_t)b * m;
        for (int i = tid; i < m; i += 256) {
            d[i] = sqrt(s[i] / z[i]);
            lm[i] = sqrt(s[i] * z[i]);
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (act) { S.pcost[b] = pcost; S.dcost[b] = dcost; S.gap_out[b] = gap; }
        if (stop) {
            S.status[b] = conv ? 1 : 2;           // 1 optimal, 2 unknown (maxiters)
            S.iters[b] = it;
            S.active[b] = 0;
        } else if (act) {
            atomicAdd(S.nactive, 1);
        }
    }
    // scaling for the factorisation: finished problems keep a benign system (di = 1)
    const bool still = act && !stop;
    const double* d = S.d + (int64_t)b * m;
    double* di = S.di + (int64_t)b * m;
    for (int i = tid; i < m; i += 256) di[i] = still ? 1.0 / d[i] : 1.0;
}

// problems whose factorisation failed leave the loop with status 'unknown' (coneprog.py:2256-2275)
__global__ void ipm_info_kernel(IpmState S, const int* info, int it, int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B && S.active[b] && info[b] > 0) {
        S.active[b] = 0;
        S.status[b] = 3;                          // unknown: singular KKT matrix
        S.iters[b] = it;
        // x_out/s_out/z_out: copy happens in ipm_freeze_kernel
        S.freeze[b] = 1;
    }
}
__global__ __launch_bounds__(256) void ipm_freeze_kernel(IpmState S) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (!S.freeze[b]) return;
    for (int i = tid; i < S.n; i += 256) S.x_out[(int64_t)b * S.n + i] = S.x[(int64_t)b * S.n + i];
    for (int i = tid; i < S.m; i += 256) {
        S.s_out[(int64_t)b * S.m + i] = S.s[(int64_t)b * S.m + i];
        S.z_out[(int64_t)b * S.m + i] = S.z[(int64_t)b * S.m + i];
    }
    for (int i = tid; i < S.p; i += 256) S.y_out[(int64_t)b * S.p + i] = S.y[(int64_t)b * S.p + i];
    __syncthreads();
    if (tid == 0) S.freeze[b] = 0;
}

__global__ __launch_bounds__(256) void ipm_rhs_kernel(IpmState S, int i01) {
    const int b = blockIdx.x, tid = threadIdx.x, m = S.m, n = S.n;
    const double mu = S.gap[b] / m;
    const double sigma = (i01 == 0) ? 0.0 : S.sigma[b];
    const double* lm = S.lmbda + (int64_t)b * m;
    const double* d = S.d + (int64_t)b * m;
    const double* rz = S.rz + (int64_t)b * m;
    const double* ws3 = S.ws3 + (int64_t)b * m;
    double* ds = S.ds + (int64_t)b * m;
    double* dz = S.dz + (int64_t)b * m;
    for (int i = tid; i < m; i += 256) {
        double v = -lm[i] * lm[i] + sigma * mu;
        if (i01 == 1 && S.correction) v -= ws3[i];      // coneprog.py:2377-2378
        v = v / lm[i];                            // sinv
        ds[i] = v;
        dz[i] = -rz[i] - d[i] * v;                // dz := -rz - W' ds
    }
    const double* rx = S.rx + (int64_t)b * n;
    double* dx = S.dx + (int64_t)b * n;
    for (int i = tid; i < n; i += 256) dx[i] = -rx[i];
    for (int i = tid; i < S.p; i += 256) S.dy[(int64_t)b * S.p + i] = -S.ry[(int64_t)b * S.p + i];
}

__global__ __launch_bounds__(256) void ipm_post_kernel(IpmState S, int i01) {
    __shared__ double sh[4];
    const int b = blockIdx.x, tid = threadIdx.x, m = S.m;
    const double* lm = S.lmbda + (int64_t)b * m;
    double* ds = S.ds + (int64_t)b * m;
    double* dz = S.dz + (int64_t)b * m;
    double* ws3 = S.ws3 + (int64_t)b * m;
    double dot = 0.0, t = 0.0;
    for (int i = tid; i < m; i += 256) {
        const double z = dz[i];
        const double s = ds[i] - z;               // ds := ds - dz
        dot += s * z;
        if (i01 == 0) ws3[i] = s * z;             // sprod
        const double sl = s / lm[i], zl = z / lm[i];   // scale2
        ds[i] = sl;
        dz[i] = zl;
        t = fmax(t, fmax(-sl, -zl));
    }
    dot = block_sum(dot, sh);
    t = block_max(t, sh);
    if (tid == 0) {
        const double step = (t == 0.0) ? 1.0 : fmin(1.0, (i01 == 0 ? 1.0 : 0.99) / t);
        S.step[b] = step;
        if (i01 == 0) {
            double sg = 1.0 - step + dot / S.gap[b] * step * step;
            sg = fmin(1.0, fmax(0.0, sg));
            sg = sg * sg * sg;
            S.sigma[b] = (sg == sg) ? sg : 0.0;
        }
    }
}

__global__ __launch_bounds__(256) void ipm_update_kernel(IpmState S) {
    __shared__ double sh[4];
    const int b = blockIdx.x, tid = threadIdx.x, m = S.m, n = S.n;
    if (!S.active[b]) return;
    const double step = S.step[b];
    double* x = S.x + (int64_t)b * n;
    const double* dx = S.dx + (int64_t)b * n;
    for (int i = tid; i < n; i += 256) x[i] += step * dx[i];
    for (int i = tid; i < S.p; i += 256) S.y[(int64_t)b * S.p + i] += step * S.dy[(int64_t)b * S.p + i];
    double* lm = S.lmbda + (int64_t)b * m;
    double* d = S.d + (int64_t)b * m;
    double* s = S.s + (int64_t)b * m;
    double* z = S.z + (int64_t)b * m;
    const double* ds = S.ds + (int64_t)b * m;
    const double* dz = S.dz + (int64_t)b * m;
    double g = 0.0;
    for (int i = tid; i < m; i += 256) {
        const double l = lm[i];
        const double a = sqrt((1.0 + step * ds[i]) * l);     // misc.py:450-451
        const double c = sqrt((1.0 + step * dz[i]) * l);
        const double dn = d[i] * a / c;                       // tbmv then tbsv
        const double ln = a * c;
        d[i] = dn;
        lm[i] = ln;
        s[i] = dn * ln;
        z[i] = ln / dn;
        g += ln * ln;
    }
    g = block_sum(g, sh);
    if (tid == 0) S.gap[b] = g;
}

void ipm_launch_start(const IpmState& S, int B, hipStream_t st) { hipLaunchKernelGGL(ipm_start_kernel, dim3(B), dim3(256), 0, st, S); }
void ipm_launch_residual(const IpmState& S, int B, int it, int maxiters, double abstol, double reltol, double feastol,
                         hipStream_t st) {
    hipLaunchKernelGGL(ipm_residual_kernel, dim3(B), dim3(256), 0, st, S, it, maxiters, abstol, reltol, feastol);
}
void ipm_launch_info(const IpmState& S, const int* d_info, int it, int B, hipStream_t st) {
    hipLaunchKernelGGL(ipm_info_kernel, dim3((B + 255) / 256), dim3(256), 0, st, S, d_info, it, B);
    hipLaunchKernelGGL(ipm_freeze_kernel, dim3(B), dim3(256), 0, st, S);
}
void ipm_launch_rhs(const IpmState& S, int B, int i01, hipStream_t st) { hipLaunchKernelGGL(ipm_rhs_kernel, dim3(B), dim3(256), 0, st, S, i01); }
void ipm_launch_post(const IpmState& S, int B, int i01, hipStream_t st) { hipLaunchKernelGGL(ipm_post_kernel, dim3(B), dim3(256), 0, st, S, i01); }
void ipm_launch_update(const IpmState& S, int B, hipStream_t st) { hipLaunchKernelGGL(ipm_update_kernel, dim3(B), dim3(256), 0, st, S); }

}  // namespace mi355kkt

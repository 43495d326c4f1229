// Cone-vector operations of the interior-point loops for 'l' and 'q' blocks, shared by conelp_ipm.hip and
// coneqp_ipm.hip: sprod, sinv, ssqr, scale2, max_step, scale (src/C/misc_solvers.c:634, :775, :256, :1052, :85; misc.py:945)
// and the Nesterov-Todd scaling of second-order cones (misc.compute_scaling misc.py:307-354, update_scaling :503-573).
// One workgroup per problem (256 threads, 1024 with 's' blocks); 'l' entries are strided over the workgroup, 'q' blocks walked one cone per
// thread, 's' blocks one after the other by the whole workgroup (cone_ops_s.h).  ST is a state struct with ml, nq, qoff, qdim,
// d, v, beta and the 's' descriptors ns, lq, ldim, sdim, soff, sloff, r, rti, sw1..3, jw (LpState / QpState).
#pragma once
#include "kkt_common.h"
#include "cone_ops_s.h"

namespace mi355kkt {

// Block reductions in a fixed order (wave results added / compared in wave order): sh holds one double per wave (<= 16).
// The loops run with 256 threads (4 waves) for 'l' / 'q' problems and 1024 when the problem has 's' blocks.
__device__ __forceinline__ double lp_block_sum(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double a = sh[0];
    for (int i = 1; i < nw; ++i) a += sh[i];
    return a;
}
__device__ __forceinline__ double lp_block_max(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double a = sh[0];
    for (int i = 1; i < nw; ++i) a = fmax(a, sh[i]);
    return a;
}

// Dynamic LDS of the loop kernels: staging area of the Jacobi iteration (s_jacobi_waves); its size in doubles travels in
// the state struct (S.lds_doubles; 0 for problems without 's' blocks).
extern __shared__ double ipm_dyn_lds[];

// 1 / x and 1 / sqrt(x) for normal positive-range arguments: v_rcp_f64 / v_rsq_f64 + two Newton steps / one Halley step
__device__ __forceinline__ double jr_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    return fma(y, e, y);
}
__device__ __forceinline__ double jr_rsqrt(double p) {
    const double y0 = __builtin_amdgcn_rsq(p);
    const double e = fma(-(p * y0), y0, 1.0);                  // 1 - p y0^2
    return fma(y0 * e, fma(0.375, e, 0.5), y0);                // y0 (1 + e/2 + 3 e^2 / 8)
}

// One-sided Jacobi rotations of cone_ops_s.h's s_jacobi for a whole workgroup: every GROUP of lpp lanes (4..64, the largest
// power of two that still gives every column pair of a round its own group) owns one pair: dot products by a butterfly
// inside the group, the rotation computed redundantly by its lanes, both columns of G and V updated by the same group --
// so a round needs ONE workgroup barrier and no scratch.  LDSMEM: G (and V) live in the dynamic LDS (copied in and out
// here); otherwise in global memory.
template <bool LDSMEM>
__device__ __noinline__ void s_jacobi_waves(double* Gg, double* Vg, int m, double* sh) {
    const int M = m + (m & 1), np = M / 2, mm = m * m;
    // lanes per pair: the rotation arithmetic, the butterfly and the pair bookkeeping cost every WAVE ~150 instructions per
    // round whatever the group size, so few lanes with ~10 rows each (fewer waves busy) beat many lanes with 2-3 rows
    int lpp = 4;
    while (lpp < 64 && m > 14 * lpp) lpp <<= 1;                // power of two nearest to m / 10 (geometric rounding)
    while (lpp > 4 && np * lpp > (int)blockDim.x) lpp >>= 1;
    const int sub = threadIdx.x & (lpp - 1), grp = threadIdx.x / lpp, ngroups = blockDim.x / lpp;
    double* G = LDSMEM ? ipm_dyn_lds : Gg;
    double* V = Vg ? (LDSMEM ? ipm_dyn_lds + mm : Vg) : nullptr;
    if (LDSMEM) {
        for (int e = threadIdx.x; e < mm; e += blockDim.x) {
            G[e] = Gg[e];
            if (V) V[e] = Vg[e];
        }
        __syncthreads();
    }
    const double tol = fmax(1e-15, 4.5e-16 * sqrt((double)m)), tol2 = tol * tol;
    constexpr int JR = 13;
    const bool cached = m <= JR * lpp;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double myrot = 0.0;
        for (int r = 0; r < M - 1; ++r) {
            for (int i = grp; i < np; i += ngroups) {
                int p, q;
                s_pair(i, r, M, p, q);
                if (p >= m || q >= m) continue;
                double *gp = G + (size_t)p * m, *gq = G + (size_t)q * m;
                double al = 0.0, be = 0.0, ga = 0.0;
                double ca[JR], cb[JR];               // the group's rows of the two columns, kept for the rotation
                if (cached) {
#pragma unroll
                    for (int t = 0; t < JR; ++t) {
                        const int k = sub + t * lpp;
                        ca[t] = (k < m) ? gp[k] : 0.0;
                        cb[t] = (k < m) ? gq[k] : 0.0;
                        al += ca[t] * ca[t];
                        be += cb[t] * cb[t];
                        ga += ca[t] * cb[t];
                    }
                } else {
                    for (int k = sub; k < m; k += lpp) {
                        const double a = gp[k], b = gq[k];
                        al += a * a;
                        be += b * b;
                        ga += a * b;
                    }
                }
                for (int o = lpp >> 1; o > 0; o >>= 1) {
                    al += __shfl_xor(al, o, 64);
                    be += __shfl_xor(be, o, 64);
                    ga += __shfl_xor(ga, o, 64);
                }
                if (ga * ga > tol2 * al * be) {
                    // the rotation with hardware reciprocal / reciprocal-square-root seeds + Newton / Halley steps (~1 ulp):
                    // every wave of the workgroup issues this sequence once per round, the IEEE divisions and square
                    // roots of the generic version were a third of the instructions of a round
                    const double zeta = (be - al) * jr_rcp(2.0 * ga);
                    const double u = fma(zeta, zeta, 1.0);
                    const double t = copysign(1.0, zeta) * jr_rcp(fabs(zeta) + u * jr_rsqrt(u));
                    const double cc = jr_rsqrt(fma(t, t, 1.0)), ss = cc * t;
                    if (cached) {
#pragma unroll
                        for (int t2 = 0; t2 < JR; ++t2) {
                            const int k = sub + t2 * lpp;
                            if (k < m) {
                                gp[k] = cc * ca[t2] - ss * cb[t2];
                                gq[k] = ss * ca[t2] + cc * cb[t2];
                            }
                        }
                    } else {
                        for (int k = sub; k < m; k += lpp) {
                            const double a = gp[k], b = gq[k];
                            gp[k] = cc * a - ss * b;
                            gq[k] = ss * a + cc * b;
                        }
                    }
                    if (V) {
                        double *vp = V + (size_t)p * m, *vq = V + (size_t)q * m;
                        for (int k = sub; k < m; k += lpp) {
                            const double a = vp[k], b = vq[k];
                            vp[k] = cc * a - ss * b;
                            vq[k] = ss * a + cc * b;
                        }
                    }
                    myrot = 1.0;
                }
            }
            __syncthreads();
        }
        if (lp_block_max(myrot, sh) == 0.0) break;
    }
    if (LDSMEM) {
        for (int e = threadIdx.x; e < mm; e += blockDim.x) {
            Gg[e] = G[e];
            if (V) Vg[e] = V[e];
        }
        __syncthreads();
    }
}

// the workgroup as the team of the 's'-block operations (cone_ops_s.h)
struct ParWG {
    double* sh;
    int lds_doubles;          // capacity of ipm_dyn_lds
    double* jw;               // Jacobi / sorting scratch (s_jw_doubles(max order, 1024))
    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ int nt() const { return blockDim.x; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ __forceinline__ double sum(double v) const { return lp_block_sum(v, sh); }
    __device__ __forceinline__ double max(double v) const { return lp_block_max(v, sh); }
    __device__ __forceinline__ void jacobi(double* G, double* V, int m, double* /*jw*/) const {
        if ((V ? 2 : 1) * m * m <= lds_doubles) s_jacobi_waves<true>(G, V, m, sh);
        else s_jacobi_waves<false>(G, V, m, sh);
    }
};
__device__ __forceinline__ double lp_dot(const double* a, const double* b, int n, double* sh) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += a[i] * b[i];
    return lp_block_sum(v, sh);
}

// One WAVE as the team: blocks of order <= SW_SMALL are handled one per wave, 16 at a time (programs with many small
// blocks), with wave-level "barriers" (memory fence + scheduling barrier; the lanes run in lock step) and the generic
// barrier-phased Jacobi rotations of cone_ops_s.h.  Uses no workgroup barrier, so different waves may work on different
// blocks of different orders at the same time.
constexpr int SW_SMALL = 16;
struct ParWave {
    double* jw;               // this wave's own scratch (s_jw_doubles(SW_SMALL, 64))
    __device__ __forceinline__ int tid() const { return threadIdx.x & 63; }
    __device__ __forceinline__ int nt() const { return 64; }
    __device__ __forceinline__ void sync() const {
        // workgroup scope: what __syncthreads() gives the workgroup team (the lanes of a wave share the compute unit's L1)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ double sum(double v) const {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
    __device__ __forceinline__ double max(double v) const {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
        return v;
    }
    __device__ __forceinline__ void jacobi(double* G, double* V, int m, double* scratch) const {
        s_jacobi_rotations(*this, G, V, m, scratch);
    }
};

// op(team, k) for every 's' block k: the small ones one per wave, the others by the whole workgroup one after the other.
// Contains workgroup barriers: uniform control flow only.
template <class ST, class F>
__device__ __forceinline__ void cv_for_sblocks(const ST& S, double* sh, F&& op) {
    __syncthreads();
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const bool serial = S.swmax < 0;                     // (debug: the wave team, but one block at a time)
    const int swmax = serial ? -S.swmax : S.swmax;       // blocks up to this order go one per wave (<= SW_SMALL; 0: none)
    if (S.smin <= swmax) {
        const ParWave pw{S.jww + (size_t)wave * s_jw_doubles(SW_SMALL, 64)};
        if (!serial) {
            for (int k = wave; k < S.ns; k += nwaves)
                if (S.sdim[k] <= swmax) op(pw, k);
        } else if (wave == 0) {
            for (int k = 0; k < S.ns; ++k)
                if (S.sdim[k] <= swmax) op(pw, k);
        }
        __syncthreads();
    }
    if (S.smax > swmax) {
        const ParWG par{sh, S.lds_doubles, S.jw};
        for (int k = 0; k < S.ns; ++k)
            if (S.sdim[k] > swmax) op(par, k);
    }
}

// ---- second-order-cone pieces, one cone (x: the cone's own entries, mk of them) -----------------------------
__host__ __device__ __forceinline__ double q_nrm1(const double* x, int mk) {      // ||x[1:]||
    double a = 0.0;
    for (int i = 1; i < mk; ++i) a += x[i] * x[i];
    return sqrt(a);
}
__host__ __device__ __forceinline__ double q_jnrm2(const double* x, int mk) {     // misc.py:848-857
    const double a = q_nrm1(x, mk);
    return sqrt(x[0] - a) * sqrt(x[0] + a);
}
__host__ __device__ __forceinline__ void q_sprod(double* x, const double* y, int mk) {           // x := x o y
    double a = 0.0;
    for (int i = 0; i < mk; ++i) a += y[i] * x[i];
    const double x0 = x[0], y0 = y[0];
    for (int i = 1; i < mk; ++i) x[i] = y0 * x[i] + x0 * y[i];
    x[0] = a;
}
__host__ __device__ __forceinline__ void q_sinv(double* x, const double* y, int mk) {            // x := y o\ x
    double a = q_nrm1(y, mk);
    a = (y[0] + a) * (y[0] - a);
    const double c = x[0];
    double d = 0.0;
    for (int i = 1; i < mk; ++i) d += x[i] * y[i];
    const double x0 = c * y[0] - d;
    const double al = a / y[0], be = d / y[0] - c, ia = 1.0 / a;
    for (int i = 1; i < mk; ++i) x[i] = (al * x[i] + be * y[i]) * ia;
    x[0] = x0 * ia;
}
__host__ __device__ __forceinline__ void q_ssqr(double* x, const double* y, int mk) {            // x := y o y
    double a = 0.0;
    for (int i = 0; i < mk; ++i) a += y[i] * y[i];
    const double y0 = y[0];
    for (int i = 1; i < mk; ++i) x[i] = 2.0 * y0 * y[i];
    x[0] = a;
}
__host__ __device__ __forceinline__ void q_scale2(const double* l, double* x, int mk, bool inverse) {
    double a = q_nrm1(l, mk);
    a = sqrt(l[0] + a) * sqrt(l[0] - a);
    double lx = 0.0;
    if (!inverse) {
        for (int i = 1; i < mk; ++i) lx += l[i] * x[i];
        lx = (l[0] * x[0] - lx) / a;
    } else {
        for (int i = 0; i < mk; ++i) lx += l[i] * x[i];
        lx = lx / a;
    }
    const double x0 = x[0];
    double b = (x0 + lx) / (l[0] / a + 1.0) / a;
    if (!inverse) b = -b;
    const double sc = inverse ? a : 1.0 / a;
    for (int i = 1; i < mk; ++i) x[i] = (x[i] + b * l[i]) * sc;
    x[0] = lx * sc;
}
__host__ __device__ __forceinline__ void q_scale(double* x, const double* v, double beta, int mk, bool inverse) {   // W x / W^-1 x
    double w = 0.0;
    if (!inverse) {
        for (int i = 0; i < mk; ++i) w += v[i] * x[i];
        x[0] = beta * (2.0 * v[0] * w - x[0]);
        for (int i = 1; i < mk; ++i) x[i] = beta * (x[i] + 2.0 * v[i] * w);
    } else {
        for (int i = 1; i < mk; ++i) w += v[i] * x[i];
        w = v[0] * x[0] - w;
        const double ib = 1.0 / beta;
        x[0] = (2.0 * v[0] * w - x[0]) * ib;
        for (int i = 1; i < mk; ++i) x[i] = (x[i] - 2.0 * v[i] * w) * ib;
    }
}

// ---- whole cone vectors (l part strided over the workgroup, q part one cone per thread, 's' blocks one by one) ------
// Functions that touch 's' blocks contain workgroup barriers: call them from uniform control flow only.
template <class ST>
__device__ __forceinline__ double cv_maxstep(const ST& S, const double* x, double* sh) {
    double t = -1e300;
    for (int i = threadIdx.x; i < S.ml; i += blockDim.x) t = fmax(t, -x[i]);
    for (int k = threadIdx.x; k < S.nq; k += blockDim.x) t = fmax(t, q_nrm1(x + S.qoff[k], S.qdim[k]) - x[S.qoff[k]]);
    if (S.ns > 0)                                        // max_step without sigma: -lambda_min of every block
        cv_for_sblocks(S, sh, [&](const auto& par, int k) {
            const int o = S.soff[k] - S.lq;
            t = fmax(t, -s_min_eig_blk(par, x + S.soff[k], S.sdim[k], S.sw1 + o, S.sw2 + o, par.jw));
        });
    return lp_block_max(t, sh);
}
// max_step with sigma (misc_solvers.c:1131-1136): the 's' blocks of x are replaced by their eigenvectors, sig (compact
// layout, sum(s) entries) receives the eigenvalues
template <class ST>
__device__ __forceinline__ double cv_maxstep_sigma(const ST& S, double* x, double* sig, double* sh) {
    double t = -1e300;
    for (int i = threadIdx.x; i < S.ml; i += blockDim.x) t = fmax(t, -x[i]);
    for (int k = threadIdx.x; k < S.nq; k += blockDim.x) t = fmax(t, q_nrm1(x + S.qoff[k], S.qdim[k]) - x[S.qoff[k]]);
    if (S.ns > 0)
        cv_for_sblocks(S, sh, [&](const auto& par, int k) {
            const int o = S.soff[k] - S.lq;
            double* sg = sig + (S.sloff[k] - S.lq);
            s_eig_blk(par, x + S.soff[k], sg, S.sdim[k], S.sw1 + o, S.sw2 + o, par.jw);
            t = fmax(t, -sg[0]);
        });
    return lp_block_max(t, sh);
}
template <class ST>
__device__ __forceinline__ void cv_add_e(const ST& S, double* x, double a, bool with_s = true) {
    for (int i = threadIdx.x; i < S.ml; i += blockDim.x) x[i] += a;
    for (int k = threadIdx.x; k < S.nq; k += blockDim.x) x[S.qoff[k]] += a;
    for (int k = 0; with_s && k < S.ns; ++k)
        for (int i = threadIdx.x; i < S.sdim[k]; i += blockDim.x) x[S.soff[k] + i * (S.sdim[k] + 1)] += a;
}
// x := x o y, y a cone vector (sprod with diag = 'N')
template <class ST>
__device__ __forceinline__ void cv_sprod(const ST& S, double* x, const double* y, double* sh) {
    for (int i = threadIdx.x; i < S.ml; i += blockDim.x) x[i] *= y[i];
    for (int k = threadIdx.x; k < S.nq; k += blockDim.x) q_sprod(x + S.qoff[k], y + S.qoff[k], S.qdim[k]);
    if (S.ns > 0)
        cv_for_sblocks(S, sh, [&](const auto& par, int k) {
            s_sprod_blk(par, x + S.soff[k], y + S.soff[k], S.sdim[k], S.sw1 + (S.soff[k] - S.lq));
        });
}
// x := x o lmbda, lmbda in its compact layout (sprod with diag = 'D')
template <class ST>
__device__ __forceinline__ void cv_sprod_diag(const ST& S, double* x, const double* l, double* sh) {
    for (int i = threadIdx.x; i < S.ml; i += blockDim.x) x[i] *= l[i];
    for (int k = threadIdx.x; k < S.nq; k += blockDim.x) q_sprod(x + S.qoff[k], l + S.qoff[k], S.qdim[k]);
    if (S.ns > 0)
        cv_for_sblocks(S, sh, [&](const auto& par, int k) { s_sprod_diag_blk(par, x + S.soff[k], l + S.sloff[k], S.sdim[k], false); });
}
// x := lmbda o\ x (sinv: the second argument is always the compact lmbda)
template <class ST>
__device__ __forceinline__ void cv_sinv(const ST& S, double* x, const double* l, double* sh) {
    for (int i = threadIdx.x; i < S.ml; i += blockDim.x) x[i] /= l[i];
    for (int k = threadIdx.x; k < S.nq; k += blockDim.x) q_sinv(x + S.qoff[k], l + S.qoff[k], S.qdim[k]);
    if (S.ns > 0)
        cv_for_sblocks(S, sh, [&](const auto& par, int k) { s_sprod_diag_blk(par, x + S.soff[k], l + S.sloff[k], S.sdim[k], true); });
}
// x := y o y for the compact lmbda layout (misc.ssqr, misc.py:945-972: the 's' part is diagonal)
template <class ST>
__device__ __forceinline__ void cv_ssqr(const ST& S, double* x, const double* y) {
    for (int i = threadIdx.x; i < S.ml; i += blockDim.x) x[i] = y[i] * y[i];
    for (int k = threadIdx.x; k < S.nq; k += blockDim.x) q_ssqr(x + S.qoff[k], y + S.qoff[k], S.qdim[k]);
    for (int i = S.lq + threadIdx.x; i < S.ldim; i += blockDim.x) x[i] = y[i] * y[i];
}
template <class ST>
__device__ __forceinline__ void cv_scale2(const ST& S, const double* l, double* x, bool inverse, double* sh) {
    for (int i = threadIdx.x; i < S.ml; i += blockDim.x) x[i] = inverse ? x[i] * l[i] : x[i] / l[i];
    for (int k = threadIdx.x; k < S.nq; k += blockDim.x) q_scale2(l + S.qoff[k], x + S.qoff[k], S.qdim[k], inverse);
    if (S.ns > 0)
        cv_for_sblocks(S, sh, [&](const auto& par, int k) { s_scale2_blk(par, l + S.sloff[k], x + S.soff[k], S.sdim[k], inverse); });
}
// misc.scale: x := W x (trans: W' x) or W^-1 x (trans: W^-T x).  'l' and 'q' blocks are symmetric; 's' blocks (misc.py:
// 118-164): r' X r | r X r' (trans) | rti X rti' (inverse) | rti' X rti (inverse, trans)
template <class ST>
__device__ __forceinline__ void cv_scale(const ST& S, double* x, bool inverse, bool trans, double* sh) {
    for (int i = threadIdx.x; i < S.ml; i += blockDim.x) x[i] = inverse ? x[i] / S.d[i] : x[i] * S.d[i];
    for (int k = threadIdx.x; k < S.nq; k += blockDim.x)
        q_scale(x + S.qoff[k], S.v + (S.qoff[k] - S.ml), S.beta[k], S.qdim[k], inverse);
    if (S.ns > 0)
        cv_for_sblocks(S, sh, [&](const auto& par, int k) {
            const int o = S.soff[k] - S.lq;
            s_scale_blk(par, x + S.soff[k], (inverse ? S.rti : S.r) + o, S.sdim[k], trans == inverse, S.sw1 + o);
        });
}
// x := the cone vector with the compact lmbda on it: copy for 'l' / 'q', diag(lmbda_k) for the 's' blocks
// (coneprog.py:1264-1273, :1404-1413)
template <class ST>
__device__ __forceinline__ void cv_expand(const ST& S, double* x, const double* l) {
    for (int i = threadIdx.x; i < S.lq; i += blockDim.x) x[i] = l[i];
    for (int k = 0; k < S.ns; ++k) {
        const int m = S.sdim[k];
        for (int e = threadIdx.x; e < m * m; e += blockDim.x) x[S.soff[k] + e] = (e % m == e / m) ? l[S.sloff[k] + e % m] : 0.0;
    }
}
// upper triangles of the 's' blocks := lower triangles (after a KKT solve, which like the reference's returns lower
// triangles only, misc_solvers.c:552-601)
template <class ST>
__device__ __forceinline__ void cv_symm(const ST& S, double* x) {
    for (int k = 0; k < S.ns; ++k) {
        const int m = S.sdim[k];
        for (int e = threadIdx.x; e < m * m; e += blockDim.x)
            if (e % m < e / m) x[S.soff[k] + e] = x[S.soff[k] + (e / m) + (e % m) * m];
    }
}
// the 's' part of "ds, dz := the factors Ls, Lz of the updated variables in the current scaling" (coneprog.py:1364-1395):
// sig := (1 + step sig) / lmbda, then column i of the block (eigenvectors scaled by scale2 inverse) *= sqrt(sig_i)
template <class ST>
__device__ __forceinline__ void cv_s_factors(const ST& S, const double* l, double* x, double* sig, double step) {
    for (int k = 0; k < S.ns; ++k) {
        const int m = S.sdim[k];
        double* sg = sig + (S.sloff[k] - S.lq);
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += blockDim.x) sg[i] = (1.0 + step * sg[i]) / l[S.sloff[k] + i];
        __syncthreads();
        for (int e = threadIdx.x; e < m * m; e += blockDim.x) x[S.soff[k] + e] *= sqrt(sg[e / m]);
    }
}

// misc.compute_scaling for one second-order cone (misc.py:307-354): v_k, beta_k, lambda_k from s_k, z_k
__host__ __device__ __forceinline__ void q_compute_scaling(const double* sk, const double* zk, double* v, double* beta,
                                                           double* lk, int mk) {
    const double aa = q_jnrm2(sk, mk), bb = q_jnrm2(zk, mk);
    *beta = sqrt(aa / bb);
    double dsz = 0.0;
    for (int i = 0; i < mk; ++i) dsz += sk[i] * zk[i];
    const double cc = sqrt((dsz / aa / bb + 1.0) / 2.0);
    // vk = 1/(2c) (sk/a + J zk/b);  then v = (vk + e) / sqrt(2 (vk0 + 1))
    for (int i = 0; i < mk; ++i) {
        double t = zk[i] * (-1.0 / bb);
        if (i == 0) t = -t;
        t += sk[i] * (1.0 / aa);
        v[i] = t * (1.0 / 2.0 / cc);
    }
    v[0] += 1.0;
    const double f = 1.0 / sqrt(2.0 * v[0]);
    for (int i = 0; i < mk; ++i) v[i] *= f;
    const double dd = 2.0 * cc + sk[0] / aa + zk[0] / bb;
    const double fs = (cc + zk[0] / bb) / dd / aa, fz = (cc + sk[0] / aa) / dd / bb, sq = sqrt(aa * bb);
    lk[0] = cc * sq;
    for (int i = 1; i < mk; ++i) lk[i] = (sk[i] * fs + zk[i] * fz) * sq;
}

// misc.update_scaling for one second-order cone (misc.py:503-573); sk, zk: the updated variables in the current scaling
// (normalised in place), v_k, beta_k, lambda_k updated
__host__ __device__ __forceinline__ void q_update_scaling(double* sk, double* zk, double* v, double* beta, double* lk, int mk) {
    const double aa = q_jnrm2(sk, mk);
    for (int i = 0; i < mk; ++i) sk[i] *= 1.0 / aa;
    const double bb = q_jnrm2(zk, mk);
    for (int i = 0; i < mk; ++i) zk[i] *= 1.0 / bb;
    double dsz = 0.0, vs = 0.0, vz1 = 0.0;
    for (int i = 0; i < mk; ++i) {
        dsz += sk[i] * zk[i];
        vs += v[i] * sk[i];
        if (i > 0) vz1 += v[i] * zk[i];
    }
    const double cc = sqrt((1.0 + dsz) / 2.0);
    const double vz = v[0] * zk[0] - vz1;                       // v' J z
    const double vq = (vs + vz) / 2.0 / cc;
    const double vu = vs - vz;
    const double wk0 = 2.0 * v[0] * vq - (sk[0] + zk[0]) / 2.0 / cc;
    const double dd = (v[0] * vu - sk[0] / 2.0 + zk[0] / 2.0) / (wk0 + 1.0);
    const double fv = 2.0 * (-dd * vq + 0.5 * vu), fs = 0.5 * (1.0 - dd / cc), fz = 0.5 * (1.0 + dd / cc);
    const double sq = sqrt(aa * bb);
    lk[0] = cc * sq;
    for (int i = 1; i < mk; ++i) lk[i] = (v[i] * fv + sk[i] * fs + zk[i] * fz) * sq;
    // v := (2 v v' - J) q, then v := v^{1/2}
    const double s0 = sk[0];
    for (int i = 0; i < mk; ++i) {
        double t = 2.0 * vq * v[i];
        if (i == 0) t -= s0 / 2.0 / cc;
        else t += sk[i] * (0.5 / cc);
        t += zk[i] * (-0.5 / cc);
        v[i] = t;
    }
    v[0] += 1.0;
    const double f = 1.0 / sqrt(2.0 * v[0]);
    for (int i = 0; i < mk; ++i) v[i] *= f;
    *beta *= sqrt(aa / bb);
}

// misc.compute_scaling, 'l' and 'q' blocks: d, (v, beta), lmbda from s, z
template <class ST>
__device__ __forceinline__ void cv_compute_scaling(const ST& S, const double* s, const double* z, double* lmbda, double* sh) {
    const int tid = threadIdx.x;
    for (int i = tid; i < S.ml; i += blockDim.x) {
        S.d[i] = sqrt(s[i] / z[i]);
        lmbda[i] = sqrt(s[i] * z[i]);
    }
    for (int k = tid; k < S.nq; k += blockDim.x) {
        const int o = S.qoff[k];
        q_compute_scaling(s + o, z + o, S.v + (o - S.ml), S.beta + k, lmbda + o, S.qdim[k]);
    }
    if (S.ns > 0)                                        // misc.py:374-417
        cv_for_sblocks(S, sh, [&](const auto& par, int k) {
            const int o = S.soff[k] - S.lq;
            s_compute_scaling_blk(par, s + S.soff[k], z + S.soff[k], S.r + o, S.rti + o, lmbda + S.sloff[k], S.sdim[k], S.sw1 + o,
                                  S.sw2 + o, S.sw3 + o, par.jw);
        });
}

// misc.update_scaling, 'l' (misc.py:444-464) and 'q' (:503-573) blocks; ds, dz: the updated variables in the current
// scaling (the 'q' parts are normalised in place)
template <class ST>
__device__ __forceinline__ void cv_update_scaling(const ST& S, double* lmbda, double* ds, double* dz, double* sh) {
    const int tid = threadIdx.x;
    for (int i = tid; i < S.ml; i += blockDim.x) {
        const double a = sqrt(ds[i]), c = sqrt(dz[i]);
        S.d[i] = S.d[i] * a / c;
        lmbda[i] = a * c;
    }
    for (int k = tid; k < S.nq; k += blockDim.x) {
        const int o = S.qoff[k];
        q_update_scaling(ds + o, dz + o, S.v + (o - S.ml), S.beta + k, lmbda + o, S.qdim[k]);
    }
    if (S.ns > 0)                                        // misc.py:592-634; the 's' blocks of ds, dz hold Ls, Lz
        cv_for_sblocks(S, sh, [&](const auto& par, int k) {
            const int o = S.soff[k] - S.lq;
            s_update_scaling_blk(par, ds + S.soff[k], dz + S.soff[k], S.r + o, S.rti + o, lmbda + S.sloff[k], S.sdim[k], S.sw1 + o,
                                 S.sw2 + o, par.jw);
        });
}

}  // namespace mi355kkt

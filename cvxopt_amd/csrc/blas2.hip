// HBM-bound level-2 kernels of solve() for gfx950 (reference src/python/misc.py:1489-1565):
//   gemv_t_scaled   zs = w .* z; x += G' (w .* zs)       misc.py:1513 scale(z) + :1524 base.gemv(Gs, z, x, trans='T')
//   gemv_n_scaled   z := alpha w .* (G x) + beta zs   misc.py:1563 base.gemv(Gs, x, z, beta=-1)
//   trsm_lower      x := L^-1 x / L^-T x   misc.py:1529 / :1555 blas.trsv  (and :1470 blas.trsm for Asct)
// G is never rescaled in memory: Gs = diag(w) G is applied on the fly, so the 1 GB G block is read
// exactly once per product and no Gs copy is written.
#include <cstdlib>

#include "kkt_common.h"

namespace mi355kkt {

struct __attribute__((aligned(8))) d2u { double x, y; };

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// zs = w .* z and zss = w .* zs   (w == nullptr: zs = z, zss unused)
__global__ __launch_bounds__(256) void scale_vec_kernel(const double* __restrict__ w, const double* z, double* zs,
                                                        double* __restrict__ zss, int m) {   // z may alias zs
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < m) {
        if (w) {
            const double t = w[i] * z[i];
            zs[i] = t;
            zss[i] = w[i] * t;
        } else {
            zs[i] = z[i];
        }
    }
}

// one wave per column: y[j] += sum_i G[i,j] zs[i]
__global__ __launch_bounds__(256) void gemv_t_kernel(const double* __restrict__ G, int64_t ldg, int m, int n,
                                                     const double* __restrict__ zs, double* __restrict__ y,
                                                     int64_t sG) {
    G += (int64_t)blockIdx.z * sG;               // batched problems: vectors are packed back to back
    zs += (int64_t)blockIdx.z * m;
    y += (int64_t)blockIdx.z * n;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= n) return;
    const double* __restrict__ g = G + (int64_t)j * ldg;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int i = lane * 2;
    for (; i + 384 + 1 < m; i += 512) {   // 4 x (64 lanes x 2 doubles)
        const d2u a0 = *reinterpret_cast<const d2u*>(g + i);
        const d2u a1 = *reinterpret_cast<const d2u*>(g + i + 128);
        const d2u a2 = *reinterpret_cast<const d2u*>(g + i + 256);
        const d2u a3 = *reinterpret_cast<const d2u*>(g + i + 384);
        const d2u z0 = *reinterpret_cast<const d2u*>(zs + i);
        const d2u z1 = *reinterpret_cast<const d2u*>(zs + i + 128);
        const d2u z2 = *reinterpret_cast<const d2u*>(zs + i + 256);
        const d2u z3 = *reinterpret_cast<const d2u*>(zs + i + 384);
        s0 += a0.x * z0.x + a0.y * z0.y;
        s1 += a1.x * z1.x + a1.y * z1.y;
        s2 += a2.x * z2.x + a2.y * z2.y;
        s3 += a3.x * z3.x + a3.y * z3.y;
    }
    for (; i < m; i += 128) {
        s0 += g[i] * zs[i];
        if (i + 1 < m) s1 += g[i + 1] * zs[i + 1];
    }
    const double s = wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) y[j] += s;
}

constexpr int GN_COLS = 256;   // columns per chunk of the row-parallel product
// partial[chunk][i] = sum_{j in chunk} G[i,j] x[j]; each thread owns two rows
__global__ __launch_bounds__(256) void gemv_n_partial_kernel(const double* __restrict__ G, int64_t ldg, int m,
                                                             int n, const double* __restrict__ x,
                                                             double* __restrict__ partial, int64_t sG) {
    G += (int64_t)blockIdx.z * sG;
    x += (int64_t)blockIdx.z * n;
    partial += (int64_t)blockIdx.z * gridDim.y * m;
    const int i = (blockIdx.x * 256 + threadIdx.x) * 2;
    const int j0 = blockIdx.y * GN_COLS;
    const int j1 = min(n, j0 + GN_COLS);
    if (i >= m) return;
    const bool two = (i + 1 < m);
    double s0 = 0.0, s1 = 0.0;
    const double* __restrict__ g = G + i + (int64_t)j0 * ldg;
    if (two) {
        int j = j0;
        for (; j + 3 < j1; j += 4) {
            const d2u a0 = *reinterpret_cast<const d2u*>(g);
            const d2u a1 = *reinterpret_cast<const d2u*>(g + ldg);
            const d2u a2 = *reinterpret_cast<const d2u*>(g + 2 * ldg);
            const d2u a3 = *reinterpret_cast<const d2u*>(g + 3 * ldg);
            const double x0 = x[j], x1 = x[j + 1], x2 = x[j + 2], x3 = x[j + 3];
            s0 += a0.x * x0;
            s1 += a0.y * x0;
            s0 += a1.x * x1;
            s1 += a1.y * x1;
            s0 += a2.x * x2;
            s1 += a2.y * x2;
            s0 += a3.x * x3;
            s1 += a3.y * x3;
            g += 4 * ldg;
        }
        for (; j < j1; ++j) {
            const d2u a0 = *reinterpret_cast<const d2u*>(g);
            s0 += a0.x * x[j];
            s1 += a0.y * x[j];
            g += ldg;
        }
        partial[(int64_t)blockIdx.y * m + i] = s0;
        partial[(int64_t)blockIdx.y * m + i + 1] = s1;
    } else {
        for (int j = j0; j < j1; ++j) {
            s0 += g[0] * x[j];
            g += ldg;
        }
        partial[(int64_t)blockIdx.y * m + i] = s0;
    }
}

// z[i] = alpha * w[i] * sum_c partial[c][i] + beta * zs[i]   (z may alias zs)
__global__ __launch_bounds__(256) void gemv_n_finish_kernel(const double* __restrict__ partial, int nchunks, int m,
                                                            const double* __restrict__ w, const double* zs,
                                                            double* z, double alpha, double beta) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    {   // batched problems along blockIdx.z
        const int64_t bz = blockIdx.z;
        partial += bz * nchunks * m;
        if (w) w += bz * m;
        zs += bz * m;
        z += bz * m;
    }
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += partial[(int64_t)c * m + i];
    z[i] = alpha * (w ? w[i] * s : s) + (beta != 0.0 ? beta * zs[i] : 0.0);
}

size_t gemv_work_doubles(int m, int n) {
    const size_t nchunks = (size_t)((n + GN_COLS - 1) / GN_COLS);
    return (nchunks ? nchunks : 1) * (size_t)(m > 0 ? m : 1);
}

// zs := w .* z (kept for the final z update);  y += (diag(w) G)' zs = G' (w .* zs).
// work: >= m doubles (only used when w != nullptr).
int launch_gemv_t_scaled(const double* G, int64_t ldg, int m, int n, const double* w, const double* z,
                         double* zs, double* y, double* work, hipStream_t st, int nbatch, int64_t sG) {
    if (m <= 0) return 0;
    const int64_t mt = (int64_t)m * nbatch;      // w, z, zs, work are [nbatch][m], contiguous
    hipLaunchKernelGGL(scale_vec_kernel, dim3((unsigned)((mt + 255) / 256)), dim3(256), 0, st, w, z, zs, work, (int)mt);
    KKT_HIP_CHECK(hipGetLastError());
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gemv_t_kernel, dim3((n + 3) / 4, 1, nbatch), dim3(256), 0, st, G, ldg, m, n, w ? work : zs, y, sG);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_gemv_n_scaled(const double* G, int64_t ldg, int m, int n, const double* w, const double* x,
                         const double* zs, double* z, double alpha, double beta, double* work, hipStream_t st,
                         int nbatch, int64_t sG) {
    if (m <= 0) return 0;
    const int nchunks = (n + GN_COLS - 1) / GN_COLS;
    if (nchunks > 0) {
        hipLaunchKernelGGL(gemv_n_partial_kernel, dim3((m + 511) / 512, nchunks, nbatch), dim3(256), 0, st, G, ldg, m, n,
                           x, work, sG);
        KKT_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(gemv_n_finish_kernel, dim3((m + 255) / 256, 1, nbatch), dim3(256), 0, st, work, nchunks, m, w, zs,
                       z, alpha, beta);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ===================================================================================================
// Residual products of the interior-point loops in ONE pass over G:  Gx = G x  and  GTz = G' z.
// (coneprog.py:2170-2186 computes rx and rz with two sgemv calls; the batched loop paid two of its seven passes over G for them.)
//   A workgroup owns 64 columns (16 per wave); a wave walks its columns in row blocks of 1024: the lane holds 16 rows (8 pairs,
//   128 apart: whole 1 KB pieces of the column per load instruction), their z in registers and 16 accumulators of G x; the
//   column's dot product with z is wave-reduced and kept in LDS.  After a row block the four waves' accumulators are summed in
//   LDS and written as partial[chunk = blockIdx.x][row]; gemv_n_finish_kernel adds the chunks in a fixed order.
// ===================================================================================================
constexpr int NT_COLS = 64;     // columns per workgroup
constexpr int NT_ROWS = 1024;   // rows per row block
__global__ __launch_bounds__(256) void gemv_nt_fused_kernel(const double* __restrict__ G, int64_t ldg, int m, int n,
                                                            const double* __restrict__ x, const double* __restrict__ z,
                                                            double* __restrict__ partial, double* __restrict__ GTz, int64_t sG) {
    __shared__ double ylds[4][NT_ROWS];
    __shared__ double dots[NT_COLS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    {   // batched problems along blockIdx.z
        const int64_t bz = blockIdx.z;
        G += bz * sG;
        x += bz * n;
        z += bz * m;
        GTz += bz * n;
        partial += bz * gridDim.x * m;
    }
    if (tid < NT_COLS) dots[tid] = 0.0;
    const int c0 = blockIdx.x * NT_COLS + w * 16;
    const int c1 = min(n, c0 + 16);
    __syncthreads();
    for (int rb = 0; rb < m; rb += NT_ROWS) {
        const int i0 = rb + 2 * lane;
        const bool whole = (rb + NT_ROWS <= m);
        double zr[16], y[16];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + 128 * k;
            zr[2 * k] = (i < m) ? z[i] : 0.0;
            zr[2 * k + 1] = (i + 1 < m) ? z[i + 1] : 0.0;
            y[2 * k] = y[2 * k + 1] = 0.0;
        }
        // (one column per trip: two columns in flight cost a wave of occupancy -- 210 instead of 138 registers -- and were slower: 415 vs 381 us)
        for (int j = c0; j < c1; ++j) {
            const double* __restrict__ g = G + (int64_t)j * ldg + i0;
            const double xj = x[j];
            double a[16];
            if (whole) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const d2u v = *reinterpret_cast<const d2u*>(g + 128 * k);
                    a[2 * k] = v.x;
                    a[2 * k + 1] = v.y;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + 128 * k;
                    a[2 * k] = (i < m) ? g[128 * k] : 0.0;
                    a[2 * k + 1] = (i + 1 < m) ? g[128 * k + 1] : 0.0;
                }
            }
            double d0 = 0.0, d1 = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                y[2 * k] = fma(a[2 * k], xj, y[2 * k]);
                y[2 * k + 1] = fma(a[2 * k + 1], xj, y[2 * k + 1]);
                d0 = fma(a[2 * k], zr[2 * k], d0);
                d1 = fma(a[2 * k + 1], zr[2 * k + 1], d1);
            }
            const double d = wave_sum(d0 + d1);
            if (lane == 0) dots[j - blockIdx.x * NT_COLS] += d;          // (row blocks in order: a fixed summation order)
        }
        // the four waves' G x of this row block -> one partial row per workgroup
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            ylds[w][2 * lane + 128 * k] = y[2 * k];
            ylds[w][2 * lane + 128 * k + 1] = y[2 * k + 1];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int il = tid + 256 * r;
            if (rb + il < m)
                partial[(int64_t)blockIdx.x * m + rb + il] = (ylds[0][il] + ylds[1][il]) + (ylds[2][il] + ylds[3][il]);
        }
        __syncthreads();
    }
    if (tid < NT_COLS && blockIdx.x * NT_COLS + tid < n) GTz[blockIdx.x * NT_COLS + tid] = dots[tid];
}

size_t gemv_nt_work_doubles(int m, int n) {
    const size_t nchunks = (size_t)((n + NT_COLS - 1) / NT_COLS);
    return (nchunks ? nchunks : 1) * (size_t)(m > 0 ? m : 1);
}

// Gx := G x,  GTz := G' z  (both overwritten).  work: >= gemv_nt_work_doubles(m, n) doubles per problem.
int launch_gemv_nt_fused(const double* G, int64_t ldg, int m, int n, const double* x, const double* z, double* Gx, double* GTz,
                         double* work, hipStream_t st, int nbatch, int64_t sG) {
    if (m <= 0 || n <= 0) return 0;
    const int nchunks = (n + NT_COLS - 1) / NT_COLS;
    hipLaunchKernelGGL(gemv_nt_fused_kernel, dim3(nchunks, 1, nbatch), dim3(256), 0, st, G, ldg, m, n, x, z, work, GTz, sG);
    KKT_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(gemv_n_finish_kernel, dim3((m + 255) / 256, 1, nbatch), dim3(256), 0, st, work, nchunks, m, nullptr, Gx, Gx,
                       1.0, 0.0);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ===================================================================================================
// Triangular solves with the Cholesky factor, blocked by 128; the 128x128 diagonal solve runs inside
// ONE wave (lane = row, x_j broadcast with readlane, no LDS, no barriers) in two 64-row halves.
// ===================================================================================================
constexpr int TB = 128;

__device__ __forceinline__ double bcast(double v, int srclane) {   // srclane wave-uniform: v_readlane, no LDS
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// forward substitution on a <=64 x <=64 lower block held column-major at Lb (ldl); lane i owns b_i.
// The lane's row of the block is preloaded into registers (coalesced per column) so that the
// 64-step dependency chain is readlane + fma only.
__device__ __forceinline__ double tri_fwd64(const double* __restrict__ Lb, int64_t ldl, int nb, double b, int lane) {
    double lrow[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) lrow[j] = (j < lane && lane < nb) ? Lb[lane + (int64_t)j * ldl] : 0.0;
    const double dinv = (lane < nb) ? 1.0 / Lb[lane + (int64_t)lane * ldl] : 0.0;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const double xj = bcast(b * dinv, j);
        if (lane == j) b = xj;
        b -= lrow[j] * xj;   // lrow[j] == 0 for lanes <= j and for j >= nb
    }
    return b;
}

// backward substitution with the transpose: solves Lb' x = b (Lb lower); lane i preloads column i.
__device__ __forceinline__ double tri_bwd64(const double* __restrict__ Lb, int64_t ldl, int nb, double b, int lane) {
    double lcol[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) lcol[j] = (j > lane && j < nb) ? Lb[j + (int64_t)lane * ldl] : 0.0;
    const double dinv = (lane < nb) ? 1.0 / Lb[lane + (int64_t)lane * ldl] : 0.0;
#pragma unroll
    for (int j = 63; j >= 0; --j) {
        const double xj = bcast(b * dinv, j);
        if (lane == j) b = xj;
        b -= lcol[j] * xj;
    }
    return b;
}

// Forward step for block row k0: x_k := L_kk^-1 x_k, one wave per right-hand side.  Every global load of
// the 128x128 block (row of L11, L21, L22 per lane) is issued before the first dependent instruction.
__global__ __launch_bounds__(64) void trsv_diag_fwd_kernel(const double* __restrict__ L, int64_t ldl, int k0, int nb,
                                                           double* __restrict__ X, int64_t ldx, int64_t sL,
                                                           int64_t sX) {
    L += (int64_t)blockIdx.z * sL;
    X += (int64_t)blockIdx.z * sX;
    const int lane = threadIdx.x;
    double* x = X + (int64_t)blockIdx.x * ldx + k0;
    const double* Lkk = L + k0 + (int64_t)k0 * ldl;
    const int n0 = min(nb, 64), n1 = nb - n0;
    double r11[64], r21[64], r22[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) r11[j] = (j < lane && lane < n0) ? Lkk[lane + (int64_t)j * ldl] : 0.0;
#pragma unroll
    for (int j = 0; j < 64; ++j) r21[j] = (lane < n1) ? Lkk[64 + lane + (int64_t)j * ldl] : 0.0;
#pragma unroll
    for (int j = 0; j < 64; ++j) r22[j] = (j < lane && lane < n1) ? Lkk[64 + lane + (int64_t)(64 + j) * ldl] : 0.0;
    const double d0 = (lane < n0) ? Lkk[lane + (int64_t)lane * ldl] : 1.0;
    const double d1 = (lane < n1) ? Lkk[64 + lane + (int64_t)(64 + lane) * ldl] : 1.0;
    double b0 = (lane < n0) ? x[lane] : 0.0;
    double b1 = (lane < n1) ? x[64 + lane] : 0.0;
    const double i0 = 1.0 / d0, i1 = 1.0 / d1;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const double xj = bcast(b0 * i0, j);
        if (lane == j) b0 = xj;
        b0 = fma(-r11[j], xj, b0);
        b1 = fma(-r21[j], xj, b1);
    }
    if (lane < n0) x[lane] = b0;
    if (n1 > 0) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const double xj = bcast(b1 * i1, j);
            if (lane == j) b1 = xj;
            b1 = fma(-r22[j], xj, b1);
        }
        if (lane < n1) x[64 + lane] = b1;
    }
}

// x[k0+nb : n) -= L[k0+nb : n, k0 : k0+nb) x_k   (row-parallel, two rows per thread)
__global__ __launch_bounds__(256) void trsv_update_fwd_kernel(const double* __restrict__ L, int64_t ldl, int n, int k0,
                                                              int nb, double* __restrict__ X, int64_t ldx, int64_t sL,
                                                              int64_t sX) {
    L += (int64_t)blockIdx.z * sL;
    X += (int64_t)blockIdx.z * sX;
    __shared__ double xs[TB];
    double* x = X + (int64_t)blockIdx.y * ldx;
    for (int t = threadIdx.x; t < nb; t += blockDim.x) xs[t] = x[k0 + t];
    __syncthreads();
    const int i = k0 + nb + (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    const double* __restrict__ l = L + i + (int64_t)k0 * ldl;
    if (i + 1 < n) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 16
        for (int j = 0; j < nb; ++j) {
            const d2u a = *reinterpret_cast<const d2u*>(l + (int64_t)j * ldl);
            s0 += a.x * xs[j];
            s1 += a.y * xs[j];
        }
        x[i] -= s0;
        x[i + 1] -= s1;
    } else {
        double s0 = 0.0;
        for (int j = 0; j < nb; ++j) s0 += l[(int64_t)j * ldl] * xs[j];
        x[i] -= s0;
    }
}

// Backward step for block k0:  x_k := L_kk^-T ( x_k - L[k0+nb:n, k0:k0+nb)' x[k0+nb:n) )
// stage 1: one wave per column c of the block computes the long dot product (coalesced)
__global__ __launch_bounds__(256) void trsv_dot_bwd_kernel(const double* __restrict__ L, int64_t ldl, int n, int k0,
                                                           int nb, double* __restrict__ X, int64_t ldx, int64_t sL,
                                                           int64_t sX) {
    L += (int64_t)blockIdx.z * sL;
    X += (int64_t)blockIdx.z * sX;
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nb) return;
    double* x = X + (int64_t)blockIdx.y * ldx;
    const double* __restrict__ l = L + (int64_t)(k0 + c) * ldl;
    double s0 = 0.0, s1 = 0.0;
    for (int i = k0 + nb + lane; i < n; i += 128) {
        s0 += l[i] * x[i];
        if (i + 64 < n) s1 += l[i + 64] * x[i + 64];
    }
    const double s = wave_sum(s0 + s1);
    if (lane == 0) x[k0 + c] -= s;
}

__global__ __launch_bounds__(64) void trsv_diag_bwd_kernel(const double* __restrict__ L, int64_t ldl, int k0, int nb,
                                                           double* __restrict__ X, int64_t ldx, int64_t sL,
                                                           int64_t sX) {
    L += (int64_t)blockIdx.z * sL;
    X += (int64_t)blockIdx.z * sX;
    const int lane = threadIdx.x;
    double* x = X + (int64_t)blockIdx.x * ldx + k0;
    const double* Lkk = L + k0 + (int64_t)k0 * ldl;
    const int n0 = min(nb, 64), n1 = nb - n0;
    // lane i owns column i of each sub-block (transpose solve): c22 = L22[j][i], c21 = L21[j][i], c11 = L11[j][i]
    double c11[64], c21[64], c22[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) c22[j] = (j > lane && j < n1) ? Lkk[64 + j + (int64_t)(64 + lane) * ldl] : 0.0;
#pragma unroll
    for (int j = 0; j < 64; ++j) c21[j] = (j < n1 && lane < n0) ? Lkk[64 + j + (int64_t)lane * ldl] : 0.0;
#pragma unroll
    for (int j = 0; j < 64; ++j) c11[j] = (j > lane && j < n0) ? Lkk[j + (int64_t)lane * ldl] : 0.0;
    const double d0 = (lane < n0) ? Lkk[lane + (int64_t)lane * ldl] : 1.0;
    const double d1 = (lane < n1) ? Lkk[64 + lane + (int64_t)(64 + lane) * ldl] : 1.0;
    double b0 = (lane < n0) ? x[lane] : 0.0;
    double b1 = (lane < n1) ? x[64 + lane] : 0.0;
    const double i0 = 1.0 / d0, i1 = 1.0 / d1;
    if (n1 > 0) {
#pragma unroll
        for (int j = 63; j >= 0; --j) {
            const double xj = bcast(b1 * i1, j);
            if (lane == j) b1 = xj;
            b1 = fma(-c22[j], xj, b1);
            b0 = fma(-c21[j], xj, b0);
        }
        if (lane < n1) x[64 + lane] = b1;
    }
#pragma unroll
    for (int j = 63; j >= 0; --j) {
        const double xj = bcast(b0 * i0, j);
        if (lane == j) b0 = xj;
        b0 = fma(-c11[j], xj, b0);
    }
    if (lane < n0) x[lane] = b0;
}

// ===================================================================================================
// A(upper) := A(lower)': the backward solve then streams L' with the same coalesced row-of-a-block-row pattern as the
// forward solve streams L (32 x 32 tiles through LDS; diagonal tiles mirror themselves).
__global__ __launch_bounds__(256) void mirror_lower_kernel(double* __restrict__ A, int64_t lda, int n,
                                                           const TrsvJob* __restrict__ jobs) {
    __shared__ double t[32][33];
    if (jobs) {                          // several matrices, blockIdx.z = job (the wide supernodes of the sparse engine)
        const TrsvJob jb = jobs[blockIdx.z];
        A = const_cast<double*>(jb.L);
        lda = jb.ld;
        n = jb.n;
    }
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bj > bi || bi * 32 >= n) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int i0 = bi * 32, j0 = bj * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + tx, j = j0 + ty + 8 * r;
        t[ty + 8 * r][tx] = (i < n && j < n) ? A[i + (int64_t)j * lda] : 0.0;     // t[jl][il] = A[i][j]
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // write A[j0 + tx][i0 + ty + 8r] = A_lower[i0 + ty + 8r][j0 + tx] = t[tx][ty + 8r]
        const int jr = j0 + tx, ic = i0 + ty + 8 * r;
        if (jr < n && ic < n && ic > jr) A[jr + (int64_t)ic * lda] = t[tx][ty + 8 * r];
    }
}
int launch_mirror_lower(double* A, int64_t lda, int n, hipStream_t st) {
    if (n <= 1) return 0;
    const int nb = (n + 31) / 32;
    hipLaunchKernelGGL(mirror_lower_kernel, dim3(nb, nb), dim3(256), 0, st, A, lda, n, nullptr);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_mirror_lower_jobs(const TrsvJob* d_jobs, int njobs, int nmax, hipStream_t st) {
    if (njobs <= 0 || nmax <= 1) return 0;
    const int nb = (nmax + 31) / 32;
    hipLaunchKernelGGL(mirror_lower_kernel, dim3(nb, nb, njobs), dim3(256), 0, st, nullptr, (int64_t)0, 0, d_jobs);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

// Persistent single-launch triangular solve (one right-hand side): workgroup k owns block row k (forward)
// or block column k (backward) of the 128-blocked factor, streams its off-diagonal blocks while it waits
// for the x blocks it depends on, solves its diagonal block inside the workgroup and publishes x_k.
// Inter-workgroup hand-off (guide G16 form R2, "the data is the flag"): the solved block travels as 256 data-tagged 8-byte
// granules {epoch, 32-bit half of a double}, each written by ONE relaxed agent-scope (write-through) store and polled with
// relaxed agent-scope loads by the thread that needs it -- every 8-byte granule is written and read atomically, so no release
// fence, no acquire fence and no second trip for the payload is needed: a hop costs ~1 us instead of ~4 (flag + two fences +
// reload; that first-round form is gone).  gran[block][256], zeroed once; epochs never repeat.  All nblk <= #CUs workgroups are
// co-resident (256 threads, no LDS pressure); every spin is bounded and a timeout sets *err instead of hanging the GPU.
// ===================================================================================================
typedef unsigned int u32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

typedef unsigned long long u64;
// INV (round 2): the diagonal block is not solved by two 64-step substitution chains (2 x ~2200 clocks of readlane / FMA
// dependencies per hop) but with M = inv(L_kk) from the factorisation (potrf_tiles_kernel) and ONE step of fixed-precision
// iterative refinement, three 128 x 128 matrix-vector products spread over the 256 threads:
//     x0 = M b,   e = b - L_kk x0,   x = x0 + M e        (backward stable like the substitution: Skeel 1980)
// minv: per 128-block 2 x 16384 doubles, M column-major then M' column-major (the backward solve reads rows of M').
// jobs (optional): several independent triangular systems in one launch, blockIdx.y = job (the wide supernodes of one level of
// the sparse engine); every job has its own TRSV_JOB_STRIDE granule blocks.
template <bool TRANS, bool INV>
__global__ __launch_bounds__(256) void trsv_persistent_kernel(const double* __restrict__ L, int64_t ldl, int n,
                                                              double* x, u32 epoch, int* err, u64* gran,
                                                              const double* __restrict__ minv,
                                                              const TrsvJob* __restrict__ jobs) {
    if (jobs) {
        const TrsvJob jb = jobs[blockIdx.y];
        L = jb.L;
        ldl = jb.ld;
        n = jb.n;
        x = jb.x;
        gran += (int64_t)blockIdx.y * TRSV_JOB_STRIDE * 256;
        if ((int)blockIdx.x * TB >= n) return;
    }
    // 256 threads: two per row (forward) / column (backward) of the block row; each holds one 64-wide half of the strip
    // of every off-diagonal block in registers BEFORE waiting for that block's x, so that nothing but 64 FMAs, one
    // partial-sum exchange and the diagonal solve sits between "x_j published" and "x_k published".
    // xs / ps are double buffered: consecutive stages alternate, so no barrier is needed just to make a buffer writable again
    __shared__ double xs2[2][TB];
    __shared__ double ps2[2][TB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = tid >> 7, r = tid & (TB - 1);
    const int nblk = (n + TB - 1) / TB;
    const int k = TRANS ? (nblk - 1 - (int)blockIdx.x) : (int)blockIdx.x;   // dispatch order ~ dependency order
    const int k0 = k * TB;
    const int nb = min(TB, n - k0);
    const int idx = k0 + r;                           // my row (forward) / my column (backward)
    const bool mine = r < nb;
    double acc = (mine && half == 0) ? x[idx] : 0.0;
    // ---- diagonal block operands: issue their loads now (independent of everything), use them at the end
    const double* Lkk = L + k0 + (int64_t)k0 * ldl;
    const int n0 = min(nb, 64), n1 = nb - n0;
    double ra[64], rb[64];
    if (INV) {
        // my half (columns 64 half .. + 63) of row r of M (forward) / of M' (backward) and of the same row of L_kk / L_kk'
        const double* Mk = minv + (int64_t)k * (2 * TB * TB) + (TRANS ? TB * TB : 0);
        const int c0 = 64 * half;
#pragma unroll
        for (int j = 0; j < 64; ++j) ra[j] = Mk[r + (int64_t)(c0 + j) * TB];
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int c = c0 + j;
            const bool in = mine && c < nb && (TRANS ? c >= r : c <= r);      // (the other triangle holds the mirrored copy)
            rb[j] = in ? Lkk[r + (int64_t)c * ldl] : 0.0;
        }
    } else if (!TRANS) {
        if (wave == 0) {   // rows 0..63: row of L11
#pragma unroll
            for (int j = 0; j < 64; ++j) ra[j] = (j < lane && lane < n0) ? Lkk[lane + (int64_t)j * ldl] : 0.0;
        } else if (wave == 1) {   // rows 64..127: row of L21 (ra) and of L22 (rb)
#pragma unroll
            for (int j = 0; j < 64; ++j) ra[j] = (lane < n1) ? Lkk[64 + lane + (int64_t)j * ldl] : 0.0;
#pragma unroll
            for (int j = 0; j < 64; ++j) rb[j] = (j < lane && lane < n1) ? Lkk[64 + lane + (int64_t)(64 + j) * ldl] : 0.0;
        }
    } else {
        if (wave == 1) {   // columns 64..127: column of L22 (solved first)
#pragma unroll
            for (int j = 0; j < 64; ++j) ra[j] = (j > lane && j < n1) ? Lkk[64 + lane + (int64_t)(64 + j) * ldl] : 0.0;
        } else if (wave == 0) {   // columns 0..63: column of L21 (ra) and of L11 (rb)
#pragma unroll
            for (int j = 0; j < 64; ++j) ra[j] = (j < n1 && lane < n0) ? Lkk[lane + (int64_t)(64 + j) * ldl] : 0.0;
#pragma unroll
            for (int j = 0; j < 64; ++j) rb[j] = (j > lane && j < n0) ? Lkk[lane + (int64_t)j * ldl] : 0.0;
        }
    }
    const int dpos = (wave == 0) ? lane : 64 + lane;
    const double dg = (dpos < nb) ? Lkk[dpos + (int64_t)dpos * ldl] : 1.0;
    // ---- off-diagonal blocks, in dependency order
    const int nsteps = TRANS ? (nblk - 1 - k) : k;
    for (int s = 0; s < nsteps; ++s) {
        const int j = TRANS ? (nblk - 1 - s) : s;      // block whose solution we consume
        const int j0 = j * TB;
        const int jb = min(TB, n - j0);
        double* xs = xs2[s & 1];
        // prefetch my half of my strip of block (k,j) before waiting
        double l0[64];
        const int ch = 64 * half;
        if (!TRANS) {
#pragma unroll
            for (int c = 0; c < 64; ++c) l0[c] = (mine && ch + c < jb) ? L[idx + (int64_t)(j0 + ch + c) * ldl] : 0.0;
        } else {
#pragma unroll
            for (int c = 0; c < 64; ++c) l0[c] = (mine && ch + c < jb) ? L[idx + (int64_t)(j0 + ch + c) * ldl] : 0.0;   // mirrored L'

        }
        {
            const u64* g = gran + (int64_t)j * 256 + tid;
            u64 v = 0;
            bool got = false;
            for (unsigned spins = 0; spins < (1u << 22); ++spins) {
                v = __hip_atomic_load(g, RLX_AGENT);
                if ((u32)(v >> 32) == epoch) { got = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            reinterpret_cast<u32*>(xs)[tid] = (u32)v;          // little endian: granule 2i / 2i+1 = low / high word of x_i
            if (__syncthreads_or(got ? 0 : 1)) {
                if (tid == 0) atomicExch(err, 1);
                return;                                     // timeout: give up (err is set)
            }
        }
        {   // four independent chains of 16 instead of one of 64 dependent FMAs (this sits on the hop's critical path)
            double a0 = acc, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
            for (int c = 0; c < 64; c += 4) {
                a0 = fma(-l0[c], xs[ch + c], a0);
                a1 = fma(-l0[c + 1], xs[ch + c + 1], a1);
                a2 = fma(-l0[c + 2], xs[ch + c + 2], a2);
                a3 = fma(-l0[c + 3], xs[ch + c + 3], a3);
            }
            acc = (a0 + a1) + (a2 + a3);
        }
    }
    // the buffer NOT read by the last step: free to write at once
    double* xs = xs2[nsteps & 1];
    double* xo = xs2[(nsteps & 1) ^ 1];
    double* ps = ps2[0];
    double* po = ps2[1];
    // ---- combine the two partial sums of every row / column
    if (half == 1) ps[r] = acc;
    __syncthreads();
    if (half == 0) acc += ps[r];
    // ---- diagonal block (threads 0..127): two 64-wide halves, the first half's solution goes through LDS to the second
    const double dinv = 1.0 / dg;
    if (INV) {
        const int ch2 = 64 * half;
        // 64-term dot product of my register strip with one half of an LDS vector, as four independent chains
        auto dot64 = [&](const double (&a)[64], const double* v) {
            double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
            for (int c = 0; c < 64; c += 4) {
                d0 = fma(a[c], v[ch2 + c], d0);
                d1 = fma(a[c + 1], v[ch2 + c + 1], d1);
                d2 = fma(a[c + 2], v[ch2 + c + 2], d2);
                d3 = fma(a[c + 3], v[ch2 + c + 3], d3);
            }
            return (d0 + d1) + (d2 + d3);
        };
        // x0 = M b
        if (half == 0) xs[r] = acc;                      // b (rows beyond nb carry zeros)
        __syncthreads();
        const double p0 = dot64(ra, xs);
        if (half == 1) po[r] = p0;
        __syncthreads();
        const double x0 = p0 + po[r];                    // (meaningful in half 0)
        // e = b - L x0
        if (half == 0) xo[r] = x0;
        __syncthreads();
        const double q0 = dot64(rb, xo);
        if (half == 1) ps[r] = q0;
        __syncthreads();
        const double e = acc - (q0 + ps[r]);
        // x = x0 + M e
        if (half == 0) xs[r] = e;
        __syncthreads();
        const double p1 = dot64(ra, xs);
        if (half == 1) po[r] = p1;
        __syncthreads();
        acc = x0 + (p1 + po[r]);
    } else if (!TRANS) {
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                const double xj = bcast(acc * dinv, j);
                if (lane == j) acc = xj;
                acc = fma(-ra[j], xj, acc);
            }
            xs[lane] = acc;
        }
        __syncthreads();
        if (wave == 1 && n1 > 0) {
#pragma unroll
            for (int j = 0; j < 64; ++j) acc = fma(-ra[j], xs[j], acc);
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                const double xj = bcast(acc * dinv, j);
                if (lane == j) acc = xj;
                acc = fma(-rb[j], xj, acc);
            }
        }
    } else {
        if (wave == 1 && n1 > 0) {
#pragma unroll
            for (int j = 63; j >= 0; --j) {
                const double xj = bcast(acc * dinv, j);
                if (lane == j) acc = xj;
                acc = fma(-ra[j], xj, acc);
            }
        }
        if (wave == 1) xs[lane] = (n1 > 0) ? acc : 0.0;
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int j = 0; j < 64; ++j) acc = fma(-ra[j], xs[j], acc);
#pragma unroll
            for (int j = 63; j >= 0; --j) {
                const double xj = bcast(acc * dinv, j);
                if (lane == j) acc = xj;
                acc = fma(-rb[j], xj, acc);
            }
        }
    }
    if (mine && half == 0) x[idx] = acc;
    if (half == 0) {          // rows beyond nb publish zeros (consumers mask them anyway): every granule gets its tag
        const double v = mine ? acc : 0.0;
        u64* g = gran + (int64_t)k * 256 + 2 * r;
        const u64 tag = (u64)epoch << 32;
        __hip_atomic_store(g, tag | (u32)__double2loint(v), RLX_AGENT);
        __hip_atomic_store(g + 1, tag | (u32)__double2hiint(v), RLX_AGENT);
    }
}

// ===================================================================================================
// Round 4: the same solve as TWO pipelined sweeps, two workgroups per block row.
//
// trsv_persistent_kernel<.., INV> solves every diagonal block with M = inv(L_kk) plus one step of fixed-precision refinement:
// three dependent matrix-vector stages per hop (x0 = M b, e = b - L_kk x0, x = x0 + M e), 4.2 us per 128 rows of which the
// hand-off is ~1 us.  Here the refinement leaves the hop: the correction is applied GLOBALLY, as a second sweep that follows the
// first one block behind.
//     sweep 1 (workgroup A_k):  b1_k = rhs_k - sum_{j<k} L_kj x0_j,   x0_k = M_k b1_k,   e_k = b1_k - L_kk x0_k
//     sweep 2 (workgroup B_k):  b2_k = e_k  - sum_{j<k} L_kj d_j,     d_k  = M_k b2_k,   x_k = x0_k + d_k
// e is exactly the residual rhs - L x0 (row block k: rhs_k - sum_{j<=k} L_kj x0_j), d the block solve of L d = e with the same
// inverses: x = x0 + d is ONE step of fixed-precision iterative refinement of the whole triangular solve -- backward stable under
// the same condition as the per-block refinement it replaces (Skeel 1980; Higham, Accuracy and Stability, Thm 12.3: the solver
// need only be "not too unstable", here eps cond(L_kk) << 1).  Each chain's hop is hand-off + 64 FMAs + ONE matrix-vector stage;
// B_k needs e_k from A_k (one more hand-off, off the chains' critical paths) and streams the strips of its block row a second
// time (they are in the Infinity Cache / L2: A_k read them a moment ago).  2 nblk co-resident workgroups: orders up to 128 x #CUs / 2.
// Hand-offs are the data-tagged granules of the kernel above: blocks [0, nblk) carry x0, [nblk, 2 nblk) d, [2 nblk, 3 nblk) e.
// Deterministic (fixed summation order); every spin is bounded and a timeout sets *err.
// ===================================================================================================
template <bool TRANS>
__global__ __launch_bounds__(256) void trsv_pair_kernel(const double* __restrict__ L, int64_t ldl, int n, double* x, u32 epoch,
                                                        int* err, u64* gran, const double* __restrict__ minv) {
    __shared__ double xs2[2][TB];
    __shared__ double ps2[2][TB];
    __shared__ double es[TB];
    const int tid = threadIdx.x;
    const int half = tid >> 7, r = tid & (TB - 1);
    const int nblk = n / TB;                                // whole 128-blocks only (checked by the launcher)
    const int role = (int)blockIdx.x & 1;                   // 0: sweep 1 (A_k), 1: sweep 2 (B_k)
    const int pos = (int)blockIdx.x >> 1;
    const int k = TRANS ? (nblk - 1 - pos) : pos;
    const int k0 = k * TB;
    const int idx = k0 + r;                                 // my row (forward) / my column (backward)
    u64* gX = gran;
    u64* gD = gran + (int64_t)nblk * 256;
    u64* gE = gran + (int64_t)2 * nblk * 256;
    const u64* gin = role == 0 ? gX : gD;                   // the solved blocks my far-field products consume
    double acc = (role == 0 && half == 0) ? x[idx] : 0.0;
    // ---- diagonal block operands: issue their loads now (independent of everything), use them at the end
    const double* Lkk = L + k0 + (int64_t)k0 * ldl;
    const double* Mk = minv + (int64_t)k * (2 * TB * TB) + (TRANS ? TB * TB : 0);
    const int c0 = 64 * half;
    double ra[64], rb[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) ra[j] = Mk[r + (int64_t)(c0 + j) * TB];
    if (role == 0) {
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            const int c = c0 + j;
            const bool in = TRANS ? c >= r : c <= r;          // (the other triangle holds the mirrored copy)
            rb[j] = in ? Lkk[r + (int64_t)c * ldl] : 0.0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 64; ++j) rb[j] = 0.0;
    }
    // all 256 threads: the 256 granules of block j of a granule set -> 128 doubles at dst; false on a timeout (err is set)
    auto wait_block = [&](const u64* gbase, int j, double* dst) -> bool {
        const u64* g = gbase + (int64_t)j * 256 + tid;
        u64 v = 0;
        bool got = false;
        for (unsigned spins = 0; spins < (1u << 22); ++spins) {
            v = __hip_atomic_load(g, RLX_AGENT);
            if ((u32)(v >> 32) == epoch) { got = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        reinterpret_cast<u32*>(dst)[tid] = (u32)v;            // little endian: granule 2i / 2i+1 = low / high word of entry i
        if (__syncthreads_or(got ? 0 : 1)) {
            if (tid == 0) atomicExch(err, 1);
            return false;
        }
        return true;
    };
    auto publish = [&](u64* gbase, double v) {                // threads of half 0: entry r of block k
        u64* g = gbase + (int64_t)k * 256 + 2 * r;
        const u64 tag = (u64)epoch << 32;
        __hip_atomic_store(g, tag | (u32)__double2loint(v), RLX_AGENT);
        __hip_atomic_store(g + 1, tag | (u32)__double2hiint(v), RLX_AGENT);
    };
    // ---- off-diagonal blocks, in dependency order
    const int nsteps = TRANS ? (nblk - 1 - k) : k;
    for (int s = 0; s < nsteps; ++s) {
        const int j = TRANS ? (nblk - 1 - s) : s;
        const int j0 = j * TB;
        double* xs = xs2[s & 1];
        double l0[64];
#pragma unroll
        for (int c = 0; c < 64; ++c) l0[c] = L[idx + (int64_t)(j0 + c0 + c) * ldl];     // (backward: the mirrored L')
        if (!wait_block(gin, j, xs)) return;
        double a0 = acc, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int c = 0; c < 64; c += 4) {
            a0 = fma(-l0[c], xs[c0 + c], a0);
            a1 = fma(-l0[c + 1], xs[c0 + c + 1], a1);
            a2 = fma(-l0[c + 2], xs[c0 + c + 2], a2);
            a3 = fma(-l0[c + 3], xs[c0 + c + 3], a3);
        }
        acc = (a0 + a1) + (a2 + a3);
    }
    double* xs = xs2[nsteps & 1];                            // the buffer NOT read by the last step
    double* xo = xs2[(nsteps & 1) ^ 1];
    double* ps = ps2[0];
    double* po = ps2[1];
    // 64-term dot product of my register strip with one half of an LDS vector, as four independent chains
    auto dot64 = [&](const double (&a)[64], const double* v) {
        double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
        for (int c = 0; c < 64; c += 4) {
            d0 = fma(a[c], v[c0 + c], d0);
            d1 = fma(a[c + 1], v[c0 + c + 1], d1);
            d2 = fma(a[c + 2], v[c0 + c + 2], d2);
            d3 = fma(a[c + 3], v[c0 + c + 3], d3);
        }
        return (d0 + d1) + (d2 + d3);
    };
    // ---- combine the two partial sums of every row / column
    if (half == 1) ps[r] = acc;
    __syncthreads();
    if (half == 0) acc += ps[r];
    if (role == 0) {
        // x0 = M b1
        if (half == 0) xs[r] = acc;
        __syncthreads();
        const double p0 = dot64(ra, xs);
        if (half == 1) po[r] = p0;
        __syncthreads();
        const double x0 = p0 + po[r];                        // (meaningful in half 0)
        if (half == 0) {
            publish(gX, x0);                                 // -> A_i, i beyond k, and B_k
            xo[r] = x0;
        }
        __syncthreads();
        // e = b1 - L_kk x0
        const double q0 = dot64(rb, xo);
        if (half == 1) ps[r] = q0;
        __syncthreads();
        if (half == 0) publish(gE, acc - (q0 + ps[r]));       // -> B_k
        return;
    }
    // ---- sweep 2: b2 = e_k - sum L_kj d_j,  d_k = M b2,  x_k = x0_k + d_k
    if (!wait_block(gE, k, es)) return;
    if (half == 0) xs[r] = acc + es[r];
    __syncthreads();
    const double p0 = dot64(ra, xs);
    if (half == 1) po[r] = p0;
    __syncthreads();
    const double dk = p0 + po[r];
    if (half == 0) publish(gD, dk);                          // -> B_i, i beyond k
    if (!wait_block(gX, k, es)) return;                      // (published by A_k before e_k: there since long)
    if (half == 0) x[idx] = es[r] + dk;
}

int launch_trsv_pair(const double* L, int64_t ldl, int n, double* x, int trans, unsigned int epoch, int* err, hipStream_t st,
                     unsigned long long* gran, const double* minv) {
    if (n <= 0 || n % TB || !gran || !minv) return -1;
    const dim3 g(2 * (n / TB)), b(256);
    if (trans)
        hipLaunchKernelGGL((trsv_pair_kernel<true>), g, b, 0, st, L, ldl, n, x, epoch, err, gran, minv);
    else
        hipLaunchKernelGGL((trsv_pair_kernel<false>), g, b, 0, st, L, ldl, n, x, epoch, err, gran, minv);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_trsv_persistent(const double* L, int64_t ldl, int n, double* x, int trans, unsigned int epoch, int* err,
                           hipStream_t st, unsigned long long* gran, const double* minv, const TrsvJob* jobs, int njobs) {
    const int nblk = (n + TB - 1) / TB;      // with jobs: n = the largest order among them
    if (nblk <= 0) return 0;
    if (!gran || (jobs && (njobs <= 0 || nblk > TRSV_JOB_STRIDE))) return -1;
    const dim3 g(nblk, jobs ? njobs : 1), b(256);
    if (minv) {
        if (trans)
            hipLaunchKernelGGL((trsv_persistent_kernel<true, true>), g, b, 0, st, L, ldl, n, x, epoch, err, gran, minv, jobs);
        else
            hipLaunchKernelGGL((trsv_persistent_kernel<false, true>), g, b, 0, st, L, ldl, n, x, epoch, err, gran, minv, jobs);
    } else if (trans)
        hipLaunchKernelGGL((trsv_persistent_kernel<true, false>), g, b, 0, st, L, ldl, n, x, epoch, err, gran, minv, jobs);
    else
        hipLaunchKernelGGL((trsv_persistent_kernel<false, false>), g, b, 0, st, L, ldl, n, x, epoch, err, gran, minv, jobs);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_trsm_lower(const double* L, int64_t ldl, int n, double* X, int64_t ldx, int nrhs, int trans,
                      hipStream_t st, int nbatch, int64_t sL, int64_t sX) {
    if (n <= 0 || nrhs <= 0) return 0;
    if (!trans) {
        for (int k0 = 0; k0 < n; k0 += TB) {
            const int nb = (n - k0 < TB) ? (n - k0) : TB;
            hipLaunchKernelGGL(trsv_diag_fwd_kernel, dim3(nrhs, 1, nbatch), dim3(64), 0, st, L, ldl, k0, nb, X, ldx, sL, sX);
            const int rem = n - k0 - nb;
            if (rem > 0)
                hipLaunchKernelGGL(trsv_update_fwd_kernel, dim3((rem + 127) / 128, nrhs, nbatch), dim3(64), 0, st, L, ldl,
                                   n, k0, nb, X, ldx, sL, sX);
        }
    } else {
        const int nblk = (n + TB - 1) / TB;
        for (int kb = nblk - 1; kb >= 0; --kb) {
            const int k0 = kb * TB;
            const int nb = (n - k0 < TB) ? (n - k0) : TB;
            if (n - k0 - nb > 0)
                hipLaunchKernelGGL(trsv_dot_bwd_kernel, dim3((nb + 3) / 4, nrhs, nbatch), dim3(256), 0, st, L, ldl, n, k0,
                                   nb, X, ldx, sL, sX);
            hipLaunchKernelGGL(trsv_diag_bwd_kernel, dim3(nrhs, 1, nbatch), dim3(64), 0, st, L, ldl, k0, nb, X, ldx, sL, sX);
        }
    }
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mi355kkt

// Dense FP64 Cholesky (lower, in place) for gfx950: the device replacement of lapack.potrf
// (reference src/C/lapack.c:1471-1523 -> dpotrf_, called from misc.py:1282, :1429, :1460, :1472).
//
// Blocked right-looking factorisation with NB = 128 column panels:
//   potf2_kernel      one workgroup, LDS-resident 128x128 diagonal block: 16-column micro panels
//                     (rank-1 updates inside the micro panel) + FP64-MFMA rank-16 updates of the rest
//   trsm_panel_kernel X L_kk' = B for the rows below the diagonal block (one row per lane)
//   nt_update_kernel  trailing update A22 -= L21 L21' on the matrix cores (gemm_f64.hip)
// A non-positive pivot sets *info = (1-based column) exactly like LAPACK's info > 0; every later
// kernel of the sequence returns immediately once *info != 0.
#include "kkt_common.h"

namespace mi355kkt {

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

constexpr int NB = 128;
constexpr int PLD = 144;   // LDS leading dimension of the diagonal block (== 16 mod 32: conflict-free frags)

__device__ __forceinline__ double readlane_d(double v, int srclane) {   // srclane: wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// d = sqrt(p), inv = 1/sqrt(p) to ~1 ulp from one v_rsq_f64 + two coupled Newton steps (p > 0, normal)
__device__ __forceinline__ void sqrt_rsqrt(double p, double& d, double& inv) {
    const double y0 = __builtin_amdgcn_rsq(p);
    double g = p * y0, h = 0.5 * y0;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-g, g, p);          // final correction of sqrt
    g = fma(r, h, g);
    d = g;
    inv = 2.0 * h;
}

// Diagonal block factorisation, one workgroup, block resident in LDS.  Per 16-column micro panel:
//   (a) wave 0 factors the 16x16 diagonal block in registers (lane = row, pivots/multipliers broadcast
//       with v_readlane: no LDS round trip, no barrier on the 16-step dependency chain)
//   (b) one thread per row below solves x L_d' = r (L_d read as LDS broadcasts)
//   (c) all four waves apply the rank-16 update to the remaining columns with v_mfma_f64_16x16x4_f64
// linv_out[blk][k][g] = inv(L_d)[g][k] (16x16 diagonal blocks, zero upper) is exported for trsm_panel_kernel.
__global__ __launch_bounds__(256) void potf2_kernel(double* __restrict__ A, int64_t lda, int nb, int col0,
                                                    int* __restrict__ info, double* __restrict__ linv_out,
                                                    int64_t bstride) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    A += (int64_t)blockIdx.z * bstride;          // batched problems along blockIdx.z
    info += blockIdx.z;
    if (linv_out) linv_out += (int64_t)blockIdx.z * 2048;
    double* As = smem;                       // NB x PLD, column-major, lower triangle valid
    double* Ld = smem + NB * PLD;            // 16 x 16 current diagonal block, Ld[c * 16 + k] = L[c][k]
    double* dinv = Ld + 256;                 // 16 reciprocal pivots of the current micro panel
    int* flag = reinterpret_cast<int*>(dinv + 16);
    if (*info != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) *flag = 0;
    {   // block -> LDS, 16 independent loads in flight per thread (the whole nb x nb square; only tril is used)
        const int r = tid & (NB - 1), c0 = tid >> 7;
        const int rr = min(r, nb - 1);
        for (int cc = 0; cc < nb; cc += 32) {
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = A[rr + (int64_t)min(cc + c0 + 2 * i, nb - 1) * lda];
#pragma unroll
            for (int i = 0; i < 16; ++i) As[(cc + c0 + 2 * i) * PLD + r] = v[i];
        }
    }
    __syncthreads();
    for (int jb = 0; jb < nb; jb += 16) {
        const int pw = min(16, nb - jb);
        // ---- (a) 16x16 diagonal block in wave 0, lane i <-> row jb+i
        if (wave == 0) {
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = (lane < pw && c <= lane) ? As[(jb + c) * PLD + jb + lane] : ((c == lane) ? 1.0 : 0.0);
            int bad = 0;
            double dv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double p = readlane_d(a[j], j);
                if (!(p > 0.0) && j < pw && bad == 0) bad = j + 1;   // uniform (p is wave-uniform)
                double d, inv;
                sqrt_rsqrt(bad ? 1.0 : p, d, inv);
                const double l = a[j] * inv;
                a[j] = (lane == j) ? d : l;
                dv[j] = inv;
                if (lane == 0) dinv[j] = inv;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) {
                    const double lc = readlane_d(l, c);
                    a[c] = fma(-l, lc, a[c]);
                }
            }
            if (bad) {
                if (lane == 0) *flag = jb + bad;
            } else {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (lane < pw && c <= lane) {
                        As[(jb + c) * PLD + jb + lane] = a[c];
                        Ld[lane * 16 + c] = a[c];
                    }
                }
                // inverse of the 16x16 diagonal block (lower), for the MFMA triangular solves:
                // row-oriented forward recurrence, rows of M broadcast with v_readlane.
                double mrow[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) mrow[j] = (j == lane) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const double dk = dv[k];
                    const double lik = (lane > k) ? a[k] : 0.0;
#pragma unroll
                    for (int j = 0; j <= k; ++j) {
                        const double mkj = readlane_d(mrow[j], k) * dk;
                        mrow[j] = (lane == k) ? mkj : fma(-lik, mkj, mrow[j]);
                    }
                }
                if (linv_out && lane < 16) {
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        linv_out[(jb / 16) * 256 + j * 16 + lane] = (j <= lane && lane < pw) ? mrow[j] : 0.0;
                }
            }
        }
        __syncthreads();
        if (*flag) break;
        // ---- (b) rows below the diagonal block: x L_d' = r, one row per thread
        {
            const int row = jb + 16 + tid;
            if (row < nb) {
                double x[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) x[c] = As[(jb + c) * PLD + row];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    x[k] *= dinv[k];
#pragma unroll
                    for (int c = k + 1; c < 16; ++c) x[c] = fma(-x[k], Ld[c * 16 + k], x[c]);
                }
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < pw) As[(jb + c) * PLD + row] = x[c];
            }
        }
        __syncthreads();
        // ---- (c) rank-16 update of columns >= jb+16 on the matrix cores (16x16 tiles, rt >= ct)
        const int t0 = jb / 16 + 1, nt = (nb + 15) / 16;
        const int ntr = nt - t0;
        const int ntiles = ntr * (ntr + 1) / 2;
        const int li = lane & 15, lq = lane >> 4;
        for (int t = wave; t < ntiles; t += 4) {
            int a = 0, rem = t;                  // t -> (ct = t0 + a, rt = ct + rem), column-major triangle
            while (rem >= ntr - a) {
                rem -= ntr - a;
                ++a;
            }
            const int ct = t0 + a, rt = ct + rem;
            d4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = As[(ct * 16 + lq + 4 * r) * PLD + rt * 16 + li];
#pragma unroll
            for (int kk = 0; kk < 16; kk += 4) {
                const double av = -As[(jb + kk + lq) * PLD + ct * 16 + li];
                const double bv = As[(jb + kk + lq) * PLD + rt * 16 + li];
                acc = MFMA_F64(av, bv, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) As[(ct * 16 + lq + 4 * r) * PLD + rt * 16 + li] = acc[r];
        }
        __syncthreads();
    }
    if (*flag) {
        if (tid == 0) *info = col0 + *flag;
        return;
    }
    {
        const int r = tid & (NB - 1), c0 = tid >> 7;
        for (int cc = 0; cc < nb; cc += 32) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = cc + c0 + 2 * i;
                if (r < nb && c < nb && r >= c) A[r + (int64_t)c * lda] = As[c * PLD + r];
            }
        }
    }
}

// X L' = B for the rows below a full 128x128 diagonal block, entirely on the matrix cores.
// One wave owns a strip of 16 rows; tiles are kept transposed (MFMA row index = column of X, MFMA
// column index = row of the strip), so that a solved tile's D registers are *directly* the B operand
// of the next products (f64 D layout: row = (lane>>4) + 4 reg  <->  B operand: k = 4 step + (lane>>4)).
// Per 16-column block cb:   R = B_cb - sum_{c<cb} X_c L[cb,c]'                  (4 cb MFMAs)
//   diagonal block by inverse + one step of fixed-precision iterative refinement (backward stable,
//   Skeel 1980):  X0 = R Linv';  E = R - X0 Ld';  X = X0 + E Linv'                (12 MFMAs)
constexpr int TRSM_ROWS = 64;
__global__ __launch_bounds__(256) void trsm_panel_kernel(const double* __restrict__ L,
                                                         const double* __restrict__ linv,
                                                         double* __restrict__ B, int64_t lda, int mrows,
                                                         const int* __restrict__ info, int64_t bstride) {
    L += (int64_t)blockIdx.z * bstride;
    B += (int64_t)blockIdx.z * bstride;
    linv += (int64_t)blockIdx.z * 2048;
    info += blockIdx.z;
    if (*info != 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int row0 = blockIdx.x * TRSM_ROWS + wave * 16;
    if (row0 >= mrows) return;
    const int row = min(row0 + li, mrows - 1);
    const bool active = row0 + li < mrows;
    d4 x[8];
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) {
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = B[row + (int64_t)(16 * cb + lq + 4 * r) * lda];
#pragma unroll
        for (int c = 0; c < cb; ++c) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const double av = -L[(16 * cb + li) + (int64_t)(16 * c + 4 * s4 + lq) * lda];
                acc = MFMA_F64(av, x[c][s4], acc);
            }
        }
        double mi[4], ld[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            mi[s4] = linv[cb * 256 + (4 * s4 + lq) * 16 + li];                       // Linv_d[li][4s+lq]
            const double lv = L[(16 * cb + li) + (int64_t)(16 * cb + 4 * s4 + lq) * lda];
            ld[s4] = (4 * s4 + lq <= li) ? -lv : 0.0;                                // -L_d[li][4s+lq], tril only
        }
        d4 x0 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) x0 = MFMA_F64(mi[s4], acc[s4], x0);
        d4 e = acc;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) e = MFMA_F64(ld[s4], x0[s4], e);
        d4 xx = x0;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) xx = MFMA_F64(mi[s4], e[s4], xx);
        x[cb] = xx;
        if (active) {
#pragma unroll
            for (int r = 0; r < 4; ++r) B[row + (int64_t)(16 * cb + lq + 4 * r) * lda] = xx[r];
        }
    }
}

int potrf_work_init_batched(PotrfWork& w, int nbatch) {
    KKT_HIP_CHECK(hipMalloc(&w.d_info, sizeof(int) * nbatch));
    KKT_HIP_CHECK(hipMalloc(&w.d_dinv, sizeof(double) * 8 * 256 * nbatch));   // inverses of the 16x16 diagonal blocks
    KKT_HIP_CHECK(hipHostMalloc(&w.h_info, sizeof(int) * nbatch));
    memset(w.h_info, 0, sizeof(int) * nbatch);
    return 0;
}

int potrf_work_init(PotrfWork& w) { return potrf_work_init_batched(w, 1); }

void potrf_work_free(PotrfWork& w) {
    if (w.d_info) (void)hipFree(w.d_info);
    if (w.d_dinv) (void)hipFree(w.d_dinv);
    if (w.h_info) (void)hipHostFree(w.h_info);
    w = PotrfWork();
}

int launch_potrf_batched(double* A, int64_t lda, int n, int nbatch, int64_t bstride, PotrfWork& w, hipStream_t st) {
    static bool attr_set = false;
    constexpr size_t lds = sizeof(double) * (NB * PLD + 256 + 16) + 16;
    if (!attr_set) {
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    KKT_HIP_CHECK(hipMemsetAsync(w.d_info, 0, sizeof(int) * nbatch, st));
    // Outer panels of 256 columns = two 128-column sub-panels; the trailing matrix is touched once per
    // outer panel with a rank-256 update (halves the C read-modify-write traffic of a rank-128 scheme).
    auto panel = [&](int k0, int nb) -> int {   // factor diagonal block at k0 and solve the rows below it
        double* Akk = A + k0 + (int64_t)k0 * lda;
        hipLaunchKernelGGL(potf2_kernel, dim3(1, 1, nbatch), dim3(256), lds, st, Akk, lda, nb, k0, w.d_info, w.d_dinv,
                           bstride);
        KKT_HIP_CHECK(hipGetLastError());
        const int m = n - k0 - nb;
        if (m > 0) {
            hipLaunchKernelGGL(trsm_panel_kernel, dim3((m + TRSM_ROWS - 1) / TRSM_ROWS, 1, nbatch), dim3(256), 0, st, Akk,
                               w.d_dinv, Akk + nb, lda, m, w.d_info, bstride);
            KKT_HIP_CHECK(hipGetLastError());
        }
        return 0;
    };
    for (int k0 = 0; k0 < n; k0 += 2 * NB) {
        const int nb1 = (n - k0 < NB) ? (n - k0) : NB;
        if (int e = panel(k0, nb1)) return e;
        const int k1 = k0 + nb1;
        if (k1 >= n) break;
        const int nb2 = (n - k1 < NB) ? (n - k1) : NB;
        // columns k1 .. k1+nb2 of the trailing matrix get the rank-nb1 update now (needed by the 2nd sub-panel)
        if (int e = launch_gemm_nt_update(A + k1 + (int64_t)k1 * lda, lda, A + k1 + (int64_t)k0 * lda, lda,
                                          A + k1 + (int64_t)k0 * lda, lda, n - k1, nb2, nb1, st, nbatch, bstride))
            return e;
        if (int e = panel(k1, nb2)) return e;
        const int k2 = k1 + nb2;
        if (k2 >= n) break;
        // rank-(nb1+nb2) update of everything to the right of the outer panel
        if (int e = launch_syrk_nt_update(A + k2 + (int64_t)k2 * lda, lda, A + k2 + (int64_t)k0 * lda, lda, n - k2,
                                          nb1 + nb2, st, nbatch, bstride))
            return e;
    }
    return 0;
}

int launch_potrf(double* A, int64_t lda, int n, PotrfWork& w, hipStream_t st) {
    return launch_potrf_batched(A, lda, n, 1, 0, w, st);
}

}  // namespace mi355kkt

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

__global__ __launch_bounds__(256) void potf2_kernel(double* __restrict__ A, int64_t lda, int nb, int col0,
                                                    int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double As[];
    if (*info != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < NB * PLD; e += 256) As[e] = 0.0;
    __syncthreads();
    for (int e = tid; e < nb * NB; e += 256) {
        const int r = e & (NB - 1), c = e >> 7;
        if (r < nb && r >= c) As[c * PLD + r] = A[r + (int64_t)c * lda];
    }
    int failed = 0;
    for (int jb = 0; jb < nb && !failed; jb += 16) {
        const int pw = min(16, nb - jb);
        // ---- micro panel: columns jb .. jb+pw-1, rows jb .. nb-1
        for (int jj = 0; jj < pw; ++jj) {
            const int j = jb + jj;
            __syncthreads();
            const double ajj = As[j * PLD + j];
            if (!(ajj > 0.0)) {   // also catches NaN (LAPACK dpotf2: `ajj <= 0 .or. disnan(ajj)`)
                failed = j + 1;
                break;
            }
            const double d = sqrt(ajj);
            const double inv = 1.0 / d;   // dpotf2: DSCAL by ONE/AJJ
            if (tid < nb - j - 1) As[j * PLD + j + 1 + tid] *= inv;
            __syncthreads();
            if (tid == 0) As[j * PLD + j] = d;
            const int ncols = jb + pw - 1 - j;
            for (int e = tid; e < ncols * NB; e += 256) {
                const int c = j + 1 + (e >> 7), r = e & (NB - 1);
                if (r >= c && r < nb) As[c * PLD + r] -= As[j * PLD + r] * As[j * PLD + c];
            }
        }
        if (failed) break;
        __syncthreads();
        // ---- rank-16 update of columns >= jb+16 on the matrix cores (16x16 tiles, rt >= ct)
        const int t0 = jb / 16 + 1, nt = (nb + 15) / 16;
        const int ntr = nt - t0;                 // tiles per side of the trailing triangle
        const int ntiles = ntr * (ntr + 1) / 2;
        const int li = lane & 15, lq = lane >> 4;
        for (int t = wave; t < ntiles; t += 4) {
            int a = 0, rem = t;                  // t -> (ct = t0 + a, rt = ct + b), column-major triangle
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
    }
    __syncthreads();
    if (failed) {
        if (tid == 0) *info = col0 + failed;
        return;
    }
    for (int e = tid; e < nb * NB; e += 256) {
        const int r = e & (NB - 1), c = e >> 7;
        if (r < nb && r >= c) A[r + (int64_t)c * lda] = As[c * PLD + r];
    }
}

// X L' = B, one row of B per lane; L (nb x nb lower) is read with wave-uniform addresses.
__global__ __launch_bounds__(64) void trsm_panel_kernel(const double* __restrict__ L, double* __restrict__ B,
                                                        int64_t lda, int mrows, int nb,
                                                        const int* __restrict__ info) {
    if (*info != 0) return;
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= mrows) return;
    double* __restrict__ b = B + row;
    for (int cb = 0; cb < nb; cb += 16) {
        const int cw = min(16, nb - cb);
        double x[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) x[c] = (c < cw) ? b[(int64_t)(cb + c) * lda] : 0.0;
        for (int k = 0; k < cb; ++k) {
            const double xk = b[(int64_t)k * lda];
            const double* __restrict__ lk = L + (int64_t)k * lda + cb;   // L[cb + c][k]
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < cw) x[c] -= xk * lk[c];
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < cw) {
#pragma unroll
                for (int k2 = 0; k2 < c; ++k2) x[c] -= x[k2] * L[(int64_t)(cb + k2) * lda + cb + c];
                x[c] /= L[(int64_t)(cb + c) * lda + cb + c];
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < cw) b[(int64_t)(cb + c) * lda] = x[c];
    }
}

int potrf_work_init(PotrfWork& w) {
    KKT_HIP_CHECK(hipMalloc(&w.d_info, sizeof(int)));
    KKT_HIP_CHECK(hipHostMalloc(&w.h_info, sizeof(int)));
    *w.h_info = 0;
    return 0;
}

void potrf_work_free(PotrfWork& w) {
    if (w.d_info) (void)hipFree(w.d_info);
    if (w.h_info) (void)hipHostFree(w.h_info);
    w = PotrfWork();
}

int launch_potrf(double* A, int64_t lda, int n, PotrfWork& w, hipStream_t st) {
    static bool attr_set = false;
    constexpr size_t lds = sizeof(double) * NB * PLD;
    if (!attr_set) {
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    KKT_HIP_CHECK(hipMemsetAsync(w.d_info, 0, sizeof(int), st));
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nb = (n - k0 < NB) ? (n - k0) : NB;
        double* Akk = A + k0 + (int64_t)k0 * lda;
        hipLaunchKernelGGL(potf2_kernel, dim3(1), dim3(256), lds, st, Akk, lda, nb, k0, w.d_info);
        KKT_HIP_CHECK(hipGetLastError());
        const int m = n - k0 - nb;
        if (m > 0) {
            double* panel = Akk + nb;
            hipLaunchKernelGGL(trsm_panel_kernel, dim3((m + 63) / 64), dim3(64), 0, st, Akk, panel, lda, m, nb,
                               w.d_info);
            KKT_HIP_CHECK(hipGetLastError());
            if (int e = launch_syrk_nt_update(Akk + nb + (int64_t)nb * lda, lda, panel, lda, m, nb, st)) return e;
        }
    }
    return 0;
}

}  // namespace mi355kkt

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
#include <cstdlib>

#include "kkt_common.h"

namespace mi355kkt {

typedef double d4 __attribute__((ext_vector_type(4)));
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

__device__ int g_potf2_skip = 0;   // developer ablation switch (bit0 a, bit1 b, bit2 c, bit3 inverses); 0 in production
int set_potf2_skip(int v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_potf2_skip), &v, sizeof(int)) == hipSuccess ? 0 : -2; }

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
                                                    int64_t bstride, const VbDesc* __restrict__ vb) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (vb) {                                    // variable batched fronts: col0 carries the panel offset k0
        const VbDesc dd = vb[blockIdx.z];
        const int k0 = col0;
        if (k0 >= dd.w) return;
        nb = min(NB, dd.w - k0);
        lda = dd.h;
        A += dd.off + k0 + (int64_t)k0 * lda;
        col0 = dd.col0 + k0;
    } else {
        A += (int64_t)blockIdx.z * bstride;      // batched problems along blockIdx.z
    }
    info += blockIdx.z;
    if (linv_out) linv_out += (int64_t)blockIdx.z * 2048;
    double* As = smem;                       // NB x PLD, column-major, lower triangle valid
    double* Ld = smem + NB * PLD;            // 16 x 16 current diagonal block, Ld[c * 16 + k] = L[c][k]
    double* dinv = Ld + 256;                 // 16 reciprocal pivots of the current micro panel
    int* flag = reinterpret_cast<int*>(dinv + 16);
    if (*info != 0) return;
    const int skip = g_potf2_skip;
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
        // ---- (a) 16x16 diagonal block in wave 0, lane i <-> row jb + (i & 15)  (lanes >= 16 mirror lanes 0..15).
        //      Written without per-element predicates: the strict upper triangle of the block carries garbage that
        //      never reaches the lower triangle (each update only mixes entries of one row).
        if (wave == 0 && !(skip & 1)) {
            const int l15 = lane & 15;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = As[(jb + c) * PLD + jb + l15];
            if (pw < 16) {                       // ragged last block: rows >= pw act as identity rows
#pragma unroll
                for (int c = 0; c < 16; ++c) a[c] = (l15 < pw) ? ((c < pw) ? a[c] : 0.0) : ((c == l15) ? 1.0 : 0.0);
            }
            int bad = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                const double p = readlane_d(a[j], j);
                if (!(p > 0.0) && bad == 0) bad = j + 1;             // uniform (p is wave-uniform)
                double d, inv;
                sqrt_rsqrt(bad ? 1.0 : p, d, inv);
                const double l = a[j] * inv;                         // lane j: p / sqrt(p) = L[j][j]
                a[j] = l;
                if (lane == 0) dinv[j] = inv;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) {
                    a[c] = fma(-l, readlane_d(l, c), a[c]);
                    if (((c - j) & 3) == 0) __builtin_amdgcn_sched_barrier(0);   // <= 4 broadcast values live in SGPRs
                }
            }
            if (bad) {
                if (lane == 0) *flag = jb + bad;
            } else if (lane < pw) {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    if (c < pw) As[(jb + c) * PLD + jb + lane] = a[c];
            }
        }
        __syncthreads();
        if (*flag) break;
        // ---- (b) rows below the diagonal block: x L_d' = r, one row per thread.  Column k+1 of L_d is
        //      prefetched from LDS (wave-wide broadcast reads) into a second register set while column k is
        //      applied, so the FMAs never wait on an LDS round trip.
        if (jb + 16 < nb && wave * 64 < nb - jb - 16 && !(skip & 2)) {
            const int row = jb + 16 + tid;
            const int rr = min(row, NB - 1);
            double x[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) x[c] = As[(jb + c) * PLD + rr];
            double colA[16], colB[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) colA[c] = As[jb * PLD + jb + c];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                double(&cur)[16] = (k & 1) ? colB : colA;
                double(&nxt)[16] = (k & 1) ? colA : colB;
                if (k < 15) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) nxt[c] = As[(jb + k + 1) * PLD + jb + c];
                }
                x[k] *= dinv[k];
#pragma unroll
                for (int c = k + 1; c < 16; ++c) x[c] = fma(-x[k], cur[c], x[c]);
            }
            if (row < nb) {
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
        for (int t = wave; t < ((skip & 4) ? 0 : ntiles); t += 4) {
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
    // inverses of the 16x16 diagonal blocks for the MFMA triangular solves (trsm_panel_kernel): row-oriented
    // forward recurrence, rows of M broadcast with v_readlane; blocks are independent -> two per wave.
    if (linv_out && !(skip & 8)) {
        for (int jb = wave * 16; jb < nb; jb += 64) {
            const int pw = min(16, nb - jb);
            const int l15 = lane & 15;
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const double v = As[(jb + c) * PLD + jb + l15];
                a[c] = (c < l15 && l15 < pw) ? v : 0.0;                  // strictly lower part of row l15
            }
            const double myinv = (l15 < pw) ? 1.0 / As[(jb + l15) * PLD + jb + l15] : 1.0;
            double mrow[16];                                             // unscaled row: e_i - sum_k L[i][k] M[k][:]
#pragma unroll
            for (int j = 0; j < 16; ++j) mrow[j] = (j == l15) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < 15; ++k) {
                __builtin_amdgcn_sched_barrier(0);
                const double dk = readlane_d(myinv, k);
#pragma unroll
                for (int j = 0; j <= k; ++j) {
                    mrow[j] = fma(-a[k], readlane_d(mrow[j], k) * dk, mrow[j]);
                    if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (lane < 16) {
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    linv_out[(jb / 16) * 256 + j * 16 + lane] = (lane < pw) ? mrow[j] * myinv : 0.0;
            }
        }
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

// ---------------------------------------------------------------------------------------------------
// potf2_la_kernel: the same diagonal-block factorisation restructured around its dependency chain (round 2).
// 512 threads.  Wave 0 is the *panel wave*: it holds the current 16-column micro panel of ALL remaining rows in
// registers (lane l <-> rows jb + l and jb + 64 + l), so the factorisation of the 16x16 diagonal block and the
// triangular solve of every row below it are ONE instruction stream - the multipliers of column j are broadcast
// with v_readlane once and applied to both row sets, no LDS round trip and no barrier inside the 16 columns.
// 1/sqrt(pivot) comes from v_rsq_f64 + one third-order (Halley) step: 4 dependent FP64 operations instead of 11.
// Waves 1..7 apply the rank-16 updates on the matrix cores one step behind (look-ahead): while the panel wave
// works on micro panel jb they finish the update of micro panel jb - 16 on the columns right of jb + 16; only the
// tiles of column block jb + 16 (<= 7, one per wave) sit between two panel phases.  The panel wave writes finished
// columns straight to global memory; the 16x16 inverses for trsm_panel_kernel are formed at the end, one wave per
// block (column-oriented forward substitution with LDS broadcasts).
// ---------------------------------------------------------------------------------------------------
constexpr int P2T = 512;

// rank-16 update of one 16x16 tile (ct, rt >= ct) of the LDS-resident block with micro panel jb
__device__ __forceinline__ void potf2_tile_update(double* __restrict__ As, int jb, int ct, int rt, int lane) {
    const int li = lane & 15, lq = lane >> 4;
    d4 acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = As[(ct * 16 + lq + 4 * r) * PLD + rt * 16 + li];
    double av[4], bv[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        av[s4] = -As[(jb + 4 * s4 + lq) * PLD + ct * 16 + li];
        bv[s4] = As[(jb + 4 * s4 + lq) * PLD + rt * 16 + li];
    }
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) acc = MFMA_F64(av[s4], bv[s4], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) As[(ct * 16 + lq + 4 * r) * PLD + rt * 16 + li] = acc[r];
}

// 16 columns of the panel wave: a[c] / b[c] = rows (jb + lane) / (jb + 64 + lane), column jb + c.
// Returns 0 or the 1-based index of the first non-positive pivot.  dinv16[j] (LDS) = 1 / L[jb+j][jb+j].
// 1/sqrt(p) for a normal p > 0: v_rsq_f64 (~23 good bits) + one third-order step -> ~1 ulp, 4 dependent operations
__device__ __forceinline__ double rsqrt_halley(double p) {
    const double y0 = __builtin_amdgcn_rsq(p);
    const double t = p * y0;
    const double e = fma(-t, y0, 1.0);                         // 1 - p y0^2
    const double q = fma(0.375, e, 0.5);
    const double w = y0 * e;
    return fma(w, q, y0);                                      // y0 (1 + e/2 + 3 e^2 / 8)
}

// The scalar recurrence of the pivots runs ahead of the vector updates: with x = A[j+1][j], y = A[j+1][j+1] (both final
// before column j is scaled) the next pivot is y - (x inv_j)^2 -- bit-identical to what the vector update of lane j+1
// produces (fma(-t, t, y) with t = x inv_j) -- so the chain per column is  mul, fma, rsq, 4 Halley operations,
// all on wave-uniform values, and the readlane broadcasts + FMAs of the column update fill its latency.
// A non-positive pivot is recorded (first one wins) and the arithmetic simply continues (NaN/Inf stay in this block,
// the caller discards it).
template <bool TWO>
__device__ __forceinline__ int potf2_panel16(double (&a)[16], double (&b)[16], double* __restrict__ dinv16,
                                              double* __restrict__ dummy, int lane) {
    double* __restrict__ dst = (lane == 0) ? dinv16 : dummy + lane;
    const double p0 = readlane_d(a[0], 0);
    int badv = (p0 > 0.0) ? 0 : 1;
    double inv = rsqrt_halley(p0);
#define P2_SB __builtin_amdgcn_sched_barrier(0)
#define P2_UPD(c_)                                                         \
    if ((c_) < 16) {                                                       \
        const double s_ = readlane_d(la, (c_) < 16 ? (c_) : 15);           \
        a[(c_) & 15] = fma(-la, s_, a[(c_) & 15]);                         \
        if (TWO) b[(c_) & 15] = fma(-lb, s_, b[(c_) & 15]);                \
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        // one operation of the scalar chain, then one column update, pinned in that order: the in-order issue then
        // hides every link of the chain under the broadcasts + FMAs of an update
        double x = 0.0, y = 1.0;
        if (j < 15) {
            x = readlane_d(a[j], j + 1);
            y = readlane_d(a[j + 1], j + 1);
        }
        dst[j] = inv;                                          // lane 0 -> dinv16[j]; the other lanes hit a dummy area (no branch)
        const double la = a[j] * inv;
        a[j] = la;
        double lb = 0.0;
        if (TWO) {
            lb = b[j] * inv;
            b[j] = lb;
        }
        const double t = x * inv;
        P2_SB;
        P2_UPD(j + 1)
        const double pn = fma(-t, t, y);
        P2_SB;
        P2_UPD(j + 2)
        const double y0 = __builtin_amdgcn_rsq(pn);
        badv = (j < 15 && !(pn > 0.0) && badv == 0) ? j + 2 : badv;
        P2_SB;
        P2_UPD(j + 3)
        const double tt = pn * y0;
        P2_SB;
        P2_UPD(j + 4)
        const double e = fma(-tt, y0, 1.0);                    // 1 - p y0^2
        P2_SB;
        P2_UPD(j + 5)
        const double q = fma(0.375, e, 0.5);
        const double w = y0 * e;
        P2_SB;
        P2_UPD(j + 6)
        const double invn = fma(w, q, y0);                     // y0 (1 + e/2 + 3 e^2 / 8)
        P2_SB;
#pragma unroll
        for (int c = j + 7; c < 16; ++c) { P2_UPD(c) }
        inv = invn;
        // pin the finished column here: otherwise LLVM sinks the whole second row set into the (conditional) stores of the
        // caller and keeps all 120 broadcast values alive in spilled SGPRs
        asm volatile("" : "+v"(a[j]));
        if (TWO) asm volatile("" : "+v"(b[j]));
        P2_SB;
    }
#undef P2_UPD
#undef P2_SB
    return __builtin_amdgcn_readfirstlane(badv);
}

__global__ __launch_bounds__(P2T) void potf2_la_kernel(double* __restrict__ A, int64_t lda, int nb, int col0,
                                                       int* __restrict__ info, double* __restrict__ linv_out,
                                                       int64_t bstride, const VbDesc* __restrict__ vb) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (vb) {                                    // variable batched fronts: col0 carries the panel offset k0
        const VbDesc dd = vb[blockIdx.z];
        const int k0 = col0;
        if (k0 >= dd.w) return;
        nb = min(NB, dd.w - k0);
        lda = dd.h;
        A += dd.off + k0 + (int64_t)k0 * lda;
        col0 = dd.col0 + k0;
    } else {
        A += (int64_t)blockIdx.z * bstride;
    }
    info += blockIdx.z;
    if (linv_out) linv_out += (int64_t)blockIdx.z * 2048;
    double* As = smem;                           // NB x PLD, column-major
    double* dinv = smem + NB * PLD;              // 128 reciprocal pivots
    double* dummy = dinv + NB;                   // 80 doubles: sink of the non-leader lanes' reciprocal-pivot stores
    int* flag = reinterpret_cast<int*>(dummy + 80);
    if (*info != 0) return;
    const int tid = threadIdx.x, wave = tid >> 6;
    int lane = tid & 63;
    const int nt = (nb + 15) >> 4;
    if (tid == 0) *flag = 0;
    double a[16], b[16];
    if (wave == 0) {                             // micro panel 0 straight from global memory into the panel wave
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            a[c] = (lane < nb && c < nb) ? A[lane + (int64_t)c * lda] : 0.0;
            b[c] = (lane + 64 < nb && c < nb) ? A[lane + 64 + (int64_t)c * lda] : 0.0;
        }
    } else {                                     // columns 16.. -> LDS (tiles on or below the diagonal), 32 elements per thread
        const int tt = tid - 64;
        double v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int e = tt + (P2T - 64) * i;
            const int r = e & (NB - 1), c = 16 + (e >> 7);
            v[i] = (r < nb && c < nb && r >= (c & ~15)) ? A[r + (int64_t)c * lda] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int e = tt + (P2T - 64) * i;
            const int r = e & (NB - 1), c = 16 + (e >> 7);
            As[c * PLD + r] = v[i];
        }
    }
    for (int jb = 0; jb < nb; jb += 16) {
        const int pw = min(16, nb - jb);
        asm volatile("" : "+v"(lane));           // opaque per iteration: the per-lane predicates below stay out of (spilled) SGPRs
        if (wave == 0) {
            if (jb > 0) {
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    a[c] = As[(jb + c) * PLD + min(jb + lane, NB - 1)];
                    b[c] = As[(jb + c) * PLD + min(jb + 64 + lane, NB - 1)];
                }
            }
            if (pw < 16 && lane < 16) {          // ragged last block: rows >= pw of the diagonal block act as identity rows
#pragma unroll
                for (int c = 0; c < 16; ++c) a[c] = (lane < pw) ? ((c < pw) ? a[c] : 0.0) : ((c == lane) ? 1.0 : 0.0);
            }
            const bool two = nb - jb > 64;       // wave-uniform
            const int bad = two ? potf2_panel16<true>(a, b, dinv + jb, dummy, lane) : potf2_panel16<false>(a, b, dinv + jb, dummy, lane);
            if (bad) {
                if (lane == 0) *flag = jb + bad;
            } else {
                const int r1 = jb + lane, r2 = jb + 64 + lane;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    if (r1 < NB) As[(jb + c) * PLD + r1] = a[c];
                    if (two && r2 < NB) As[(jb + c) * PLD + r2] = b[c];
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) {   // finished columns -> global memory (lower triangle only)
                    if (c < pw) {
                        if (r1 < nb && lane >= c) A[r1 + (int64_t)(jb + c) * lda] = a[c];
                        if (two && r2 < nb) A[r2 + (int64_t)(jb + c) * lda] = b[c];
                    }
                }
            }
        } else if (jb >= 16) {                   // look-ahead: the rest of the previous micro panel's update
            const int pjb = jb - 16, t1 = pjb / 16 + 2;          // column blocks t1 .. nt-1
            const int ntr = nt - t1;
            const int ntiles = ntr > 0 ? ntr * (ntr + 1) / 2 : 0;
            for (int t = wave - 1; t < ntiles; t += 7) {
                int aa = 0, rem = t;
                while (rem >= ntr - aa) {
                    rem -= ntr - aa;
                    ++aa;
                }
                potf2_tile_update(As, pjb, t1 + aa, t1 + aa + rem, lane);
            }
        }
        __syncthreads();                         // micro panel jb is in LDS; update jb-16 is complete
        if (*flag) break;
        if (jb + 16 >= nb) break;
        {   // column block jb/16 + 1 (the next micro panel): one tile per wave
            const int t0 = jb / 16 + 1;
            if (t0 + wave < nt) potf2_tile_update(As, jb, t0, t0 + wave, lane);
        }
        __syncthreads();
    }
    if (*flag) {
        if (tid == 0) *info = col0 + *flag;
        return;
    }
    // inverses of the 16x16 diagonal blocks (trsm_panel_kernel): wave w <-> block w, lane j <-> column j of inv(L_d)
    if (linv_out && wave < nt) {
        const int jb = wave * 16, pw = min(16, nb - jb);
        const int j = lane & 15;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; ++k) s = fma(-As[(jb + k) * PLD + jb + i], x[k], s);
            x[i] = s * dinv[jb + i];
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) linv_out[wave * 256 + j * 16 + i] = (i < pw && j < pw && i >= j) ? x[i] : 0.0;
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
template <bool full>     // full: nb == 128 (every panel of a dense potrf); ragged panels only occur in sparse fronts
__global__ __launch_bounds__(256) void trsm_panel_kernel(const double* __restrict__ L,
                                                         const double* __restrict__ linv,
                                                         double* __restrict__ B, int64_t lda, int mrows,
                                                         const int* __restrict__ info, int64_t bstride, int nb,
                                                         const VbDesc* __restrict__ vb) {
    if (vb) {                                    // variable batched fronts: mrows carries the panel offset k0; L = B = base
        const VbDesc dd = vb[blockIdx.z];
        const int k0 = mrows;
        if (k0 >= dd.w) return;
        nb = min(128, dd.w - k0);
        lda = dd.h;
        L += dd.off + k0 + (int64_t)k0 * lda;
        B += dd.off + k0 + nb + (int64_t)k0 * lda;
        mrows = dd.h - k0 - nb;
    } else {
        L += (int64_t)blockIdx.z * bstride;
        B += (int64_t)blockIdx.z * bstride;
    }
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
        if (!full && 16 * cb >= nb) break;           // ragged panels (sparse fronts): fewer 16-column blocks
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 16 * cb + lq + 4 * r;
            acc[r] = (full || col < nb) ? B[row + (int64_t)col * lda] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < cb; ++c) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const double lv = (full || 16 * cb + li < nb) ? L[(16 * cb + li) + (int64_t)(16 * c + 4 * s4 + lq) * lda] : 0.0;
                acc = MFMA_F64(-lv, x[c][s4], acc);
            }
        }
        double mi[4], ld[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            mi[s4] = linv[cb * 256 + (4 * s4 + lq) * 16 + li];                       // Linv_d[li][4s+lq]
            const bool in = (4 * s4 + lq <= li) && (full || 16 * cb + li < nb);
            const double lv = in ? L[(16 * cb + li) + (int64_t)(16 * cb + 4 * s4 + lq) * lda] : 0.0;
            ld[s4] = -lv;                                                            // -L_d[li][4s+lq], tril only
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
            for (int r = 0; r < 4; ++r) {
                const int col = 16 * cb + lq + 4 * r;
                if (full || col < nb) B[row + (int64_t)col * lda] = xx[r];
            }
        }
    }
}

// One diagonal-block factorisation launch (nz blocks along blockIdx.z).  MI355KKT_POTF2=old selects the round-1 kernel.
static int launch_potf2(double* A, int64_t lda, int nb, int col0, int* info, double* linv, int64_t bstride,
                        const VbDesc* vb, int nz, hipStream_t st) {
    static const bool use_old = getenv("MI355KKT_POTF2") && !strcmp(getenv("MI355KKT_POTF2"), "old");
    static bool attr_set = false;
    constexpr size_t lds_old = sizeof(double) * (NB * PLD + 256 + 16) + 16;
    constexpr size_t lds_la = sizeof(double) * (NB * PLD + NB + 80) + 16;
    if (!attr_set) {
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_old));
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_la_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_la));
        attr_set = true;
    }
    if (use_old)
        hipLaunchKernelGGL(potf2_kernel, dim3(1, 1, nz), dim3(256), lds_old, st, A, lda, nb, col0, info, linv, bstride, vb);
    else
        hipLaunchKernelGGL(potf2_la_kernel, dim3(1, 1, nz), dim3(P2T), lds_la, st, A, lda, nb, col0, info, linv, bstride, vb);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

int potrf_work_init_batched(PotrfWork& w, int nbatch) {
    KKT_HIP_CHECK(hipMalloc(&w.d_info, sizeof(int) * nbatch));
    KKT_HIP_CHECK(hipMalloc(&w.d_dinv, sizeof(double) * 8 * 256 * nbatch));   // inverses of the 16x16 diagonal blocks
    KKT_HIP_CHECK(hipHostMalloc(&w.h_info, sizeof(int) * nbatch));
    memset(w.h_info, 0, sizeof(int) * nbatch);
    {   // the bulk updates yield to the critical-path kernels of the main stream at workgroup granularity
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&w.side, hipStreamNonBlocking, least) != hipSuccess)
            KKT_HIP_CHECK(hipStreamCreateWithFlags(&w.side, hipStreamNonBlocking));
        if (hipStreamCreateWithPriority(&w.aux, hipStreamNonBlocking, greatest) != hipSuccess)
            KKT_HIP_CHECK(hipStreamCreateWithFlags(&w.aux, hipStreamNonBlocking));
    }
    return 0;
}

int potrf_work_init(PotrfWork& w) { return potrf_work_init_batched(w, 1); }

void potrf_work_free(PotrfWork& w) {
    if (w.d_info) (void)hipFree(w.d_info);
    if (w.d_dinv) (void)hipFree(w.d_dinv);
    if (w.h_info) (void)hipHostFree(w.h_info);
    for (auto e : w.ev_panel) (void)hipEventDestroy(e);
    for (auto e : w.ev_bulk) (void)hipEventDestroy(e);
    for (auto e : w.ev_t1) (void)hipEventDestroy(e);
    for (auto e : w.ev_usr) (void)hipEventDestroy(e);
    for (auto e : w.ev_ir) (void)hipEventDestroy(e);
    if (w.side) (void)hipStreamDestroy(w.side);
    if (w.aux) (void)hipStreamDestroy(w.aux);
    w = PotrfWork();
}

int launch_potrf_batched(double* A, int64_t lda, int n, int nbatch, int64_t bstride, PotrfWork& w, hipStream_t st) {
    KKT_HIP_CHECK(hipMemsetAsync(w.d_info, 0, sizeof(int) * nbatch, st));
    // Outer panels of 256 columns = two 128-column sub-panels; the trailing matrix is touched once per
    // outer panel with a rank-256 update (halves the C read-modify-write traffic of a rank-128 scheme).
    auto panel = [&](int k0, int nb) -> int {   // factor diagonal block at k0 and solve the rows below it
        double* Akk = A + k0 + (int64_t)k0 * lda;
        if (int e = launch_potf2(Akk, lda, nb, k0, w.d_info, w.d_dinv, bstride, nullptr, nbatch, st)) return e;
        const int m = n - k0 - nb;
        if (m > 0) {
            if (nb == NB)
                hipLaunchKernelGGL(trsm_panel_kernel<true>, dim3((m + TRSM_ROWS - 1) / TRSM_ROWS, 1, nbatch), dim3(256), 0, st,
                                   Akk, w.d_dinv, Akk + nb, lda, m, w.d_info, bstride, nb, nullptr);
            else
                hipLaunchKernelGGL(trsm_panel_kernel<false>, dim3((m + TRSM_ROWS - 1) / TRSM_ROWS, 1, nbatch), dim3(256), 0, st,
                                   Akk, w.d_dinv, Akk + nb, lda, m, w.d_info, bstride, nb, nullptr);
            KKT_HIP_CHECK(hipGetLastError());
        }
        return 0;
    };
    // ---- look-ahead variant (single large matrix): the update of the NEXT outer panel's columns stays on
    //      `st`; the rest of the trailing update runs on w.side, concurrently with the next panel's
    //      potf2 / trsm (which occupy only a few compute units).
    const char* mode_env = getenv("MI355KKT_POTRF_STREAMS");
    const int lookahead_streams = mode_env ? atoi(mode_env) : 2;
    if (nbatch == 1 && n >= 8 * NB && w.side && w.aux && lookahead_streams == 3) {
        // Three streams.  st (critical path): potf2 -> trsm -> diagonal-block-only updates -> next potf2 ...
        // w.aux: the rest of the skinny updates (rows below the next diagonal block) - overlaps with the next potf2.
        // w.side (low priority, one workgroup per CU): the bulk rank-256 update of everything further right.
        const int nsteps = (n + 2 * NB - 1) / (2 * NB);
        auto grow = [&](std::vector<hipEvent_t>& v) -> int {
            while ((int)v.size() < nsteps + 1) {
                hipEvent_t e1;
                KKT_HIP_CHECK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
                v.push_back(e1);
            }
            return 0;
        };
        if (int e = grow(w.ev_panel)) return e;
        if (int e = grow(w.ev_bulk)) return e;
        if (int e = grow(w.ev_t1)) return e;
        if (int e = grow(w.ev_usr)) return e;
        if (int e = grow(w.ev_ir)) return e;
        auto potf2 = [&](int k0, int nb) {
            (void)launch_potf2(A + k0 + (int64_t)k0 * lda, lda, nb, k0, w.d_info, w.d_dinv, (int64_t)0, nullptr, 1, st);
        };
        auto trsm = [&](int k0, int nb) {
            const int m = n - k0 - nb;
            if (m <= 0) return;
            double* Akk = A + k0 + (int64_t)k0 * lda;
            if (nb == NB)
                hipLaunchKernelGGL(trsm_panel_kernel<true>, dim3((m + TRSM_ROWS - 1) / TRSM_ROWS), dim3(256), 0, st, Akk, w.d_dinv,
                                   Akk + nb, lda, m, w.d_info, (int64_t)0, nb, nullptr);
            else
                hipLaunchKernelGGL(trsm_panel_kernel<false>, dim3((m + TRSM_ROWS - 1) / TRSM_ROWS), dim3(256), 0, st, Akk, w.d_dinv,
                                   Akk + nb, lda, m, w.d_info, (int64_t)0, nb, nullptr);
        };
        // C[r0:r0+M, c0:c0+N] -= L[r0:.., kp:kp+K] L[c0:.., kp:kp+K]'
        auto upd = [&](int r0, int M, int c0, int N, int kp, int K, hipStream_t s_) -> int {
            if (M <= 0 || N <= 0) return 0;
            return launch_gemm_nt_update(A + r0 + (int64_t)c0 * lda, lda, A + r0 + (int64_t)kp * lda, lda,
                                         A + c0 + (int64_t)kp * lda, lda, M, N, K, s_);
        };
        int step = 0;
        bool bulk_pending = false, ir_pending = false;
        for (int k0 = 0; k0 < n; k0 += 2 * NB, ++step) {
            const int nb1 = (n - k0 < NB) ? (n - k0) : NB;
            potf2(k0, nb1);
            if (ir_pending) KKT_HIP_CHECK(hipStreamWaitEvent(st, w.ev_ir[step - 1], 0));   // rows below were updated on aux
            trsm(k0, nb1);
            const int k1 = k0 + nb1;
            if (k1 >= n) break;
            const int nb2 = (n - k1 < NB) ? (n - k1) : NB;
            KKT_HIP_CHECK(hipEventRecord(w.ev_t1[step], st));
            // sub-panel 2: its diagonal block on the critical path, the rows below it on aux
            if (int e = upd(k1, nb2, k1, nb2, k0, nb1, st)) return e;
            KKT_HIP_CHECK(hipStreamWaitEvent(w.aux, w.ev_t1[step], 0));
            if (int e = upd(k1 + nb2, n - k1 - nb2, k1, nb2, k0, nb1, w.aux)) return e;
            KKT_HIP_CHECK(hipEventRecord(w.ev_usr[step], w.aux));
            potf2(k1, nb2);
            KKT_HIP_CHECK(hipStreamWaitEvent(st, w.ev_usr[step], 0));
            trsm(k1, nb2);
            const int k2 = k1 + nb2;
            if (k2 >= n) { ir_pending = false; break; }
            KKT_HIP_CHECK(hipEventRecord(w.ev_panel[step], st));          // outer panel `step` complete
            const int K = nb1 + nb2;
            const int wnext = (n - k2 < 2 * NB) ? (n - k2) : 2 * NB;      // width of the next outer panel
            const int d1 = (wnext < NB) ? wnext : NB;                     // its first diagonal block
            if (bulk_pending) {
                KKT_HIP_CHECK(hipStreamWaitEvent(st, w.ev_bulk[step - 1], 0));
                KKT_HIP_CHECK(hipStreamWaitEvent(w.aux, w.ev_bulk[step - 1], 0));
            }
            // next panel: first diagonal block on the critical path, all rows below it (both sub-panels' columns) on aux
            if (int e = upd(k2, d1, k2, d1, k0, K, st)) return e;
            KKT_HIP_CHECK(hipStreamWaitEvent(w.aux, w.ev_panel[step], 0));
            if (int e = upd(k2 + d1, n - k2 - d1, k2, wnext, k0, K, w.aux)) return e;
            KKT_HIP_CHECK(hipEventRecord(w.ev_ir[step], w.aux));
            ir_pending = true;
            // everything to the right of the next panel: bulk, low priority, one workgroup per CU
            const int k3 = k2 + wnext;
            if (k3 < n) {
                // released after the skinny next-panel update on aux (which then shares the machine only with potf2)
                KKT_HIP_CHECK(hipStreamWaitEvent(w.side, getenv("MI355KKT_BULK_EARLY") ? w.ev_panel[step] : w.ev_ir[step], 0));
                if (int e = launch_syrk_nt_update(A + k3 + (int64_t)k3 * lda, lda, A + k3 + (int64_t)k0 * lda, lda, n - k3, K,
                                                  w.side, 1, 0, getenv("MI355KKT_BULK_1WG") != nullptr))
                    return e;
                KKT_HIP_CHECK(hipEventRecord(w.ev_bulk[step], w.side));
                bulk_pending = true;
            } else {
                bulk_pending = false;
            }
        }
        // join: nothing may be in flight on aux / side when the caller's stream continues
        hipEvent_t ej = w.ev_t1[nsteps];
        KKT_HIP_CHECK(hipEventRecord(ej, w.aux));
        KKT_HIP_CHECK(hipStreamWaitEvent(st, ej, 0));
        hipEvent_t eb = w.ev_usr[nsteps];
        KKT_HIP_CHECK(hipEventRecord(eb, w.side));
        KKT_HIP_CHECK(hipStreamWaitEvent(st, eb, 0));
        KKT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    static const int outer1_max_n = getenv("MI355KKT_POTRF_OUTER1") ? atoi(getenv("MI355KKT_POTRF_OUTER1")) : 2048;
    if (nbatch == 1 && n >= 8 * NB && w.side && lookahead_streams == 2) {
        // Outer panels of two 128-column sub-panels while the trailing matrix is large (rank-256 bulk updates touch it half
        // as often), of ONE sub-panel once at most outer1_max_n columns remain: there the chain of panel kernels dominates
        // and potf2 + trsm + one K=128 skinny update per 128 columns is the shorter chain.
        const int nsteps = (n + NB - 1) / NB + 1;
        while ((int)w.ev_panel.size() < nsteps + 1) {
            hipEvent_t e1, e2;
            KKT_HIP_CHECK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
            KKT_HIP_CHECK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
            w.ev_panel.push_back(e1);
            w.ev_bulk.push_back(e2);
        }
        int step = 0;
        bool bulk_pending = false;
        auto width_at = [&](int k) { return (n - k > outer1_max_n) ? 2 * NB : NB; };   // outer-panel width that starts at column k
        for (int k0 = 0; k0 < n; ++step) {
            const bool two = width_at(k0) == 2 * NB;
            const int nb1 = (n - k0 < NB) ? (n - k0) : NB;
            if (int e = panel(k0, nb1)) return e;
            int k2 = k0 + nb1;
            if (k2 >= n) break;
            if (two) {
                const int k1 = k2;
                const int nb2 = (n - k1 < NB) ? (n - k1) : NB;
                if (int e = launch_gemm_nt_update(A + k1 + (int64_t)k1 * lda, lda, A + k1 + (int64_t)k0 * lda, lda,
                                                  A + k1 + (int64_t)k0 * lda, lda, n - k1, nb2, nb1, st))
                    return e;
                if (int e = panel(k1, nb2)) return e;
                k2 = k1 + nb2;
                if (k2 >= n) break;
            }
            const int K = k2 - k0;
            int wnext = width_at(k2);
            if (wnext > n - k2) wnext = n - k2;
            const double* Lp = A + k2 + (int64_t)k0 * lda;                // rows k2.., panel columns
            // the previous bulk update wrote the region both updates below touch
            if (bulk_pending) KKT_HIP_CHECK(hipStreamWaitEvent(st, w.ev_bulk[step - 1], 0));
            // (i) next panel's columns, on the main stream
            if (int e = launch_gemm_nt_update(A + k2 + (int64_t)k2 * lda, lda, Lp, lda, Lp, lda, n - k2, wnext, K, st))
                return e;
            // panel `step` complete.  The bulk update is released only now, after the skinny update above: alone on the
            // machine that update takes ~50 us, sharing every CU with the bulk kernel ~170 us (potrf 10.3 -> 9.6 ms at n = 8192).
            KKT_HIP_CHECK(hipEventRecord(w.ev_panel[step], st));
            // (ii) everything to the right of it, on the side stream
            const int k3 = k2 + wnext;
            if (k3 < n) {
                KKT_HIP_CHECK(hipStreamWaitEvent(w.side, w.ev_panel[step], 0));
                if (int e = launch_syrk_nt_update(A + k3 + (int64_t)k3 * lda, lda, A + k3 + (int64_t)k0 * lda, lda, n - k3, K,
                                                  w.side))
                    return e;
                KKT_HIP_CHECK(hipEventRecord(w.ev_bulk[step], w.side));
                bulk_pending = true;
            } else {
                bulk_pending = false;
            }
            k0 = k2;
        }
        if (bulk_pending) KKT_HIP_CHECK(hipStreamWaitEvent(st, w.ev_bulk[step > 0 ? step - 1 : 0], 0));
        return 0;
    }
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

// Partial factorisation of a frontal matrix: eliminate the first `ncols` columns of the h x h matrix F (lower), leaving
// the Schur complement in F[ncols:, ncols:].  Same kernels as launch_potrf; *w.d_info is NOT reset here.
int launch_potrf_partial(double* F, int64_t ld, int h, int ncols, PotrfWork& w, hipStream_t st) {
    for (int k0 = 0; k0 < ncols; k0 += NB) {
        const int nb = (ncols - k0 < NB) ? (ncols - k0) : NB;
        double* Fkk = F + k0 + (int64_t)k0 * ld;
        if (int e = launch_potf2(Fkk, ld, nb, k0, w.d_info, w.d_dinv, (int64_t)0, nullptr, 1, st)) return e;
        const int m = h - k0 - nb;
        if (m > 0) {
            if (nb == NB)
                hipLaunchKernelGGL(trsm_panel_kernel<true>, dim3((m + TRSM_ROWS - 1) / TRSM_ROWS), dim3(256), 0, st, Fkk,
                                   w.d_dinv, Fkk + nb, ld, m, w.d_info, (int64_t)0, nb, nullptr);
            else
                hipLaunchKernelGGL(trsm_panel_kernel<false>, dim3((m + TRSM_ROWS - 1) / TRSM_ROWS), dim3(256), 0, st, Fkk,
                                   w.d_dinv, Fkk + nb, ld, m, w.d_info, (int64_t)0, nb, nullptr);
            if (int e = launch_syrk_nt_update(Fkk + nb + (int64_t)nb * ld, ld, Fkk + nb, ld, m, nb, st)) return e;
        }
    }
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_potrf_partial_vb(double* base, const VbDesc* d_desc, int nfronts, int maxh, int maxw, PotrfWork& w,
                            hipStream_t st) {
    if (nfronts <= 0) return 0;
    KKT_HIP_CHECK(hipMemsetAsync(w.d_info, 0, sizeof(int) * nfronts, st));
    for (int k0 = 0; k0 < maxw; k0 += NB) {
        if (int e = launch_potf2(base, (int64_t)0, 0, k0, w.d_info, w.d_dinv, (int64_t)0, d_desc, nfronts, st)) return e;
        const int mmax = maxh - k0 - 1;           // a front's panel may be as narrow as one column
        if (mmax > 0) {
            hipLaunchKernelGGL(trsm_panel_kernel<false>, dim3((mmax + TRSM_ROWS - 1) / TRSM_ROWS, 1, nfronts), dim3(256), 0, st,
                               base, w.d_dinv, base, (int64_t)0, k0, w.d_info, (int64_t)0, 0, d_desc);
            if (int e = launch_syrk_nt_update_vb(base, d_desc, nfronts, k0, maxh, st)) return e;
        }
    }
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_potrf(double* A, int64_t lda, int n, PotrfWork& w, hipStream_t st) {
    return launch_potrf_batched(A, lda, n, 1, 0, w, st);
}

}  // namespace mi355kkt

// Dense FP64 Cholesky (lower, in place) for gfx950: the device replacement of lapack.potrf
// (reference src/C/lapack.c:1471-1523 -> dpotrf_, called from misc.py:1282, :1429, :1460, :1472).
//
// n >= 1024: ONE persistent left-looking tile kernel (potrf_tiles_kernel, below).  Smaller matrices, batches and the sparse
// engine's small fronts: blocked right-looking factorisation with NB = 128 column panels:
//   potf2_la_kernel   one workgroup, LDS-resident 128x128 diagonal block: 16-column micro panels held by a panel wave,
//                     FP64-MFMA rank-16 updates of the rest one step behind (look-ahead)
//   trsm_panel_kernel X L_kk' = B for the rows below the diagonal block (one row per lane)
//   nt_update_kernel  trailing update A22 -= L21 L21' on the matrix cores (gemm_f64.hip)
// A non-positive pivot sets *info = (1-based column) exactly like LAPACK's info > 0; every later
// kernel of the sequence returns immediately once *info != 0.
#include <cstdlib>

#include "kkt_common.h"
#include <type_traits>

// The fence-free hand-offs of the tile kernel (st_wt / ld_l2 + s_waitcnt, DESIGN 4a') rest on gfx942 / gfx950 behaviour: agent-scope
// relaxed stores are write-through past the XCD's L2 (sc1), agent-scope relaxed loads bypass it, and a wave's stores are
// acknowledged in issue order (vmcnt).  Other targets must not compile this file silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__)
#error "potrf.hip: the streaming hand-off protocol is written for gfx942 / gfx950 only"
#endif

namespace mi355kkt {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2_ __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(8))) d2u_ { double x, y; };   // 8-byte aligned pair (one dwordx4 load)
#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

constexpr int NB = 128;
constexpr int PLD = 144;   // LDS leading dimension of the diagonal block (== 16 mod 32: conflict-free frags)

__device__ __forceinline__ double readlane_d(double v, int srclane) {   // srclane: wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------------
// potf2_la_kernel: the diagonal-block factorisation, structured around its dependency chain (round 2).
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
// Write-through (sc1) store / L1-bypassing (sc1) load of a double: data handed from one compute unit to another inside a launch
// WITHOUT fences -- the producer drains its sc1 stores (s_waitcnt vmcnt(0)) before a relaxed agent-scope flag store, the consumer
// polls the flag and reads with sc1 loads (guide G16: "sc1 stores and loads both sides").  A release fence would write back every
// dirty line of the XCD's L2 (~2-6 us with a freshly written tile), an acquire fence costs ~1.7 us: too much for 16-column steps.
__device__ __forceinline__ void st_wt(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_l2(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// developer aid (-DMI355KKT_DEBUG builds only, include/mi355kkt_debug.h): when set, wave 0 / lane 0 logs s_memtime at phase boundaries
#ifdef MI355KKT_DEBUG
__device__ long long* g_potf2_ts = nullptr;
#define POTF2_TS_PTR g_potf2_ts
#else
#define POTF2_TS_PTR ((long long*)nullptr)
#endif
#define P2_TS(i_) do { if (ts && tid == 0) ts[(i_)] = (long long)__builtin_readcyclecounter(); } while (0)

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
// all on wave-uniform values.  Column j is written to its final place in LDS as soon as it is scaled; the multipliers
// L[jb+c][jb+j] of the columns c >= j+3 are then read back as LDS broadcasts (one ds_read per value, no SGPR hop), only
// the two columns the scalar chain needs next (j+1, j+2) take the v_readlane path.  (Measured on gfx950: the all-readlane
// version issues ~45 instructions per column = 277 clocks; this one ~27.)
// A non-positive pivot is recorded (first one wins) and the arithmetic simply continues (NaN/Inf stay in this block,
// the caller discards it).  col = LDS address of element (row jb, column jb); rowa / rowb = row offsets of the lane's
// two rows relative to jb (rows past the block are redirected to the padding rows 128..143 of the column).
template <bool TWO>
__device__ __forceinline__ int potf2_panel16(double (&a)[16], double (&b)[16], double* __restrict__ col, int rowa, int rowb,
                                              double* __restrict__ dinv16, double* __restrict__ dummy, int lane) {
    double* __restrict__ dst = (lane == 0) ? dinv16 : dummy + lane;
    const double p0 = readlane_d(a[0], 0);
    int badv = (p0 > 0.0) ? 0 : 1;
    double inv = rsqrt_halley(p0);
#define P2_SB __builtin_amdgcn_sched_barrier(0)
#define P2_UPD_RL(c_)                                                      \
    if ((c_) < 16) {                                                       \
        const double s_ = readlane_d(la, (c_) < 16 ? (c_) : 15);           \
        a[(c_) & 15] = fma(-la, s_, a[(c_) & 15]);                         \
        if (TWO) b[(c_) & 15] = fma(-lb, s_, b[(c_) & 15]);                \
    }
#define P2_UPD_LDS(c_)                                                     \
    if ((c_) < 16) {                                                       \
        a[(c_) & 15] = fma(-la, sv[(c_) & 15], a[(c_) & 15]);              \
        if (TWO) b[(c_) & 15] = fma(-lb, sv[(c_) & 15], b[(c_) & 15]);     \
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        double x = 0.0, y = 1.0;
        if (j < 15) {
            x = readlane_d(a[j], j + 1);
            y = readlane_d(a[j + 1], j + 1);
        }
        dst[j] = inv;                                          // lane 0 -> dinv16[j]; the other lanes hit a dummy area (no branch)
        const double la = a[j] * inv;
        double lb = 0.0;
        col[j * PLD + rowa] = la;                              // column j is final: to its place in the LDS block
        if (TWO) {
            lb = b[j] * inv;
            col[j * PLD + rowb] = lb;
        }
        double sv[16];
#pragma unroll
        for (int c = j + 3; c < 16; ++c) sv[c] = col[j * PLD + c];   // wave-uniform addresses: LDS broadcasts (in order behind the store)
        // order pinned: the two readlane-path columns and the whole scalar chain first (they run while the LDS broadcasts
        // are in flight), then the LDS-path columns
        const double t = x * inv;
        P2_SB;
        P2_UPD_RL(j + 1)
        const double pn = fma(-t, t, y);
        P2_SB;
        P2_UPD_RL(j + 2)
        const double y0 = __builtin_amdgcn_rsq(pn);
        badv = (j < 15 && !(pn > 0.0) && badv == 0) ? j + 2 : badv;
        const double tt = pn * y0;
        const double e = fma(-tt, y0, 1.0);                    // 1 - p y0^2
        const double q = fma(0.375, e, 0.5);
        const double w = y0 * e;
        const double invn = fma(w, q, y0);                     // y0 (1 + e/2 + 3 e^2 / 8)
        P2_SB;
#pragma unroll
        for (int c = j + 3; c < 16; ++c) { P2_UPD_LDS(c) }
        inv = invn;
        P2_SB;
    }
#undef P2_UPD_RL
#undef P2_UPD_LDS
#undef P2_SB
    return __builtin_amdgcn_readfirstlane(badv);
}

// The diagonal-block factorisation proper.  FROM_LDS: the block (lower triangle, column-major, leading dimension PLD)
// already sits at the start of `smem` (the persistent tile kernel dumps its accumulators there); otherwise it is read
// from A.  L goes to A (lower triangle only), the inverses of the 16x16 diagonal blocks to linv_out (may be null).
// Returns 0 or the 1-based column of the first non-positive pivot (uniform over the workgroup).  All P2T threads call it.
// half_word (tile kernel, full 128 x 128 blocks only): as soon as columns 0..63 of L and the inverses of the diagonal blocks
// 0..3 are in global memory, *half_word = 1 is published (agent scope) -- the triangular solve of the tile below starts on
// its first four column blocks while this block's second half is still being factored.
// stream (tile kernel, round 3): every finished 16-column micro panel (and the inverse of its diagonal block) goes to memory with
// write-through stores and *half_word = number of micro panels in memory is published one step behind, without fences: the tiles
// below consume L(j,j) panel by panel (tile_process).  The caller publishes the final count (8) after this function returns.
template <bool FROM_LDS>
__device__ __forceinline__ int potf2_la_body(double* __restrict__ A, int64_t lda, int nb, double* __restrict__ linv_out,
                                             double* __restrict__ smem, long long* ts, unsigned* half_word = nullptr,
                                             bool stream = false) {
    double* As = smem;                           // NB x PLD, column-major
    double* dinv = smem + NB * PLD;              // 128 reciprocal pivots
    double* dummy = dinv + NB;                   // 80 doubles: sink of the non-leader lanes' reciprocal-pivot stores
    int* flag = reinterpret_cast<int*>(dummy + 80);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // keeps this body's per-thread addresses out of the caller's loop preheader
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lane = tid & 63;
    const int nt = (nb + 15) >> 4;
    const bool halfp = half_word != nullptr && nb == NB && linv_out != nullptr;
    P2_TS(0);
    if (tid == 0) *flag = 0;
    double a[16], b[16];
    if (!FROM_LDS) {
        if (wave == 0) {                         // micro panel 0 straight from global memory into the panel wave
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                a[c] = (lane < nb && c < nb) ? A[lane + (int64_t)c * lda] : 0.0;
                b[c] = (lane + 64 < nb && c < nb) ? A[lane + 64 + (int64_t)c * lda] : 0.0;
            }
        } else {                                 // columns 16.. -> LDS (tiles on or below the diagonal), 32 elements per thread
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
    } else {
        __syncthreads();                         // flag reset visible; the caller's block is complete in LDS
    }
    for (int jb = 0; jb < nb; jb += 16) {
        const int pw = min(16, nb - jb);
        asm volatile("" : "+v"(lane));           // opaque per iteration: the per-lane predicates below stay out of (spilled) SGPRs
        P2_TS(1 + (jb >> 4) * 5);
        if (wave == 0) {
            if (FROM_LDS || jb > 0) {
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
            const int rowa = (jb + lane < NB) ? lane : NB - jb + (lane & 15);
            const int rowb = (jb + 64 + lane < NB) ? 64 + lane : NB - jb + (lane & 15);
            double* col = As + jb * PLD + jb;
            P2_TS(2 + (jb >> 4) * 5);
            const int bad = two ? potf2_panel16<true>(a, b, col, rowa, rowb, dinv + jb, dummy, lane)
                                : potf2_panel16<false>(a, b, col, rowa, rowb, dinv + jb, dummy, lane);
            P2_TS(3 + (jb >> 4) * 5);
            if (bad && lane == 0) *flag = jb + bad;
        } else if (jb >= 16) {                   // look-ahead: the rest of the previous micro panel's update
            const int pjb = jb - 16, t1 = pjb / 16 + 2;          // column blocks t1 .. nt-1
            {   // micro panel pjb is final: LDS -> global memory (lower triangle), off the panel wave's chain
                const int tt = tid - 64;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int e = tt + (P2T - 64) * i;
                    const int r = e & (NB - 1), c = pjb + (e >> 7);
                    if (e < 16 * NB && r >= c && r < nb && c < nb) {
                        if (stream) st_wt(A + r + (int64_t)c * lda, As[c * PLD + r]);
                        else A[r + (int64_t)c * lda] = As[c * PLD + r];
                    }
                }
            }
            if (halfp && wave == 7 && (stream || pjb < 64)) {
                // inverse of the 16 x 16 diagonal block of micro panel pjb (final since the last barrier), column-oriented
                // substitution as at the end of this function; the look-ahead waves have slack behind the panel wave
                const int j = lane & 15;
                double x[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = (i == j) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    x[k] *= dinv[pjb + k];
#pragma unroll
                    for (int i = k + 1; i < 16; ++i) x[i] = fma(-As[(pjb + k) * PLD + pjb + i], x[k], x[i]);
                }
                if (lane < 16) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const double v = (i >= j) ? x[i] : 0.0;
                        if (stream) st_wt(linv_out + (pjb >> 4) * 256 + j * 16 + i, v);
                        else linv_out[(pjb >> 4) * 256 + j * 16 + i] = v;
                    }
                }
            }
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
        P2_TS(4 + (jb >> 4) * 5);
        if (halfp && (stream ? jb >= 16 : jb == 64))            // (the stores were issued a micro step ago: nothing to wait for)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // half: columns 0..63 + inverses 0..3; stream: micro panel jb - 16
        __syncthreads();                         // micro panel jb is in LDS; update jb-16 is complete
        P2_TS(5 + (jb >> 4) * 5);
        if (*flag) break;
        if (halfp && stream) {
            if (jb >= 16 && tid == P2T - 64)     // micro panels 0 .. jb/16 - 1 and their inverses are in memory (sc1 stores, drained)
                __hip_atomic_store(half_word, (unsigned)(jb >> 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (halfp && jb == 64 && tid == P2T - 64) {   // wave 7 has no tile of the next column block: off the chain
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(half_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (jb + 16 >= nb) break;
        {   // column block jb/16 + 1 (the next micro panel): one tile per wave
            const int t0 = jb / 16 + 1;
            if (t0 + wave < nt) potf2_tile_update(As, jb, t0, t0 + wave, lane);
        }
        __syncthreads();
    }
    P2_TS(41);
    const int failed = *flag;
    if (failed) return failed;
    {   // the last micro panel -> global memory
        const int ljb = (nt - 1) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + P2T * i;
            const int r = e & (NB - 1), c = ljb + (e >> 7);
            if (r >= c && r < nb && c < nb) {
                if (stream) st_wt(A + r + (int64_t)c * lda, As[c * PLD + r]);
                else A[r + (int64_t)c * lda] = As[c * PLD + r];
            }
        }
    }
    // inverses of the 16x16 diagonal blocks (trsm_panel_kernel): wave w <-> block w, lane j <-> column j of inv(L_d)
    // (blocks 0..3 were done on the way when half_word is set; all but the last one in stream mode)
    if (linv_out && wave < nt && !(halfp && (stream ? wave < nt - 1 : wave < 4))) {
        const int jb = wave * 16, pw = min(16, nb - jb);
        const int j = lane & 15;
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = (i == j) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {            // column-oriented substitution: 16-step chain, independent FMAs behind it
            x[k] *= dinv[jb + k];
#pragma unroll
            for (int i = k + 1; i < 16; ++i) x[i] = fma(-As[(jb + k) * PLD + jb + i], x[k], x[i]);
        }
        if (lane < 16) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double v = (i < pw && j < pw && i >= j) ? x[i] : 0.0;
                if (stream) st_wt(linv_out + wave * 256 + j * 16 + i, v);
                else linv_out[wave * 256 + j * 16 + i] = v;
            }
        }
    } else if (linv_out && wave >= nt && wave < 8) {
        // a ragged block (nb < 128): the inverse blocks beyond its last 16 x 16 block are ZERO, written here -- the consumers stage
        // all eight blocks of a tile (tile_invert_diag, the persistent solves' M) and multiply what lies beyond nb by zeros; 0 x
        // whatever the allocation held is only 0 if that was finite.  (Round 5, found by the 0xff-poisoned test allocator:
        // tests/test_gpu_stress.py n = 1500 returned NaNs; with the product's zero-filled blocks it had been right by accident.)
#pragma unroll
        for (int q = 0; q < 4; ++q) linv_out[wave * 256 + lane + 64 * q] = 0.0;
    }
    P2_TS(42);
    return 0;
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
    if (*info != 0) return;
    long long* ts = (blockIdx.z == 0) ? POTF2_TS_PTR : nullptr;
    const int failed = potf2_la_body<false>(A, lda, nb, linv_out, smem, ts);
    if (failed && threadIdx.x == 0) *info = col0 + failed;
}
#ifdef MI355KKT_DEBUG
int set_potf2_ts(long long* dptr) { return hipMemcpyToSymbol(HIP_SYMBOL(g_potf2_ts), &dptr, sizeof(dptr)) == hipSuccess ? 0 : -2; }
#endif

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

// ===================================================================================================
// potrf_tiles_kernel (round 2): the whole dense Cholesky as ONE persistent launch, left-looking by 128 x 128 tiles.
//
// Why: as a chain of ~330 launches (potf2 / trsm / skinny update per 128 columns, bulk updates on a side stream) the
// factorisation is bound by the chain kernels, and every chain kernel that overlaps the bulk update runs 3-5 x slower
// (profiles/r02_*: trsm 18 -> 88 us, update 33 -> 99 us) because it has to queue for compute units behind bulk
// workgroups.  Here every compute unit runs one 512-thread workgroup for the whole factorisation:
//   * tiles (i, j), i >= j, are handed out in column-major order by a ticket counter (dynamic, so the scheme cannot
//     deadlock whatever subset of the grid is resident: the smallest unfinished tile is always owned by a running
//     workgroup and depends only on smaller tiles);
//   * the owner keeps the tile in its MFMA accumulators and subtracts L(i, k) L(j, k)' for k = 0 .. j-1 as those
//     tiles become final (left-looking: C is read and written ONCE, K grows to 128 j), operands staged through LDS;
//   * diagonal tile: accumulators -> LDS -> the look-ahead potf2 above; off-diagonal: wait for L(j, j), accumulators
//     -> LDS -> X L(j,j)' = B on the matrix cores (same 16 x 16 inverse + refinement scheme as trsm_panel_kernel);
//   * prog[r] = number of final tiles in block row r (they become final in column order) is the only synchronisation:
//     producer: all stores drained -> __syncthreads -> one lane: agent-scope release fence, s_waitcnt, relaxed agent store;
//     consumer: one lane polls relaxed (bounded, s_sleep) -> agent-scope acquire fence -> __syncthreads -> plain loads
//     (guide G16; one acquire covers every k that has become available since the last look).
// ===================================================================================================
constexpr int PT_THREADS = 512;
// developer aid (-DMI355KKT_DEBUG builds only): 8 stamps per tile (s_memtime) when set
#ifdef MI355KKT_DEBUG
__device__ long long* g_tile_ts = nullptr;
#define TILE_TS_PTR g_tile_ts
#else
#define TILE_TS_PTR ((long long*)nullptr)
#endif
#define PT_TS(k_) do { if (tts && tid == 0) tts[(int64_t)t * 8 + (k_)] = (long long)__builtin_readcyclecounter(); } while (0)
struct TileCtl {
    unsigned ticket, abort_flag, pad0, pad1;
    unsigned prog[252];          // up to 252 block rows (n <= 32256); zeroed before every launch
    unsigned half[252];          // dense factorisation: 1 once columns 0..63 of L(j,j) and their inverses are published;
                                 // stream mode: the number of 16-column micro panels of L(j,j) (and inverses) in memory, 0..8
    unsigned micro[252];         // stream mode: the number of 16-column blocks of tile (i, i-1) in memory, 0..8
};
#define RLX_AGENT_ __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// acc -= Xi[:, 0:K] Xj[:, 0:K]'  (Xi: rows of the tile, Xj: its columns; both 128 x K panels of L, column-major, lda)
// 8 waves: wave w owns rows 16 w .. 16 w + 15 and ALL 128 columns: acc[t] = 16-column block t in the transposed MFMA
// layout (lane (li, lq), register r <-> row 16 w + li, column 16 t + lq + 4 r) -- exactly the layout the triangular
// solve of the finished tile works in, so an off-diagonal tile never leaves its registers.
// rt = the 16-row block of the tile this wave owns: its own index, except on DIAGONAL tiles, where row block r needs only
// the r + 1 column blocks left of / on the diagonal: waves w and w + 4 share a SIMD, so they take row blocks w and 7 - w
// (9 MFMA tiles per SIMD instead of up to 12).
constexpr int TILE_PF = 2;
// TAIL: K need not be a multiple of TILE_PF * BK (the ragged last factored tile of a frontal matrix); the dense factorisation always
// runs the TAIL = false instance, whose inner loop carries no extra checks.
// L2: the operands are read with sc1 loads (written a moment ago with write-through stores by another compute unit and announced
// without fences: the streamed last 128 columns of a diagonal tile).
// TBK: k-depth of one LDS stage (16, or 32: half the barriers per MFMA -- the dense factorisation's bulk; K a multiple of TBK)
template <bool diag, bool TAIL, bool L2 = false, int TBKP = BK>
__device__ __forceinline__ void tile_accumulate(d4 (&acc)[8], const double* __restrict__ Xi, const double* __restrict__ Xj,
                                                int64_t lda, int K, int mi, int mj, double* __restrict__ smem, int tid, int rt) {
    // (four 16-deep steps in flight — 64 more registers — spill and measured slower: potrf(8192) 4.51 -> 5.19 ms)
    constexpr int TBK = TBKP;
    constexpr int PF = TBKP == 32 ? 1 : TILE_PF;              // k-steps of operands in flight (global -> registers); same depth in k
    const int lane = tid & 63;
    const int li = lane & 15, lq = lane >> 4;
    const int ip = (tid & 63) * 2, kq = tid >> 6;            // staging: index pair, k = kq + 8 r
    constexpr int STG = TBK * LDT_M;                           // doubles per operand per stage
    constexpr int NR = TBK / 8;                                // k rows per thread and stage (8 waves)
    auto sJ = [&](int s_) -> double* { return smem + s_ * 2 * STG; };
    auto sI = [&](int s_) -> double* { return smem + s_ * 2 * STG + STG; };
    const int nkt = (K + TBK - 1) / TBK;                        // dense potrf: K is a multiple of 128; fronts: any K > 0
    double rJ[PF][2 * NR], rI[PF][2 * NR];
    const bool fullJ = (mj == 128), fullI = (mi == 128);
    auto fetch = [&](int kt, double (&J)[2 * NR], double (&I)[2 * NR]) {
        const bool ktail = TAIL && (kt + 1) * TBK > K;         // last step of a ragged k range: rows k >= K read as zero
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int64_t k = (int64_t)kt * TBK + kq + 8 * r;
            const double* pj = Xj + k * lda + ip;
            if (ktail) {
                const bool kin = k < K;
                J[2 * r] = (kin && ip < mj) ? (L2 ? ld_l2(pj) : pj[0]) : 0.0;
                J[2 * r + 1] = (kin && ip + 1 < mj) ? (L2 ? ld_l2(pj + 1) : pj[1]) : 0.0;
                if (!diag) {
                    const double* pi = Xi + k * lda + ip;
                    I[2 * r] = (kin && ip < mi) ? (L2 ? ld_l2(pi) : pi[0]) : 0.0;
                    I[2 * r + 1] = (kin && ip + 1 < mi) ? (L2 ? ld_l2(pi + 1) : pi[1]) : 0.0;
                }
                continue;
            }
            if (L2) {
                J[2 * r] = (ip < mj) ? ld_l2(pj) : 0.0;
                J[2 * r + 1] = (ip + 1 < mj) ? ld_l2(pj + 1) : 0.0;
            } else if (fullJ) {
                const d2u_ v = *reinterpret_cast<const d2u_*>(pj);
                J[2 * r] = v.x;
                J[2 * r + 1] = v.y;
            } else {
                J[2 * r] = (ip < mj) ? pj[0] : 0.0;
                J[2 * r + 1] = (ip + 1 < mj) ? pj[1] : 0.0;
            }
            if (!diag) {
                const double* pi = Xi + k * lda + ip;
                if (L2) {
                    I[2 * r] = (ip < mi) ? ld_l2(pi) : 0.0;
                    I[2 * r + 1] = (ip + 1 < mi) ? ld_l2(pi + 1) : 0.0;
                } else if (fullI) {
                    const d2u_ v = *reinterpret_cast<const d2u_*>(pi);
                    I[2 * r] = v.x;
                    I[2 * r + 1] = v.y;
                } else {
                    I[2 * r] = (ip < mi) ? pi[0] : 0.0;
                    I[2 * r + 1] = (ip + 1 < mi) ? pi[1] : 0.0;
                }
            }
        }
    };
    auto stash = [&](int s_, const double (&J)[2 * NR], const double (&I)[2 * NR]) {   // J stored negated: the MFMAs accumulate C - A B'
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int k = kq + 8 * r;
            d2_ vj = {-J[2 * r], -J[2 * r + 1]};
            *reinterpret_cast<d2_*>(sJ(s_) + k * LDT_M + ip) = vj;
            d2_ vi = {diag ? J[2 * r] : I[2 * r], diag ? J[2 * r + 1] : I[2 * r + 1]};
            *reinterpret_cast<d2_*>(sI(s_) + k * LDT_M + ip) = vi;
        }
    };
    if (nkt <= 0) return;
    // Operands that were written a moment ago by another compute unit come from HBM / the other XCD's write-back (2-3 us):
    // with one step of look-ahead every 16-deep step of the chain-critical last column waits for them; PF steps are in flight.
#pragma unroll
    for (int q = 0; q < PF; ++q)
        if (!TAIL || q < nkt) fetch(q, rJ[q], rI[q]);         // (TAIL = false: K is a multiple of 128, nkt >= 8)
    stash(0, rJ[0], rI[0]);
    __syncthreads();
    for (int kt0 = 0; kt0 < nkt; kt0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int kt = kt0 + u;
            if (TAIL && kt >= nkt) break;                     // (uniform) odd number of steps
            if (kt + PF < nkt) fetch(kt + PF, rJ[u], rI[u]);  // slot u held step kt, which went to LDS one step ago
            const double* __restrict__ Js = sJ(kt & 1);
            const double* __restrict__ Is = sI(kt & 1) + rt * 16;
            if (!diag) {                                      // branch-free body for the common case
#pragma unroll
                for (int kk = 0; kk < TBK; kk += 4) {
                    const double bv = Is[li + (kk + lq) * LDT_M];
                    double av[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) av[t] = Js[(t * 16 + li) + (kk + lq) * LDT_M];
#pragma unroll
                    for (int t = 0; t < 8; ++t) acc[t] = MFMA_F64(av[t], bv, acc[t]);
                }
            } else {                                          // diagonal tile: column blocks right of the row block are not needed
#pragma unroll
                for (int kk = 0; kk < TBK; kk += 4) {
                    const double bv = Is[li + (kk + lq) * LDT_M];
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        if (t <= rt) {
                            const double av = Js[(t * 16 + li) + (kk + lq) * LDT_M];
                            acc[t] = MFMA_F64(av, bv, acc[t]);
                        }
                    }
                }
            }
            if (kt + 1 < nkt) stash((kt + 1) & 1, rJ[(u + 1) % PF], rI[(u + 1) % PF]);
            __syncthreads();
        }
    }
}

// one lane: wait until *p >= need (or abort); returns the value seen, 0xffffffff on abort / timeout
// acquire = false: the data behind the word was stored write-through and is read with sc1 loads (no L1 invalidate needed)
__device__ __forceinline__ unsigned tile_wait(const unsigned* p, const unsigned* p2, unsigned need, TileCtl* ctl, int* err,
                                              bool acquire = true) {
    for (unsigned spins = 0; spins < (1u << 24); ++spins) {
        unsigned v = __hip_atomic_load(p, RLX_AGENT_);
        if (p2) {
            const unsigned v2 = __hip_atomic_load(p2, RLX_AGENT_);
            v = v < v2 ? v : v2;
        }
        if (v >= need) {
            if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            return v;
        }
        if (__hip_atomic_load(&ctl->abort_flag, RLX_AGENT_)) return 0xffffffffu;
        __builtin_amdgcn_s_sleep(4);
    }
    atomicExch(err, -7);                                      // err aliases *info: a negative value = hand-off timeout
    __hip_atomic_store(&ctl->abort_flag, 1u, RLX_AGENT_);
    return 0xffffffffu;
}

// all threads: make this workgroup's global stores visible device-wide, then publish *p = value
__device__ __forceinline__ void tile_publish(unsigned* p, unsigned value, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every storing wave drains its own stores
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p, value, RLX_AGENT_);
    }
}

// M = inv(L(j,j)) for the persistent triangular solves (blas2.hip: one hop = three 128 x 128 matrix-vector products
// instead of two 64-step substitution chains).  L(j,j) is still in the LDS image; the inverses D_i of its 16 x 16 diagonal
// blocks are brought into the padding rows.  Block column c of M is independent of the others: wave c computes
//     M_cc = D_c,   M_ic = -D_i * sum_{k=c}^{i-1} L_ik M_kc   (i = c+1 .. 7)
// on the matrix cores with tiles held transposed-free: a finished block's D registers (rows lq + 4 r) are exactly the B
// operand of the next products (k = 4 s + lq with s = r).  Off the factorisation's critical path (runs after the publish).
__device__ __forceinline__ void tile_invert_diag(double* __restrict__ As, const double* __restrict__ linv, int mj,
                                                 double* __restrict__ Mout, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = tid + PT_THREADS * q;                     // e = cb * 256 + k * 16 + g  ->  As[(16 cb + k) PLD + 128 + g]
        As[(e >> 4) * PLD + NB + (e & 15)] = linv[e];
    }
    __syncthreads();
    const int c = wave;                                       // my block column
    d4 Mb[8];                                                 // Mb[i]: rows 16 i + lq + 4 r, columns 16 c + li
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i < c) {
            Mb[i] = d4{0.0, 0.0, 0.0, 0.0};
        } else if (i == c) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Mb[i][r] = As[(16 * c + li) * PLD + NB + lq + 4 * r];     // D_c[lq+4r][li]
        } else {
            d4 t = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k >= c && k < i) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const double lv = As[(16 * k + 4 * s4 + lq) * PLD + 16 * i + li];   // L[16i+li][16k+4s+lq]
                        t = MFMA_F64(lv, Mb[k][s4], t);
                    }
                }
            }
            d4 m = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const double dv = -As[(16 * i + 4 * s4 + lq) * PLD + NB + li];              // -D_i[li][4s+lq]
                m = MFMA_F64(dv, t[s4], m);
            }
            Mb[i] = m;
        }
    }
    // column-major 128 x 128, M[row][col] at Mout[col * 128 + row]; rows / columns beyond mj come out as zeros
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double v = (i >= c) ? Mb[i][r] : 0.0;
            Mout[(16 * c + li) * NB + 16 * i + lq + 4 * r] = v;                  // M, column-major
            Mout[NB * NB + (16 * i + lq + 4 * r) * NB + 16 * c + li] = v;         // M', column-major (upper triangular)
        }
    (void)mj;
}

// One tile of a left-looking tiled (partial) Cholesky, shared by the dense kernel and the batched-fronts kernel.
struct TileJob {
    double* A;               // the matrix (dense S, or one frontal matrix), column-major
    int64_t lda;
    int i0, j0, mi, mj;      // tile origin and extent (<= 128 each)
    bool diag;
    int nk;                  // number of factored tile columns left of the tile to accumulate over
    int q, w;                // k-tile boundaries: column 128 kt for kt <= q, w beyond (fronts: ragged last factored tile)
    bool factored;           // true: the tile belongs to the factored columns (potf2 / trsm + publish); false: Schur part, store
    unsigned* half_j;        // dense only: half word of the diagonal tile of column j (nullptr: not used)
    unsigned* micro_i;       // dense only (round 3, tile_process<.., STREAM>): 16-column streaming of L(j,j) to the tiles below (half_j
                             //   counts micro panels) and of tile (i, i-1) to the diagonal tile (i, i): micro word of block row i
    unsigned* prog_i;        // progress word of the tile's block row (published to when factored)
    unsigned* prog_j;        // progress word of block row j (the tile's column index); == prog_i on diagonal tiles
    unsigned pub;            // value published / waited for once column j is final: j + 1
    double* linv;            // 2048 doubles: inverses of the 16 x 16 diagonal blocks of L(j,j)
    double* minv;            // dense only: 2 x 128 x 128 inverse of L(j,j) for the triangular solves, or nullptr
    int* info;               // failing pivot -> *info = info_base + column (1-based inside the tile)
    int info_base;
    bool abort_on_fail;      // dense: a non-positive pivot ends the factorisation; fronts: the column is published anyway so
                             // that the front's other tiles run to completion (on garbage) instead of waiting for it
};
__device__ __forceinline__ int tile_kcol(const TileJob& J, int kt) { return kt <= J.q ? kt * NB : J.w; }

// returns false when the workgroup has to leave the kernel (abort / timeout).  VB: batched fronts (ragged k ranges possible)
template <bool VB, bool STREAM = false, int TBK = BK>
__device__ __forceinline__ bool tile_process(const TileJob& J, TileCtl* ctl, int* err, double* __restrict__ smem,
                                             unsigned* ctlw, int tid, unsigned t, long long* tts) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    double* As = smem;
    double* const A = J.A;
    const int64_t lda = J.lda;
    const int i0 = J.i0, j0 = J.j0, mi = J.mi, mj = J.mj;
    const bool diag = J.diag;
    PT_TS(0);
    // ---- the tile itself -> accumulators (its loads fly while the first operands are fetched)
    const int rt = (diag && wave >= 4) ? 11 - wave : wave; // my 16-row block (see tile_accumulate)
    const int row = rt * 16 + li;                         // my row of the tile
    d4 acc[8];
#pragma unroll
    for (int tt = 0; tt < 8; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = tt * 16 + lq + 4 * r;
            acc[tt][r] = (row < mi && col < mj && (!diag || row >= col)) ? A[i0 + row + (int64_t)(j0 + col) * lda] : 0.0;
        }
    // ---- left-looking accumulation over the columns that are final, as they become final
    // sdiag (stream mode): the last 128 columns of a diagonal tile -- tile (i, i-1), the one the chain waits for -- are consumed
    // as 16-column blocks while the triangular solve that produces them is still running (micro word of block row i)
    // (dense STREAM: n is a multiple of 128, every tile is whole; fronts: the kernel sets micro_i only where tile (i, i-1) is whole)
    const bool sdiag = STREAM && diag && J.nk >= 1 && (!VB || (J.micro_i != nullptr && J.factored));
    const int nk_plain = sdiag ? J.nk - 1 : J.nk;
    int kdone = 0;
    while (kdone < nk_plain) {
        if (tid == 0) {
            const unsigned v = tile_wait(J.prog_i, diag ? nullptr : J.prog_j, (unsigned)kdone + 1, ctl, err);
            ctlw[1] = (v == 0xffffffffu) ? v : (v < (unsigned)nk_plain ? v : (unsigned)nk_plain);
        }
        __syncthreads();
        const unsigned ka = ctlw[1];
        if (ka == 0xffffffffu) return false;
        if ((int)ka == J.nk && kdone == J.nk - 1) PT_TS(5);   // exactly the last missing column has arrived
        const int c0 = tile_kcol(J, kdone), c1 = tile_kcol(J, (int)ka);
        if (VB && ((c1 - c0) & (TILE_PF * BK - 1))) {         // k range that is not a whole number of unrolled steps (fronts only)
            if (diag)
                tile_accumulate<true, true>(acc, A + i0 + (int64_t)c0 * lda, A + j0 + (int64_t)c0 * lda, lda, c1 - c0, mi, mj, smem, tid, rt);
            else
                tile_accumulate<false, true>(acc, A + i0 + (int64_t)c0 * lda, A + j0 + (int64_t)c0 * lda, lda, c1 - c0, mi, mj, smem, tid, rt);
        } else if (diag)
            tile_accumulate<true, false, false, TBK>(acc, A + i0 + (int64_t)c0 * lda, A + j0 + (int64_t)c0 * lda, lda, c1 - c0, mi, mj, smem, tid, rt);
        else
            tile_accumulate<false, false, false, TBK>(acc, A + i0 + (int64_t)c0 * lda, A + j0 + (int64_t)c0 * lda, lda, c1 - c0, mi, mj, smem, tid, rt);
        kdone = (int)ka;                                  // (tile_accumulate ends with a barrier: ctlw[1] is free again)
    }
    if (sdiag) {
        const int c0 = tile_kcol(J, J.nk - 1);
        int kb = 0;
        while (kb < 8) {
            if (tid == 0) ctlw[1] = tile_wait(J.micro_i, nullptr, (unsigned)kb + 1, ctl, err, false);
            __syncthreads();
            const unsigned v = ctlw[1];
            if (v == 0xffffffffu) return false;
            const int kb1 = v < 8u ? (int)v : 8;
            if (kb1 == 8) PT_TS(5);                           // the last block has arrived
            const double* X = A + i0 + (int64_t)(c0 + 16 * kb) * lda;
            tile_accumulate<true, true, true>(acc, X, X, lda, 16 * (kb1 - kb), mi, mj, smem, tid, rt);
            kb = kb1;                                         // (ends with a barrier: ctlw[1] is free again)
        }
    }
    PT_TS(1);
    if (!J.factored) {
        // ---- Schur-complement tile of a frontal matrix: the update is complete, back to memory (lower part on the diagonal)
#pragma unroll
        for (int tt = 0; tt < 8; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = tt * 16 + lq + 4 * r;
                if (row < mi && col < mj && (!diag || row >= col)) A[i0 + row + (int64_t)(j0 + col) * lda] = acc[tt][r];
            }
        return true;
    }
    if (diag) {
        // ---- accumulators -> LDS image of the tile (column-major, leading dimension PLD), then the look-ahead potf2
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 8; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) As[(tt * 16 + lq + 4 * r) * PLD + row] = acc[tt][r];
        const bool dstream = STREAM && (!VB || (J.half_j != nullptr && mj == NB));
        const int failed = potf2_la_body<true>(A + j0 + (int64_t)j0 * lda, lda, mj, J.linv, smem, nullptr, VB ? nullptr : J.half_j,
                                               dstream);
        if (failed) {
            if (tid == 0) atomicCAS(J.info, 0, J.info_base + failed);
            if (J.abort_on_fail) {
                if (tid == 0) __hip_atomic_store(&ctl->abort_flag, 1u, RLX_AGENT_);
                return false;
            }
            if (dstream && tid == 0) __hip_atomic_store(J.half_j, 8u, RLX_AGENT_);   // releases the streaming tiles below (on garbage)
            tile_publish(J.prog_i, J.pub, tid);
            return true;
        }
        PT_TS(3);
        if (dstream) {            // the last micro panel and inverse (write-through stores): drained -> all eight are announced
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(J.half_j, 8u, RLX_AGENT_);
        }
        tile_publish(J.prog_i, J.pub, tid);
        PT_TS(4);
        if (J.minv) {             // after the publish: nobody on the factorisation's chain waits for this
            __syncthreads();
            tile_invert_diag(As, J.linv, mj, J.minv, tid);
        }
        return true;
    }
    // ---- X L(j,j)' = B on the accumulators (transposed tiles, diagonal blocks by inverse + one refinement step, as
    //      trsm_panel_kernel), operands from the LDS image of L(j,j); column block cb needs row block cb of L(j,j) (its columns
    //      0 .. 16 cb + 15) and the inverse of the diagonal block cb
    auto solve_block = [&](auto cbc) {
        constexpr int cb = decltype(cbc)::value;
        d4 a4 = acc[cb], a5 = d4{0.0, 0.0, 0.0, 0.0};     // two chains: the matrix pipe is not left waiting on one accumulator
#pragma unroll
        for (int c = 0; c < cb; ++c) {
#pragma unroll
            for (int s4 = 0; s4 < 4; s4 += 2) {
                const double lv0 = As[(16 * c + 4 * s4 + lq) * PLD + 16 * cb + li];
                const double lv1 = As[(16 * c + 4 * (s4 + 1) + lq) * PLD + 16 * cb + li];
                a4 = MFMA_F64(-lv0, acc[c][s4], a4);
                a5 = MFMA_F64(-lv1, acc[c][s4 + 1], a5);
            }
        }
        if (cb > 0) a4 += a5;
        double mi4[4], ld4[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            mi4[s4] = As[(16 * cb + 4 * s4 + lq) * PLD + NB + li];
            const double lv = As[(16 * cb + 4 * s4 + lq) * PLD + 16 * cb + li];
            ld4[s4] = (4 * s4 + lq <= li) ? -lv : 0.0;
        }
        d4 x0 = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) x0 = MFMA_F64(mi4[s4], a4[s4], x0);
        d4 e4 = a4;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) e4 = MFMA_F64(ld4[s4], x0[s4], e4);
        d4 xx = x0;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) xx = MFMA_F64(mi4[s4], e4[s4], xx);
        acc[cb] = xx;
        __builtin_amdgcn_sched_barrier(0);            // operand reads run ahead inside one column block only (register budget)
    };
    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;
    using C4 = std::integral_constant<int, 4>; using C5 = std::integral_constant<int, 5>;
    using C6 = std::integral_constant<int, 6>; using C7 = std::integral_constant<int, 7>;
    const bool smode = STREAM && (!VB || (J.half_j != nullptr && mj == NB));
    if (smode) {
        // ---- stream mode (round 3): L(j,j) arrives as 16-column micro panels (write-through stores of the diagonal tile's owner,
        //      half word = panels in memory, no fences): every look takes what has been announced, row block by row block, with
        //      sc1 loads; the finished column blocks of this tile go back to memory the same way.  Tile (j+1, j) -- the one the next
        //      diagonal tile waits for -- announces its column blocks through the micro word of its block row, one block behind.
        const double* __restrict__ Lg = A + j0 + (int64_t)j0 * lda;
        const bool announce = J.micro_i != nullptr;
        int staged = 0;                                   // row blocks of L(j,j) in the LDS image
        auto step = [&](auto cbc) -> bool {
            constexpr int cb = decltype(cbc)::value;
            if (staged <= cb) {
                if (tid == 0) ctlw[1] = tile_wait(J.half_j, nullptr, (unsigned)cb + 1, ctl, err, false);
                __syncthreads();
                const unsigned v = ctlw[1];
                if (v == 0xffffffffu) return false;
                const int upto = v < 8u ? (int)v : 8;
                if (cb == 0) PT_TS(2);
                for (int b = staged; b < upto; ++b) {     // row block b: rows 16 b .. + 15, columns 0 .. 16 b + 15, and D_b
                    const int nel = 256 * (b + 1);
                    for (int e = tid; e < nel; e += PT_THREADS) {
                        const int r = 16 * b + (e & 15), c = e >> 4;
                        As[c * PLD + r] = (r >= (c & ~15)) ? ld_l2(Lg + r + (int64_t)c * lda) : 0.0;
                    }
                    if (tid < 256) As[(16 * b + (tid >> 4)) * PLD + NB + (tid & 15)] = ld_l2(J.linv + 256 * b + tid);
                }
                staged = upto;
                __syncthreads();
            }
            solve_block(cbc);
            // column block cb of the tile is final: to memory (write-through)
            if (announce && mi == NB) {
                // a whole tile on the chain: every lane stores, NO branch between this block's four stores and the wait -- they sit
                // in one basic block, which is what tools/check_stream_isa.py verifies on the compiled code
#pragma unroll
                for (int r = 0; r < 4; ++r) st_wt(A + i0 + row + (int64_t)(j0 + 16 * cb + lq + 4 * r) * lda, acc[cb][r]);
                if (cb > 0) {                             // blocks 0 .. cb-1 were stored a block ago: drained by now
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // (all but this block's four stores)
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(J.micro_i, (unsigned)cb, RLX_AGENT_);
                }
            } else {
                if (row < mi) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) st_wt(A + i0 + row + (int64_t)(j0 + 16 * cb + lq + 4 * r) * lda, acc[cb][r]);
                }
                if (announce && cb > 0) {                 // ragged last factored tile of a front: lanes past its rows store nothing,
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // so the count of outstanding stores is not known: drain all
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(J.micro_i, (unsigned)cb, RLX_AGENT_);
                }
            }
            return true;
        };
        if (!step(C0{}) || !step(C1{}) || !step(C2{}) || !step(C3{}) || !step(C4{}) || !step(C5{}) || !step(C6{}) || !step(C7{}))
            return false;
        PT_TS(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (announce) __hip_atomic_store(J.micro_i, 8u, RLX_AGENT_);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // (nothing dirty to write back: the tile went out write-through)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(J.prog_i, J.pub, RLX_AGENT_);
        }
        PT_TS(4);
        return true;
    }
    if constexpr (VB || !STREAM) {
    // halfmode (dense, full tiles): the diagonal tile publishes its columns 0..63 (+ the inverses of the diagonal blocks
    // 0..3) while its second half is still being factored; the first four column blocks of this solve need nothing else
    const bool halfmode = !VB && J.half_j != nullptr && mj == NB;
    if (tid == 0) ctlw[1] = halfmode ? tile_wait(J.half_j, nullptr, 1u, ctl, err) : tile_wait(J.prog_j, nullptr, J.pub, ctl, err);
    __syncthreads();
    if (ctlw[1] == 0xffffffffu) return false;
    PT_TS(2);
    // ---- L(j,j) (tiles on / below the diagonal) and the inverses of its 16 x 16 diagonal blocks -> LDS, once for
    //      the eight waves: one coalesced sweep instead of strided global loads in front of every MFMA group.  The
    //      inverses ride in the 16 padding rows of the image: element (block cb, column k, row g) at As[(16cb+k) PLD + 128 + g].
    //      Pass 0 = columns 0..63 (all rows) and the inverses 0..3, pass 1 = columns 64..127 and the inverses 4..7.
    auto stage = [&](int half) {
        const double* __restrict__ Lg = A + j0 + (int64_t)j0 * lda;
        const double* __restrict__ linv = J.linv;
        {                                             // 18 values in flight per thread (register budget)
            double v[18];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = tid + PT_THREADS * (16 * half + q);
                const int r = e & (NB - 1), c = e >> 7;
                v[q] = (r >= (c & ~15) && r < mj && c < mj) ? Lg[r + (int64_t)c * lda] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) v[16 + q] = linv[tid + PT_THREADS * (2 * half + q)];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = tid + PT_THREADS * (16 * half + q);
                As[(e >> 7) * PLD + (e & (NB - 1))] = v[q];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = tid + PT_THREADS * (2 * half + q);     // e = cb * 256 + k * 16 + g
                As[(e >> 4) * PLD + NB + (e & 15)] = v[16 + q];
            }
        }
    };
    stage(0);
    if (!halfmode) stage(1);
    __syncthreads();
    {
        solve_block(C0{});
        solve_block(C1{});
        solve_block(C2{});
        solve_block(C3{});
        if (halfmode) {                               // the second half of L(j,j): wait for the tile's final publish
            if (tid == 0) ctlw[1] = tile_wait(J.prog_j, nullptr, J.pub, ctl, err);
            __syncthreads();
            if (ctlw[1] == 0xffffffffu) return false;
            stage(1);
            __syncthreads();
        }
        solve_block(C4{});
        solve_block(C5{});
        solve_block(C6{});
        solve_block(C7{});
    }
    // (stores after the branch-free solve: the compiler can then run the LDS operand reads ahead of the MFMAs)
    if (mi == NB && mj == NB) {
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) A[i0 + row + (int64_t)(j0 + 16 * cb + lq + 4 * r) * lda] = acc[cb][r];
    } else if (row < mi) {
#pragma unroll
        for (int cb = 0; cb < 8; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = 16 * cb + lq + 4 * r;
                if (col < mj) A[i0 + row + (int64_t)(j0 + col) * lda] = acc[cb][r];
            }
    }
    }
    PT_TS(3);
    tile_publish(J.prog_i, J.pub, tid);
    PT_TS(4);
    return true;
}

template <bool STREAM, int TBK>
__global__ __launch_bounds__(PT_THREADS) void potrf_tiles_kernel(double* __restrict__ A, int64_t lda, int n, TileCtl* ctl,
                                                                 double* __restrict__ linv_all, int* __restrict__ info,
                                                                 int* __restrict__ err, double* __restrict__ minv_all,
                                                                 int use_half) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // the control words of the workgroup live behind the potf2 image (all LDS in the dynamic region, guide G17)
    unsigned* ctlw = reinterpret_cast<unsigned*>(smem + NB * PLD + NB + 80 + 2);
    const int NT = (n + NB - 1) / NB;
    const unsigned ntiles = (unsigned)NT * (NT + 1) / 2;
    for (;;) {
        // the thread index is made opaque once per tile: otherwise every per-thread address (column * lda products of the
        // loads / stores below) is hoisted out of the tile loop and lives in spilled registers
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (tid == 0) ctlw[0] = __hip_atomic_fetch_add(&ctl->ticket, 1u, RLX_AGENT_);
        __syncthreads();
        const unsigned t = ctlw[0];
        __syncthreads();                                      // ctlw[0] may be rewritten below
        if (t >= ntiles) return;
        int j = 0, rem = (int)t;                              // column-major tile order
        while (rem >= NT - j) {
            rem -= NT - j;
            ++j;
        }
        const int i = j + rem;
        TileJob J;
        J.A = A; J.lda = lda;
        J.i0 = i * NB; J.j0 = j * NB;
        J.mi = min(NB, n - J.i0); J.mj = min(NB, n - J.j0);
        J.diag = (i == j);
        J.nk = j; J.q = 0x3fffffff; J.w = 0;
        J.factored = true;
        J.prog_i = &ctl->prog[i]; J.prog_j = &ctl->prog[j];
        J.half_j = use_half ? &ctl->half[j] : nullptr;
        // the micro word of block row i: written by tile (i, i-1), read by the diagonal tile (i, i)
        J.micro_i = (STREAM && (i == j || i == j + 1)) ? &ctl->micro[i] : nullptr;
        J.pub = (unsigned)j + 1;
        J.linv = linv_all + (int64_t)j * 2048;
        J.minv = minv_all ? minv_all + (int64_t)j * 2 * NB * NB : nullptr;
        J.info = info; J.info_base = J.j0;
        J.abort_on_fail = true;
        if (!tile_process<false, STREAM, TBK>(J, ctl, err, smem, ctlw, tid, t, TILE_TS_PTR)) return;
        __syncthreads();                                      // LDS (image, control words) is reused by the next tile
    }
}

// ---------------------------------------------------------------------------------------------------
// potrf_tiles_vb_kernel: the same machine for ALL big frontal matrices of one level of the supernodal tree in one launch
// (sparse engine; replaces, per level, a potf2 / trsm / update launch triple for every 128 columns of the widest front).
// A front is h x h with its first w columns to eliminate: tile boundaries at 0, 128, .., 128 q, w, w + 128, .. (the last
// factored tile is as narrow as it has to be, the Schur part starts on its own tile boundary), tiles (i, j) of all fronts
// are handed out by ticket in the order (j, front, i) prepared by the symbolic analysis, so every front's chain starts at
// once.  Tiles left of w are finalised and published exactly as in the dense kernel; tiles of the Schur complement just
// receive the left-looking update and go back to memory.  A non-positive pivot is recorded for its front
// (info[front] = permuted column + 1) and releases that front's waiters; the other fronts are unaffected.
// ---------------------------------------------------------------------------------------------------
struct VbTicket { int front, i, j, pad; };
__global__ __launch_bounds__(PT_THREADS) void potrf_tiles_vb_kernel(double* __restrict__ base, const VbDesc* __restrict__ desc,
                                                                    const VbTicket* __restrict__ tickets, unsigned ntickets,
                                                                    const int* __restrict__ prog_off, const int* __restrict__ linv_off,
                                                                    TileCtl* ctl, unsigned* __restrict__ prog,
                                                                    double* __restrict__ linv_all, int* __restrict__ info,
                                                                    int* __restrict__ err, int nprog, int stream) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    unsigned* ctlw = reinterpret_cast<unsigned*>(smem + NB * PLD + NB + 80 + 2);
    for (;;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (tid == 0) ctlw[0] = __hip_atomic_fetch_add(&ctl->ticket, 1u, RLX_AGENT_);
        __syncthreads();
        const unsigned t = ctlw[0];
        __syncthreads();
        if (t >= ntickets) return;
        const VbTicket tk = tickets[t];
        const VbDesc dd = desc[tk.front];
        const int q = dd.w / NB, wr = dd.w - q * NB, ntf = q + (wr > 0 ? 1 : 0);
        auto origin = [&](int tt) { return tt < q ? tt * NB : (tt < ntf ? q * NB : dd.w + (tt - ntf) * NB); };
        auto extent = [&](int tt) { return tt < q ? NB : (tt < ntf ? wr : min(NB, dd.h - (dd.w + (tt - ntf) * NB))); };
        unsigned* fprog = prog + prog_off[tk.front];
        TileJob J;
        J.A = base + dd.off; J.lda = dd.h;
        J.i0 = origin(tk.i); J.j0 = origin(tk.j);
        J.mi = extent(tk.i); J.mj = extent(tk.j);
        J.diag = (tk.i == tk.j);
        J.factored = tk.j < ntf;
        J.nk = J.factored ? tk.j : ntf;
        J.q = q; J.w = dd.w;
        J.prog_i = fprog + tk.i; J.prog_j = fprog + tk.j;
        // round 3: 16-column streaming inside a front wherever the diagonal tile of the column is a whole 128 x 128 factored tile
        // (half / micro words of the level: the second and third third of the progress array)
        const bool scol = stream && J.factored && tk.j < q;                        // column tk.j streams L(j,j)
        J.half_j = scol ? fprog + nprog + tk.j : nullptr;
        J.micro_i = nullptr;
        if (stream && tk.i >= 1 && tk.i - 1 < q && tk.i < ntf) {                    // tile (i, i-1) is whole and (i, i) is factored
            if (tk.i == tk.j || tk.j == tk.i - 1) J.micro_i = fprog + 2 * nprog + tk.i;
        }
        J.pub = (unsigned)tk.j + 1;
        J.linv = linv_all + (int64_t)(linv_off[tk.front] + (J.factored ? tk.j : 0)) * 2048;
        J.minv = nullptr;
        J.info = info + tk.front; J.info_base = dd.col0 + J.j0;
        J.abort_on_fail = false;
        if (!tile_process<true, true>(J, ctl, err, smem, ctlw, tid, t, nullptr)) return;
        __syncthreads();
    }
}

#ifdef MI355KKT_DEBUG
int set_tile_ts(long long* dptr) { return hipMemcpyToSymbol(HIP_SYMBOL(g_tile_ts), &dptr, sizeof(dptr)) == hipSuccess ? 0 : -2; }
#endif

// One diagonal-block factorisation launch (nz blocks along blockIdx.z).
static int launch_potf2(double* A, int64_t lda, int nb, int col0, int* info, double* linv, int64_t bstride,
                        const VbDesc* vb, int nz, hipStream_t st) {
    static bool attr_set = false;
    constexpr size_t lds_la = sizeof(double) * (NB * PLD + NB + 80) + 16;
    if (!attr_set) {
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(potf2_la_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_la));
        attr_set = true;
    }
    hipLaunchKernelGGL(potf2_la_kernel, dim3(1, 1, nz), dim3(P2T), lds_la, st, A, lda, nb, col0, info, linv, bstride, vb);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

int potrf_work_init_batched(PotrfWork& w, int nbatch) {
    KKT_HIP_CHECK(DEV_ALLOC(&w.d_info, sizeof(int) * nbatch));
    KKT_HIP_CHECK(DEV_ALLOC(&w.d_dinv, sizeof(double) * 8 * 256 * nbatch));   // inverses of the 16x16 diagonal blocks
    KKT_HIP_CHECK(hipHostMalloc(&w.h_info, sizeof(int) * nbatch));
    memset(w.h_info, 0, sizeof(int) * nbatch);
    {   // the bulk updates yield to the critical-path kernels of the main stream at workgroup granularity
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&w.side, hipStreamNonBlocking, least) != hipSuccess)
            KKT_HIP_CHECK(hipStreamCreateWithFlags(&w.side, hipStreamNonBlocking));
    }
    return 0;
}

int potrf_work_init(PotrfWork& w) { return potrf_work_init_batched(w, 1); }

void potrf_work_free(PotrfWork& w) {
    if (w.d_info) (void)dev_free(w.d_info);
    if (w.d_dinv) (void)dev_free(w.d_dinv);
    if (w.h_info) (void)hipHostFree(w.h_info);
    if (w.d_ctl) (void)dev_free(w.d_ctl);
    if (w.d_linv_all) (void)dev_free(w.d_linv_all);
    if (w.d_minv) (void)dev_free(w.d_minv);
    if (w.d_m512) (void)dev_free(w.d_m512);
    if (w.d_m512_scratch) (void)dev_free(w.d_m512_scratch);
    if (w.d_gran512) (void)dev_free(w.d_gran512);
    w.d_m512 = w.d_m512_scratch = nullptr;
    w.d_gran512 = nullptr;
    w.m512_blocks = w.m512_n = 0;
    for (auto e : w.ev_panel) (void)hipEventDestroy(e);
    for (auto e : w.ev_bulk) (void)hipEventDestroy(e);
    if (w.side) (void)hipStreamDestroy(w.side);
    w = PotrfWork();
}

// all big fronts of one level of the supernodal tree (see potrf_tiles_vb_kernel); d_info: nfronts ints, zeroed here
int launch_potrf_tiles_vb(double* base, const VbDesc* d_desc, int nfronts, const void* d_tickets, int ntickets,
                          const int* d_prog_off, const int* d_linv_off, void* d_ctl, unsigned* d_prog, int nprog, double* d_linv,
                          int* d_info, hipStream_t st, bool prezeroed) {
    if (nfronts <= 0 || ntickets <= 0) return 0;
    static int num_cus = 0;
    static bool attr_set = false;
    constexpr size_t lds = sizeof(double) * (NB * PLD + NB + 80 + 2 + 2) + 16;
    if (!attr_set) {
        int dev = 0;
        KKT_HIP_CHECK(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        KKT_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_tiles_vb_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    if (!prezeroed) {       // (the sparse engine zeroes the state of all its levels with one memset per factorisation)
        KKT_HIP_CHECK(hipMemsetAsync(d_info, 0, sizeof(int) * nfronts, st));
        KKT_HIP_CHECK(hipMemsetAsync(d_ctl, 0, sizeof(TileCtl), st));
        KKT_HIP_CHECK(hipMemsetAsync(d_prog, 0, sizeof(unsigned) * 3 * (nprog > 0 ? nprog : 1), st));
    }
    // d_prog holds 3 * nprog words: progress, half (micro panels of L(j,j) in memory), micro (blocks of tile (i, i-1) in memory)
    constexpr int stream = 1;        // 16-column streaming inside whole tiles of the fronts (DESIGN 4a')
    const int grid = ntickets < num_cus ? ntickets : num_cus;
    hipLaunchKernelGGL(potrf_tiles_vb_kernel, dim3(grid), dim3(PT_THREADS), lds, st, base, d_desc,
                       reinterpret_cast<const VbTicket*>(d_tickets), (unsigned)ntickets, d_prog_off, d_linv_off,
                       reinterpret_cast<TileCtl*>(d_ctl), d_prog, d_linv, d_info, d_info, nprog > 0 ? nprog : 1, stream);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}
size_t potrf_tile_ctl_bytes() { return sizeof(TileCtl); }

// device state of the persistent tile kernel for matrices up to n x n (idempotent; launch_potrf calls it lazily)
int potrf_work_reserve(PotrfWork& w, int n) {
    const int NT = (n + NB - 1) / NB;
    if (!w.d_ctl) KKT_HIP_CHECK(DEV_ALLOC(&w.d_ctl, sizeof(TileCtl)));
    if (w.linv_tiles < NT) {
        if (w.d_linv_all) (void)dev_free(w.d_linv_all);
        w.d_linv_all = nullptr;
        w.linv_tiles = 0;
        w.minv_n = 0;
        KKT_HIP_CHECK(DEV_ALLOC(&w.d_linv_all, sizeof(double) * 2048 * (size_t)NT));
        if (w.d_minv) (void)dev_free(w.d_minv);
        w.d_minv = nullptr;
        KKT_HIP_CHECK(DEV_ALLOC(&w.d_minv, sizeof(double) * 2 * NB * NB * (size_t)NT));
        w.linv_tiles = NT;
    }
    return 0;
}

// the persistent tile kernel: one 512-thread workgroup per compute unit, all of the factorisation in one launch
static int launch_potrf_tiles(double* A, int64_t lda, int n, PotrfWork& w, hipStream_t st) {
    static int num_cus = 0;
    static bool attr_set = false;
    constexpr size_t lds = sizeof(double) * (NB * PLD + NB + 80 + 2 + 2) + 16;
    if (!attr_set) {
        int dev = 0;
        KKT_HIP_CHECK(hipGetDevice(&dev));
        hipDeviceProp_t prop;
        KKT_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_tiles_kernel<false, 16>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_tiles_kernel<true, 16>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int NT = (n + NB - 1) / NB;
    if (int e = potrf_work_reserve(w, n)) return e;
    KKT_HIP_CHECK(hipMemsetAsync(w.d_ctl, 0, sizeof(TileCtl), st));
    const int ntiles = NT * (NT + 1) / 2;
    const int grid = ntiles < num_cus ? ntiles : num_cus;
    // whole tiles only: 16-column streaming along the chain (round 3); ragged last tiles: the half-tile hand-off of L(j,j) (round 2)
    const int use_half = (n % NB) ? 1 : 2;
    if (use_half == 2)
        hipLaunchKernelGGL((potrf_tiles_kernel<true, 16>), dim3(grid), dim3(PT_THREADS), lds, st, A, lda, n,
                           reinterpret_cast<TileCtl*>(w.d_ctl), w.d_linv_all, w.d_info, w.d_info, w.d_minv, use_half);
    else
        hipLaunchKernelGGL((potrf_tiles_kernel<false, 16>), dim3(grid), dim3(PT_THREADS), lds, st, A, lda, n,
                           reinterpret_cast<TileCtl*>(w.d_ctl), w.d_linv_all, w.d_info, w.d_info, w.d_minv, use_half);
    KKT_HIP_CHECK(hipGetLastError());
    w.minv_n = n;                        // the 128 x 128 inverses of this factor's diagonal blocks are valid
    w.minv_of = A;
    return 0;
}

int launch_potrf_batched(double* A, int64_t lda, int n, int nbatch, int64_t bstride, PotrfWork& w, hipStream_t st) {
    // (round 4 measured the batch as nbatch "fronts" of the variable-batched tile kernel, one launch instead of this chain: 512
    //  problems of n = 512: factor 4.35 -> 4.8 ms, SLOWER -- profiles/r04_batch_tiles_ab.txt; not kept)
    if (w.m512_of == A) w.m512_n = 0;        // the 512 x 512 inverses of an earlier factor of this matrix are stale
    KKT_HIP_CHECK(hipMemsetAsync(w.d_info, 0, sizeof(int) * nbatch, st));
    // single large matrix: the persistent left-looking tile kernel (up to 252 block columns: TileCtl)
    constexpr int tiles_min_n = 1024;
    if (nbatch == 1 && n >= tiles_min_n && (n + NB - 1) / NB <= 252) return launch_potrf_tiles(A, lda, n, w, st);
    // the launch chain below does not produce the 128 x 128 diagonal-block inverses: those of an earlier tile factorisation of THIS
    // matrix are stale now; those of another matrix (S, while the small Schur complement K goes through here) stay valid
    if (w.minv_of == A) w.minv_n = 0;
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
    constexpr int outer1_max_n = 2048;
    if (nbatch == 1 && n >= 8 * NB && w.side) {     // (only orders beyond the tile kernel's 252 block columns get here)
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

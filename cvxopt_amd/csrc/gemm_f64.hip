// FP64 MFMA (v_mfma_f64_16x16x4_f64) level-3 kernels for gfx950 / MI355X.
//
//   syrk_tn_kernel     S(lower) = P(lower) + Gs' Gs,  Gs = diag(di) G applied while staging
//                      (replaces reference misc.py:1418 `Gs = diag(di) G` + :1422/:1451 base.syrk +
//                       :1426/:1455 `S += H`; and misc.py:1271-1276 scale+syrk+`K += H` for 'l' cones)
//   nt_update_kernel   C -= A B'   (Cholesky trailing / panel updates; the dsyrk/dgemm inside dpotrf,
//                       reference src/C/lapack.c:1508)
//
// One 256-thread workgroup (4 waves, 2x2) owns a 128x128 tile of C; each wave owns 64x64 = 4x4 MFMA
// tiles (16 accumulators of 4 doubles = 128 VGPRs).  Operand tiles are staged global -> registers ->
// LDS (the scaling by di rides on the register hop), double buffered, one barrier per 16-deep k-step.
// MFMA operand roles are chosen so that the D layout (col = lane&15) runs along C's contiguous (row)
// dimension: A-operand <- C-column block (J), B-operand <- C-row block (I).
#include "kkt_common.h"
#include <type_traits>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>

namespace mi355kkt {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(8))) d2u { double x, y; };   // 8-byte aligned pair (dwordx4 load)

#define MFMA_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// Developer ablation switch of the SYRK (bit0 no global fetch, bit1 no LDS stash, bit2 no barrier, bit4 no static priority, bit5 long
// diagonal tiles through the general path, bit6 no block masks on diagonal tiles there): results are WRONG when it is set, so it only
// exists in -DMI355KKT_DEBUG builds (include/mi355kkt_debug.h); production code compiles the constant 0.
#ifdef MI355KKT_DEBUG
__device__ int g_syrk_skip = 0;
int set_syrk_skip(int v) { return hipMemcpyToSymbol(HIP_SYMBOL(g_syrk_skip), &v, sizeof(int)) == hipSuccess ? 0 : -2; }
#define SYRK_SKIP g_syrk_skip
#else
#define SYRK_SKIP 0
#endif

// ---------------------------------------------------------------------------------------------------
// 64x64 wave tile: acc[t][u] += Js[t-th 16 rows][k] * Is[u-th 16 rows][k] over one BK slab.
//   SI / SK: LDS element strides along the tile index / along k.
// ---------------------------------------------------------------------------------------------------
template <int SI, int SK>
__device__ __forceinline__ void wave_mma(const double* __restrict__ Js, const double* __restrict__ Is,
                                         d4 (&acc)[4][4], int lane) {
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
        double a[4], b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = Js[(t * 16 + li) * SI + (kk + lk) * SK];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = Is[(u * 16 + li) * SI + (kk + lk) * SK];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[t][u] = MFMA_F64(a[t], b[u], acc[t][u]);
    }
}

__device__ __forceinline__ void zero_acc(d4 (&acc)[4][4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = d4{0.0, 0.0, 0.0, 0.0};
}

// ===================================================================================================
// Scaled SYRK, "TN": S[i,j] = P[i,j] + sum_k di[k]^2 G[k,i] G[k,j],  G column-major (k contiguous)
// ===================================================================================================
// staging map: thread -> columns (tid>>3) + 32 r (r = 0..3), k pair (tid&7)*2
template <bool FAST>
__device__ __forceinline__ void tn_load(const double* __restrict__ G, int64_t ldg, int col0, int n,
                                        int k, int kend, int sc, double (&reg)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = col0 + sc + 32 * r;
        const double* p = G + (int64_t)c * ldg + k;
        if (FAST) {
            const d2u v = *reinterpret_cast<const d2u*>(p);
            reg[2 * r] = v.x;
            reg[2 * r + 1] = v.y;
        } else {
            const bool cok = c < n;
            reg[2 * r] = (cok && k < kend) ? p[0] : 0.0;
            reg[2 * r + 1] = (cok && k + 1 < kend) ? p[1] : 0.0;
        }
    }
}

__device__ __forceinline__ void tn_store(double* __restrict__ Xs, int sc, int sk, const double (&reg)[8],
                                         double w0, double w1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        d2 v = {reg[2 * r] * w0, reg[2 * r + 1] * w1};
        *reinterpret_cast<d2*>(Xs + (sc + 32 * r) * LDT_K + sk) = v;
    }
}

// DMASK: diagonal tiles of short contractions run the pipelined loop with block masks (see there); <false> is the kernel of the
// long contractions (their diagonal tiles take the nine-blocks-per-wave path) with no trace of the masks in its code.
// One call = one contraction segment of one tile; the kernel below calls it once per segment of its workgroup.
template <bool DMASK>
__device__ __forceinline__ void syrk_tile(
    const SyrkItem& it, const double* __restrict__ G, int ldg32, const double* __restrict__ di, int n, int fast_ok,
    double* __restrict__ C, int ldc32, const double* __restrict__ P, int ldp32, double* __restrict__ slabs,
    double* __restrict__ smem, int wave_s) {
    // (leading dimensions travel as 32-bit scalars across the kernel's segment loop: three scalar registers less to keep live)
    const int64_t ldg = ldg32, ldc = ldc32, ldp = ldp32;
    // the thread index is rebuilt per segment from the wave's index (a scalar) and the lane id, and made opaque: nothing derived
    // from it is hoisted out of the kernel's segment loop and no vector register stays live across segments (the kernel sits at
    // 256 registers)
    int tid = wave_s * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    int wj = wave >> 1, wi = wave & 1;
    const int i0 = it.ti * TILE, j0 = it.tj * TILE;
    const bool diag = (it.ti == it.tj);
    const int sc = tid >> 3, sk = (tid & 7) * 2;
    const bool tile_fast = fast_ok && (i0 + TILE <= n) && (j0 + TILE <= n);

    // stage s: J operand at smem + s*2*STAGE, I operand right after (aliased for diagonal tiles)
    const int ioff = diag ? 0 : STAGE_DOUBLES;
    auto sJ = [&](int s) -> double* { return smem + s * 2 * STAGE_DOUBLES; };
    auto sI = [&](int s) -> double* { return smem + s * 2 * STAGE_DOUBLES + ioff; };

    d4 acc[4][4];
    zero_acc(acc);

    const int nkt = (it.k1 - it.k0 + BK - 1) / BK;
    double rJ[8], rI[8], w0 = 1.0, w1 = 1.0;

    auto fetch = [&](int kt) {
        const int k = it.k0 + kt * BK + sk;
        const bool fast = tile_fast && (it.k0 + (kt + 1) * BK <= it.k1);
        if (fast) {
            tn_load<true>(G, ldg, j0, n, k, it.k1, sc, rJ);
            if (!diag) tn_load<true>(G, ldg, i0, n, k, it.k1, sc, rI);
        } else {
            tn_load<false>(G, ldg, j0, n, k, it.k1, sc, rJ);
            if (!diag) tn_load<false>(G, ldg, i0, n, k, it.k1, sc, rI);
        }
        if (di) {
            w0 = (k < it.k1) ? di[k] : 0.0;
            w1 = (k + 1 < it.k1) ? di[k + 1] : 0.0;
        }
    };
    auto stash = [&](int s) {
        tn_store(sJ(s), sc, sk, rJ, w0, w1);
        if (!diag) tn_store(sI(s), sc, sk, rI, w0, w1);
    };

    if (nkt > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    const int skip = SYRK_SKIP & 15;
    const bool no_prio = (SYRK_SKIP >> 4) & 1;
    if (diag && (it.k1 - it.k0) >= 8192 && !((SYRK_SKIP >> 5) & 1)) {
        // ---- diagonal tiles (round 3): only the 36 16 x 16 blocks on / below the diagonal are computed, 9 per wave -- wave w owns
        //      block rows w and 7 - w (w + 1 and 8 - w blocks) -- instead of 16 per wave with 28 of the 64 above the diagonal
        //      (the whole wave (0, 1) among them).  At n = 8192 the diagonal tiles are 3 % of all tiles, at the batched engine's
        //      n = 512 they are 40 %.  Plain double-buffered loop; the MFMA predicates are wave-uniform.  Measured: n = 8192,
        //      m = 16384: 17.53 -> 17.30 ms; with short contractions (the batch: K = 1024) the plain loop costs more than the
        //      skipped blocks save (63 -> 58 k problem-iterations/s), hence the threshold on K.
        const int w = __builtin_amdgcn_readfirstlane(wave);     // scalar: the predicates below become s_cbranch (MFMA ignores EXEC)
        const int li2 = lane & 15, lk2 = lane >> 4;
        d4 acc0[8], acc1[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc0[t] = acc1[t] = d4{0.0, 0.0, 0.0, 0.0};
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nkt) fetch(kt + 1);
            const double* __restrict__ Xs = sJ(cur);
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                const double b0 = Xs[(16 * w + li2) * LDT_K + kk + lk2];
                const double b1 = Xs[(16 * (7 - w) + li2) * LDT_K + kk + lk2];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    if (t <= 7 - w) {
                        const double a = Xs[(16 * t + li2) * LDT_K + kk + lk2];
                        if (t <= w) acc0[t] = MFMA_F64(a, b0, acc0[t]);
                        acc1[t] = MFMA_F64(a, b1, acc1[t]);
                    }
                }
            }
            if (kt + 1 < nkt) stash(cur ^ 1);
            __syncthreads();
        }
        // lane holds D[row = (lane >> 4) + 4 r -> column j][col = lane & 15 -> row i]
        double* slab = it.slot >= 0 ? slabs + (int64_t)it.slot * TILE * TILE : nullptr;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int rb = half ? 7 - w : w;                 // block row
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (t <= rb) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int il = 16 * rb + li2, jl = 16 * t + lk2 + 4 * r;
                        const double v0 = half ? acc1[t][r] : acc0[t][r];
                        if (slab) {
                            slab[il + jl * TILE] = v0;
                        } else {
                            const int i = i0 + il, j = j0 + jl;
                            if (i < n && j < n && i >= j) {
                                double v = v0;
                                if (P) v += P[i + (int64_t)j * ldp];
                                C[i + (int64_t)j * ldc] = v;
                            }
                        }
                    }
                }
            }
        }
        return;
    }
    if (tile_fast && ((it.k1 - it.k0) % BK) == 0 && !skip) {
        // ---- software-pipelined main loop (interior tiles): the 8 global loads of tile kt+1 are issued one
        //      after every second MFMA quad of the first half, the 8 {scale, ds_write} units that stage it into
        //      the other LDS buffer after every second quad of the second half; sched_barriers pin that order so
        //      VMEM / VALU / LDS-write issue hides under the 64-cycle MFMAs instead of in front of the barrier.
        const double* gp[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gp[r] = G + (int64_t)(j0 + sc + 32 * r) * ldg + it.k0 + sk + BK;        // J operand, tile kt+1
            gp[4 + r] = G + (int64_t)(i0 + sc + 32 * r) * ldg + it.k0 + sk + BK;    // I operand
        }
        const double* dp = di ? di + it.k0 + sk + BK : nullptr;
        const int li = lane & 15, lk = lane >> 4;
        // static priority for the second-dispatched half of the workgroup (guide T5 / MICROARCH "static priority for the younger
        // half"): one s_setprio for the whole main loop, no per-phase flips.  Measured on the n = 8192, m = 16384 SYRK:
        // 17.20 -> 16.97 ms (all waves at priority 1, or alternate workgroups: no gain).  SYRK_SKIP bit 4 switches it off.
        if (!no_prio && wave >= 2) __builtin_amdgcn_s_setprio(1);
        // Diagonal tiles in THIS loop (round 3; short contractions -- the batched engine's K = 1024, where 4 of a problem's 10 tiles
        // are diagonal -- and every diagonal tile when bit 5 of SYRK_SKIP sends the long ones here too): of the four 64 x 64
        // quadrants, (i-half 0, j-half 1) lies above the diagonal and the two diagonal ones need 10 of their 16 blocks.  The wave
        // of the unused quadrant takes the j-blocks 2, 3 of quadrant (1, 0), its owner keeps j-blocks 0, 1 (8 blocks each); the
        // diagonal quadrants skip the 6 blocks above the diagonal (10 blocks): 10 instead of 16 MFMAs per 4-deep k group on the
        // critical wave.  All predicates are wave-uniform (scalar branches; MFMA ignores EXEC).  SYRK_SKIP bit 6: off.
        unsigned tmask = 0xFu;                 // the j-blocks (t) this wave computes
        bool tri = false;                      // only blocks with i-block u >= j-block t
        const bool dmask = DMASK && diag && it.slot < 0 && !((SYRK_SKIP >> 6) & 1);
        if (DMASK && dmask) {
            const int w = __builtin_amdgcn_readfirstlane(wave);
            if (w == 2) { wj = 0; wi = 1; tmask = 0xCu; }
            else if (w == 1) tmask = 0x3u;
            else tri = true;
        }
        auto main_loop = [&](auto DMc) {
        constexpr bool DM = decltype(DMc)::value;
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            const bool has_next = kt + 1 < nkt;
            const double* __restrict__ Js = sJ(cur) + wj * 64 * LDT_K;
            const double* __restrict__ Is = sI(cur) + wi * 64 * LDT_K;
            double* nJ = sJ(cur ^ 1);
            double* nI = sI(cur ^ 1);
            d2u ld[8];
            d2u wv = {1.0, 1.0};
            double a[2][4], b[2][4];          // operand fragments, double buffered across the 4-deep k groups
#pragma unroll
            for (int t = 0; t < 4; ++t) a[0][t] = Js[(t * 16 + li) * LDT_K + lk];
#pragma unroll
            for (int u = 0; u < 4; ++u) b[0][u] = Is[(u * 16 + li) * LDT_K + lk];
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                const int pb = (kk / 4) & 1;
                if (kk + 4 < BK) {            // prefetch the next group's fragments before this group's MFMAs
#pragma unroll
                    for (int t = 0; t < 4; ++t) a[pb ^ 1][t] = Js[(t * 16 + li) * LDT_K + kk + 4 + lk];
#pragma unroll
                    for (int u = 0; u < 4; ++u) b[pb ^ 1][u] = Is[(u * 16 + li) * LDT_K + kk + 4 + lk];
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (!DM || ((tmask >> t) & 1u)) {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (!DM || !tri || u >= t) acc[t][u] = MFMA_F64(a[pb][t], b[pb][u], acc[t][u]);
                    }
                    const int q = (kk / 4) * 4 + t;          // quad index 0..15 (compile-time after unrolling)
                    if (has_next) {
                        if (q < 8) {
                            if (q < 4 || !diag) ld[q] = *reinterpret_cast<const d2u*>(gp[q]);
                            if (q == 7 && dp) wv = *reinterpret_cast<const d2u*>(dp);
                        } else {
                            const int s8 = q - 8;
                            if (s8 < 4 || !diag) {
                                d2 v = {ld[s8].x * wv.x, ld[s8].y * wv.y};
                                double* dst = (s8 < 4 ? nJ : nI) + (sc + 32 * (s8 & 3)) * LDT_K + sk;
                                *reinterpret_cast<d2*>(dst) = v;
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) gp[r] += BK;
            if (dp) dp += BK;
            __syncthreads();
        }
        };
        if (DMASK && dmask) {
            main_loop(std::true_type{});
            // epilogue of a masked diagonal tile: only the blocks this wave computed (i >= j sorts out the diagonal quadrants)
            const int li = lane & 15, lq = lane >> 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!((tmask >> t) & 1u)) continue;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = i0 + wi * 64 + u * 16 + li;
                        const int j = j0 + wj * 64 + t * 16 + lq + 4 * r;
                        if (i >= j) {                        // (tile_fast: the whole tile is inside the matrix)
                            double v = acc[t][u][r];
                            if (P) v += P[i + (int64_t)j * ldp];
                            C[i + (int64_t)j * ldc] = v;
                        }
                    }
            }
            return;
        }
        main_loop(std::false_type{});
    } else {
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            if (kt + 1 < nkt && !(skip & 1)) fetch(kt + 1);
            wave_mma<LDT_K, 1>(sJ(cur) + wj * 64 * LDT_K, sI(cur) + wi * 64 * LDT_K, acc, lane);
            if (kt + 1 < nkt && !(skip & 2)) stash(cur ^ 1);
            if (!(skip & 4)) __syncthreads();
        }
    }

    // ---- epilogue: lane holds D[row=(lane>>4)+4r -> j][col=lane&15 -> i]
    const int li = lane & 15, lq = lane >> 4;
    if (it.slot < 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = i0 + wi * 64 + u * 16 + li;
                    const int j = j0 + wj * 64 + t * 16 + lq + 4 * r;
                    if (i < n && j < n && i >= j) {
                        double v = acc[t][u][r];
                        if (P) v += P[i + (int64_t)j * ldp];
                        C[i + (int64_t)j * ldc] = v;
                    }
                }
    } else {
        double* S = slabs + (int64_t)it.slot * TILE * TILE;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int il = wi * 64 + u * 16 + li;
                    const int jl = wj * 64 + t * 16 + lq + 4 * r;
                    S[il + jl * TILE] = acc[t][u][r];
                }
    }
}

template <bool DMASK>
__global__ __launch_bounds__(256, 2) void syrk_tn_kernel(
    const double* __restrict__ G, int64_t ldg, const double* __restrict__ di, int n, int fast_ok,
    const SyrkItem* __restrict__ items, double* __restrict__ C, int64_t ldc,
    const double* __restrict__ P, int64_t ldp, double* __restrict__ slabs, BatchStrides bs) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    {   // batched problems along blockIdx.z (all strides 0 for a single problem)
        const int64_t bz = blockIdx.z;
        G += bz * bs.a;
        if (di) di += bz * bs.b;
        C += bz * bs.c;
        if (P) P += bz * bs.d;
    }
    // stream-K: a workgroup of the remainder round owns an equal share of that round's k-steps, i.e. the end of one tile's
    // contraction and the beginning of the next one's -- a chain of segments, each with its own partial slab
    int idx = blockIdx.x;
    const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    for (;;) {
        const SyrkItem it = items[idx];
        syrk_tile<DMASK>(it, G, (int)ldg, di, n, fast_ok, C, (int)ldc, P, (int)ldp, slabs, smem, wave_s);
        if (it.next == 0) break;
        idx = it.next - 1;
        __syncthreads();                   // the segment's last reads of the stage buffers are done before the next one fills them
    }
}

// Deterministic fix-up for split tiles: C = P + slab[first] + slab[first+1] + ...  (fixed order)
__global__ __launch_bounds__(256) void syrk_reduce_kernel(const SyrkItem* __restrict__ tiles, int n,
                                                          const double* __restrict__ slabs,
                                                          double* __restrict__ C, int64_t ldc,
                                                          const double* __restrict__ P, int64_t ldp) {
    const SyrkItem it = tiles[blockIdx.x];
    const int i0 = it.ti * TILE, j0 = it.tj * TILE;
    // grid.y = 16 slices of 8 tile columns: enough workgroups to stream the slabs at HBM speed
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = blockIdx.y * 1024 + r * 256 + threadIdx.x;
        const int il = e & (TILE - 1), jl = e >> 7;
        const int i = i0 + il, j = j0 + jl;
        if (i < n && j < n && i >= j) {
            double v = P ? P[i + (int64_t)j * ldp] : 0.0;
            const double* __restrict__ sl = slabs + (int64_t)it.first * TILE * TILE + e;
            int s2 = 0;
            for (; s2 + 4 <= it.nparts; s2 += 4) {            // fixed summation order, four loads in flight
                const double a0 = sl[(int64_t)s2 * TILE * TILE], a1 = sl[(int64_t)(s2 + 1) * TILE * TILE];
                const double a2 = sl[(int64_t)(s2 + 2) * TILE * TILE], a3 = sl[(int64_t)(s2 + 3) * TILE * TILE];
                v += a0; v += a1; v += a2; v += a3;
            }
            for (; s2 < it.nparts; ++s2) v += sl[(int64_t)s2 * TILE * TILE];
            C[i + (int64_t)j * ldc] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Host: static schedule.  Tiles of the lower triangle are ordered by 8x8 super-tiles so that the
// workgroups one XCD runs concurrently (ids = x mod 8, observed dispatch) share operand panels in
// that XCD's private L2; whole rounds run unsplit tiles; the remainder round is stream-K: its k-steps
// are dealt out evenly over all workgroup slots (a workgroup finishes one tile's contraction and begins
// the next one's), with slabs summed in a fixed order by syrk_reduce_kernel.
// (Round 4 also built an XCD-local synchronisation of the k progress of the 64 workgroups an XCD runs side by side, so
//  that shared panels are fetched once per XCD: HBM traffic 35.3 -> 10.6 GB per launch, L2 hit rate 50 -> 84 %, kernel
//  7.5-24 % slower, clock unchanged -- removed; DESIGN 4b', profiles/r04_syrk_xcd_sync_experiment.txt, commit 4e2b5cc.)
// ---------------------------------------------------------------------------------------------------
// the static schedule itself (host only, no device memory): segments in launch order, the split tiles, the slab count
void make_syrk_items(int n, int K, int num_cus, bool allow_split, std::vector<SyrkItem>& items, int& nlaunch,
                     std::vector<SyrkItem>& split_tiles, int& nslabs) {
    items.clear();
    split_tiles.clear();
    nslabs = 0;
    nlaunch = 0;
    if (n <= 0) return;
    const int nt = (n + TILE - 1) / TILE;
    std::vector<std::pair<int, int>> seq;
    const int ns = (nt + 7) / 8;
    for (int SI = 0; SI < ns; ++SI)
        for (int SJ = 0; SJ <= SI; ++SJ)
            for (int a = 0; a < 8; ++a)
                for (int b = 0; b < 8; ++b) {
                    const int ti = SI * 8 + a, tj = SJ * 8 + b;
                    if (ti < nt && tj <= ti) seq.emplace_back(ti, tj);
                }
    const int T = (int)seq.size();
    const int slots = std::max(8, 2 * num_cus);
    const int nfull = (T / slots) * slots;
    const int R = T - nfull;
    auto full_item = [&](int t) { return SyrkItem{seq[t].first, seq[t].second, 0, K, -1, 0, 0, 0}; };

    // XCD-contiguous permutation of the full tiles: id -> xcd = id % 8 gets a contiguous chunk
    {
        const int N = nfull, q = N / 8, r = N % 8;
        items.resize(N);
        for (int id = 0; id < N; ++id) {
            const int x = id % 8, l = id / 8;
            const int pos = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l;
            items[id] = full_item(pos);
        }
    }
    nlaunch = nfull;
    if (R == 0) return;
    const int nkt = (K + BK - 1) / BK;
    const int64_t U = (int64_t)R * nkt;                 // k-steps of the remainder round
    const int W = (int)std::min<int64_t>(slots, U / 8); // its workgroups: at least 8 k-steps each
    if (!allow_split || W <= R) {                       // (short contractions, batched launches: whole tiles)
        for (int t = nfull; t < T; ++t) items.push_back(full_item(t));
        nlaunch = T;
        return;
    }
    // stream-K: workgroup w owns the k-steps [U w / W, U (w+1) / W) of the concatenated remainder tiles
    struct Seg { int t, ka, kb, w; };
    std::vector<Seg> segs;
    for (int w = 0; w < W; ++w) {
        int64_t u0 = U * w / W;
        const int64_t u1 = U * (w + 1) / W;
        while (u0 < u1) {
            const int t = (int)(u0 / nkt);
            const int64_t ue = std::min<int64_t>(u1, (int64_t)(t + 1) * nkt);
            segs.push_back({t, (int)(u0 - (int64_t)t * nkt) * BK, std::min(K, (int)(ue - (int64_t)t * nkt) * BK), w});
            u0 = ue;
        }
    }
    // slabs: the pieces of a tile are consecutive in segs (k order); a tile in one piece needs none
    std::vector<SyrkItem> seg_items(segs.size());
    for (size_t a = 0; a < segs.size();) {
        size_t b = a;
        while (b < segs.size() && segs[b].t == segs[a].t) ++b;
        const int parts = (int)(b - a), t = nfull + segs[a].t;
        if (parts == 1) {
            seg_items[a] = full_item(t);
        } else {
            for (size_t c = a; c < b; ++c)
                seg_items[c] = SyrkItem{seq[t].first, seq[t].second, segs[c].ka, segs[c].kb, nslabs + (int)(c - a), nslabs, parts, 0};
            split_tiles.push_back(SyrkItem{seq[t].first, seq[t].second, 0, K, -1, nslabs, parts, 0});
            nslabs += parts;
        }
        a = b;
    }
    // launch order: the first segment of every workgroup, then the continuation segments (chained through `next`)
    std::vector<int> where(segs.size(), -1);
    int extra = nfull + W;
    for (size_t a = 0; a < segs.size();) {
        size_t b = a;
        while (b < segs.size() && segs[b].w == segs[a].w) ++b;
        where[a] = nfull + segs[a].w;
        for (size_t c = a + 1; c < b; ++c) where[c] = extra++;
        a = b;
    }
    items.resize(extra);
    for (size_t c = 0; c < segs.size(); ++c) {
        items[where[c]] = seg_items[c];
        if (c + 1 < segs.size() && segs[c + 1].w == segs[c].w) items[where[c]].next = where[c + 1] + 1;
    }
    nlaunch = nfull + W;
}

int build_syrk_plan(SyrkPlan& plan, int n, int K, int num_cus, bool allow_split) {
    free_syrk_plan(plan);
    plan.n = n;
    plan.K = K;
    if (n <= 0) return 0;
    std::vector<SyrkItem> items, split_tiles;
    int nslabs = 0, nlaunch = 0;
    make_syrk_items(n, K, num_cus, allow_split, items, nlaunch, split_tiles, nslabs);
    plan.nitems = nlaunch;
    plan.nitems_total = (int)items.size();
    plan.nslabs = nslabs;
    plan.nsplit_tiles = (int)split_tiles.size();
    KKT_HIP_CHECK(DEV_ALLOC(&plan.d_items, sizeof(SyrkItem) * std::max<size_t>(1, items.size())));
    KKT_HIP_CHECK(memcpy_sync(plan.d_items, items.data(), sizeof(SyrkItem) * items.size(), hipMemcpyHostToDevice));
    if (!split_tiles.empty()) {
        KKT_HIP_CHECK(DEV_ALLOC(&plan.d_split_tiles, sizeof(SyrkItem) * split_tiles.size()));
        KKT_HIP_CHECK(memcpy_sync(plan.d_split_tiles, split_tiles.data(), sizeof(SyrkItem) * split_tiles.size(),
                                hipMemcpyHostToDevice));
        KKT_HIP_CHECK(DEV_ALLOC(&plan.d_slabs, sizeof(double) * (size_t)nslabs * TILE * TILE));
    }
    return 0;
}

void free_syrk_plan(SyrkPlan& plan) {
    if (plan.d_items) (void)dev_free(plan.d_items);
    if (plan.d_split_tiles) (void)dev_free(plan.d_split_tiles);
    if (plan.d_slabs) (void)dev_free(plan.d_slabs);
    plan = SyrkPlan();
}

static constexpr size_t kGemmLds = sizeof(double) * 4 * STAGE_DOUBLES;   // 73,728 B

static constexpr size_t kGemmLdsWide = 84 * 1024;                        // > 80 KiB: one workgroup per CU

int launch_syrk_scaled(const SyrkPlan& plan, const double* G, int64_t ldg, const double* di, double* C,
                       int64_t ldc, const double* P, int64_t ldp, hipStream_t st, hipEvent_t* kernel_events,
                       int nbatch, BatchStrides bs) {
    if (plan.n == 0) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(syrk_tn_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(syrk_tn_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
        attr_set = true;
    }
    const int fast_ok = ((reinterpret_cast<uintptr_t>(G) & 7) == 0) ? 1 : 0;   // 8-byte aligned pairs suffice
    if (kernel_events) KKT_HIP_CHECK(hipEventRecord(kernel_events[0], st));
    if (nbatch > 1 && plan.nsplit_tiles) {
        set_last_error("launch_syrk_scaled: batched launch needs an unsplit plan");
        return -1;
    }
    {
        // diagonal tiles: contractions of 8192 rows and more take the nine-blocks-per-wave path inside the kernel; shorter ones (the
        // batched engine: K = 1024, 4 of a problem's 10 tiles are diagonal) the block masks of the pipelined loop
        if (plan.K >= 8192)
            hipLaunchKernelGGL(syrk_tn_kernel<false>, dim3(plan.nitems, 1, nbatch), dim3(256), kGemmLds, st, G, ldg, di, plan.n,
                               fast_ok, plan.d_items, C, ldc, P, ldp, plan.d_slabs, bs);
        else
            // (work items dealt so that all tiles of a problem run on one XCD -- ids grouped by 8 problems -- were measured too:
            //  +-1 %, the operand panels of a problem come from the Infinity Cache either way; not kept)
            hipLaunchKernelGGL(syrk_tn_kernel<true>, dim3(plan.nitems, 1, nbatch), dim3(256), kGemmLds, st, G, ldg, di, plan.n,
                               fast_ok, plan.d_items, C, ldc, P, ldp, plan.d_slabs, bs);
    }
    KKT_HIP_CHECK(hipGetLastError());
    if (kernel_events) KKT_HIP_CHECK(hipEventRecord(kernel_events[1], st));
    if (plan.nsplit_tiles) {
        hipLaunchKernelGGL(syrk_reduce_kernel, dim3(plan.nsplit_tiles, 16), dim3(256), 0, st, plan.d_split_tiles,
                           plan.n, plan.d_slabs, C, ldc, P, ldp);
        KKT_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

// ===================================================================================================
// "NT" update: C -= A B'   (A: rows of C, M-major; B: cols of C, M-major; K small)
// ===================================================================================================
// staging map: thread -> idx pair (tid&63)*2, k = (tid>>6) + 4 r
template <bool FAST>
__device__ __forceinline__ void nt_load(const double* __restrict__ X, int64_t ldx, int row0, int nrows,
                                        int k0, int K, int tid, double (&reg)[8]) {
    const int ip = (tid & 63) * 2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + (tid >> 6) + 4 * r;
        const double* p = X + (int64_t)k * ldx + row0 + ip;
        if (FAST) {
            const d2u v = *reinterpret_cast<const d2u*>(p);
            reg[2 * r] = v.x;
            reg[2 * r + 1] = v.y;
        } else {
            const bool kok = k < K;
            reg[2 * r] = (kok && row0 + ip < nrows) ? p[0] : 0.0;
            reg[2 * r + 1] = (kok && row0 + ip + 1 < nrows) ? p[1] : 0.0;
        }
    }
}

__device__ __forceinline__ void nt_store(double* __restrict__ Xs, int tid, const double (&reg)[8]) {
    const int ip = (tid & 63) * 2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = (tid >> 6) + 4 * r;
        d2 v = {reg[2 * r], reg[2 * r + 1]};
        *reinterpret_cast<d2*>(Xs + k * LDT_M + ip) = v;
    }
}

// SYM: A == B, lower-triangular tile set (linear triangular blockIdx.x), only i >= j written.
template <bool SYM>
__global__ __launch_bounds__(256, 2) void nt_update_kernel(double* __restrict__ C, int64_t ldc,
                                                           const double* __restrict__ A, int64_t lda,
                                                           const double* __restrict__ B, int64_t ldb,
                                                           int M, int N, int K, int fast_ok, int64_t bstride,
                                                           const VbDesc* __restrict__ vb) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (vb) {      // variable batched fronts (SYM only): K carries the panel offset k0; C = A = B = base
        const VbDesc dd = vb[blockIdx.z];
        const int k0 = K;
        if (k0 >= dd.w) return;
        const int nb = min(128, dd.w - k0);
        const int m = dd.h - k0 - nb;
        const int nt = (m + TILE - 1) / TILE;
        if ((int)blockIdx.x >= nt * (nt + 1) / 2) return;
        ldc = lda = ldb = dd.h;
        A += dd.off + (k0 + nb) + (int64_t)k0 * dd.h;
        B = A;
        C += dd.off + (k0 + nb) + (int64_t)(k0 + nb) * dd.h;
        M = N = m;
        K = nb;
    } else {
        C += (int64_t)blockIdx.z * bstride;     // batched problems: C, A, B live in the same matrix
        A += (int64_t)blockIdx.z * bstride;
        B += (int64_t)blockIdx.z * bstride;
    }
    int ti, tj;
    if (SYM) {
        const int t = blockIdx.x;
        ti = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        while (ti * (ti + 1) / 2 > t) --ti;
        tj = t - ti * (ti + 1) / 2;
    } else {
        ti = blockIdx.x;
        tj = blockIdx.y;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wj = wave >> 1, wi = wave & 1;
    const int i0 = ti * TILE, j0 = tj * TILE;
    const bool diag = SYM && (ti == tj);
    const bool tile_fast = fast_ok && (i0 + TILE <= M) && (j0 + TILE <= N);

    auto sJ = [&](int s) -> double* { return smem + s * 2 * STAGE_DOUBLES; };
    auto sI = [&](int s) -> double* { return smem + s * 2 * STAGE_DOUBLES + STAGE_DOUBLES; };
    const int nkt = (K + BK - 1) / BK;
    double rJ[8], rI[8];
    auto fetch = [&](int kt) {
        const bool fast = tile_fast && ((kt + 1) * BK <= K);
        if (fast) {
            nt_load<true>(B, ldb, j0, N, kt * BK, K, tid, rJ);
            if (!diag) nt_load<true>(A, lda, i0, M, kt * BK, K, tid, rI);
        } else {
            nt_load<false>(B, ldb, j0, N, kt * BK, K, tid, rJ);
            if (!diag) nt_load<false>(A, lda, i0, M, kt * BK, K, tid, rI);
        }
    };
    auto stash = [&](int s) {      // the J operand is stored negated: the MFMAs then accumulate C - A B'
        double nJ[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) nJ[q] = -rJ[q];
        nt_store(sJ(s), tid, nJ);
        if (diag) nt_store(sI(s), tid, rJ);
        else nt_store(sI(s), tid, rI);
    };
    if (nkt > 0) fetch(0);
    // accumulators start as the C tile: its loads fly together with the first operand fetch, and the
    // epilogue is store-only (no read-modify-write latency at the tail of every tile)
    const int li = lane & 15, lq = lane >> 4;
    d4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi * 64 + u * 16 + li;
                const int j = j0 + wj * 64 + t * 16 + lq + 4 * r;
                acc[t][u][r] = (i < M && j < N && (!SYM || i >= j)) ? C[i + (int64_t)j * ldc] : 0.0;
            }
    if (nkt > 0) stash(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) fetch(kt + 1);
        wave_mma<1, LDT_M>(sJ(cur) + wj * 64, sI(cur) + wi * 64, acc, lane);
        if (kt + 1 < nkt) stash(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi * 64 + u * 16 + li;
                const int j = j0 + wj * 64 + t * 16 + lq + 4 * r;
                if (i < M && j < N && (!SYM || i >= j)) C[i + (int64_t)j * ldc] = acc[t][u][r];
            }
}

// ---------------------------------------------------------------------------------------------------
// Short-tile variants for the skinny updates on the Cholesky critical path (a few dozen 128 x 128 tiles on 256 CUs):
// a workgroup owns a 64 or 32 (rows) x 128 (columns) tile, the four waves split the columns (32 each, 2 x 4 or 2 x 2 MFMA
// tiles).  Two / four times as many workgroups, a half / quarter of the MFMA work each: the kernel's latency is one short tile.
// ---------------------------------------------------------------------------------------------------
template <bool FAST, int ROWS>
__device__ __forceinline__ void nt_load_rows(const double* __restrict__ X, int64_t ldx, int row0, int nrows, int k0, int K,
                                             int tid, double (&reg)[ROWS / 16]) {
    constexpr int TPK = ROWS / 2, KPP = 256 / TPK, PASSES = BK / KPP;    // threads per k, k rows per pass, passes
    const int ip = (tid % TPK) * 2;
#pragma unroll
    for (int r = 0; r < PASSES; ++r) {
        const int k = k0 + tid / TPK + KPP * r;
        const double* p = X + (int64_t)k * ldx + row0 + ip;
        if (FAST) {
            const d2u v = *reinterpret_cast<const d2u*>(p);
            reg[2 * r] = v.x;
            reg[2 * r + 1] = v.y;
        } else {
            const bool kok = k < K;
            reg[2 * r] = (kok && row0 + ip < nrows) ? p[0] : 0.0;
            reg[2 * r + 1] = (kok && row0 + ip + 1 < nrows) ? p[1] : 0.0;
        }
    }
}
template <int ROWS>
__device__ __forceinline__ void nt_store_rows(double* __restrict__ Xs, int tid, const double (&reg)[ROWS / 16]) {
    constexpr int TPK = ROWS / 2, KPP = 256 / TPK, PASSES = BK / KPP;
    const int ip = (tid % TPK) * 2;
#pragma unroll
    for (int r = 0; r < PASSES; ++r) {
        const int k = tid / TPK + KPP * r;
        d2 v = {reg[2 * r], reg[2 * r + 1]};
        *reinterpret_cast<d2*>(Xs + k * LDT_M + ip) = v;
    }
}

// ROWS = 64 or 32 rows of C per workgroup, 128 columns (32 per wave)
template <int ROWS, bool VB>
__global__ __launch_bounds__(256, 2) void nt_update_short_kernel(double* __restrict__ C, int64_t ldc,
                                                                 const double* __restrict__ A, int64_t lda,
                                                                 const double* __restrict__ B, int64_t ldb,
                                                                 int M, int N, int K, int fast_ok,
                                                                 const VbDesc* __restrict__ vb) {
    constexpr int UI = ROWS / 16;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int ti = blockIdx.x, tj = blockIdx.y;
    const int i0 = ti * ROWS, j0 = tj * TILE;
    if (VB) {      // fronts of a sparse level (symmetric update, lower part): K carries the panel offset k0; C = A = B = base
        const VbDesc dd = vb[blockIdx.z];
        const int k0 = K;
        if (k0 >= dd.w) return;
        const int nb = min(128, dd.w - k0);
        const int m = dd.h - k0 - nb;
        if (i0 >= m || j0 >= m || i0 + ROWS - 1 < j0) return;      // outside the front / strictly above the diagonal
        ldc = lda = ldb = dd.h;
        A += dd.off + (k0 + nb) + (int64_t)k0 * dd.h;
        B = A;
        C += dd.off + (k0 + nb) + (int64_t)(k0 + nb) * dd.h;
        M = N = m;
        K = nb;
    }
    const int tid = threadIdx.x, lane = tid & 63, wj = tid >> 6;
    const bool tile_fast = fast_ok && (i0 + ROWS <= M) && (j0 + TILE <= N);
    auto sJ = [&](int s) -> double* { return smem + s * 2 * STAGE_DOUBLES; };
    auto sI = [&](int s) -> double* { return smem + s * 2 * STAGE_DOUBLES + STAGE_DOUBLES; };
    const int nkt = (K + BK - 1) / BK;
    double rJ[8], rI[ROWS / 16];
    auto fetch = [&](int kt) {
        const bool fast = tile_fast && ((kt + 1) * BK <= K);
        if (fast) {
            nt_load<true>(B, ldb, j0, N, kt * BK, K, tid, rJ);
            nt_load_rows<true, ROWS>(A, lda, i0, M, kt * BK, K, tid, rI);
        } else {
            nt_load<false>(B, ldb, j0, N, kt * BK, K, tid, rJ);
            nt_load_rows<false, ROWS>(A, lda, i0, M, kt * BK, K, tid, rI);
        }
    };
    auto stash = [&](int s) {      // the J operand is stored negated: the MFMAs then accumulate C - A B'
        double nJ[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) nJ[q] = -rJ[q];
        nt_store(sJ(s), tid, nJ);
        nt_store_rows<ROWS>(sI(s), tid, rI);
    };
    if (nkt > 0) fetch(0);
    const int li = lane & 15, lq = lane >> 4;
    d4 acc[2][UI];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < UI; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + u * 16 + li;
                const int j = j0 + wj * 32 + t * 16 + lq + 4 * r;
                acc[t][u][r] = (i < M && j < N && (!VB || i >= j)) ? C[i + (int64_t)j * ldc] : 0.0;
            }
    if (nkt > 0) stash(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) fetch(kt + 1);
        const double* __restrict__ Js = sJ(cur) + wj * 32;
        const double* __restrict__ Is = sI(cur);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            double a[2], b[UI];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = Js[(t * 16 + li) + (kk + lq) * LDT_M];
#pragma unroll
            for (int u = 0; u < UI; ++u) b[u] = Is[(u * 16 + li) + (kk + lq) * LDT_M];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < UI; ++u) acc[t][u] = MFMA_F64(a[t], b[u], acc[t][u]);
        }
        if (kt + 1 < nkt) stash(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < UI; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + u * 16 + li;
                const int j = j0 + wj * 32 + t * 16 + lq + 4 * r;
                if (i < M && j < N && (!VB || i >= j)) C[i + (int64_t)j * ldc] = acc[t][u][r];
            }
}

static int nt_attr() {
    static bool done = false;
    if (!done) {
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(nt_update_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLdsWide));
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(nt_update_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nt_update_short_kernel<64, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nt_update_short_kernel<32, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nt_update_short_kernel<32, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
        done = true;
    }
    return 0;
}

int launch_syrk_nt_update(double* C, int64_t ldc, const double* A, int64_t lda, int nrows, int K,
                          hipStream_t st, int nbatch, int64_t bstride, bool one_wg_per_cu) {
    if (nrows <= 0 || K <= 0) return 0;
    if (int e = nt_attr()) return e;
    const int nt = (nrows + TILE - 1) / TILE;
    const int fast_ok = ((reinterpret_cast<uintptr_t>(A) & 7) == 0) ? 1 : 0;
    // one_wg_per_cu: ask for > half of the LDS so that a background (look-ahead) update leaves half of every CU
    // to the critical-path kernels of the other stream
    const size_t lds = one_wg_per_cu ? kGemmLdsWide : kGemmLds;
    hipLaunchKernelGGL(nt_update_kernel<true>, dim3(nt * (nt + 1) / 2, 1, nbatch), dim3(256), lds, st, C, ldc, A,
                       lda, A, lda, nrows, nrows, K, fast_ok, bstride, nullptr);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_syrk_nt_update_vb(double* base, const VbDesc* d_desc, int nfronts, int k0, int maxh, hipStream_t st) {
    if (int e = nt_attr()) return e;
    const int nt = (maxh - k0 - 1 + TILE - 1) / TILE;
    if (nt <= 0) return 0;
    // few tiles in the whole level (the top of the supernodal tree): 32-row tiles, four times the workgroups
    constexpr int vb_short_max = 1024;
    if ((int64_t)nfronts * (nt * (nt + 1) / 2) <= vb_short_max) {
        hipLaunchKernelGGL((nt_update_short_kernel<32, true>), dim3((maxh - k0 - 1 + 31) / 32, nt, nfronts), dim3(256), kGemmLds, st, base,
                           (int64_t)0, base, (int64_t)0, base, (int64_t)0, 0, 0, k0, 1, d_desc);
        KKT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(nt_update_kernel<true>, dim3(nt * (nt + 1) / 2, 1, nfronts), dim3(256), kGemmLds, st, base, (int64_t)0,
                       base, (int64_t)0, base, (int64_t)0, 0, 0, k0, 1, (int64_t)0, d_desc);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_gemm_nt_update(double* C, int64_t ldc, const double* A, int64_t lda, const double* B, int64_t ldb,
                          int M, int N, int K, hipStream_t st, int nbatch, int64_t bstride) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    if (int e = nt_attr()) return e;
    const int fast_ok = (((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 7) == 0) ? 1 : 0;
    // skinny updates of the Cholesky chain (few tiles): half-height tiles, twice the workgroups, half the latency
    constexpr bool half_ok = true;
    const int full_tiles = ((M + TILE - 1) / TILE) * ((N + TILE - 1) / TILE);
    constexpr int quarter_max = 128;
    if (half_ok && nbatch == 1 && full_tiles <= 192) {
        if (full_tiles <= quarter_max)
            hipLaunchKernelGGL((nt_update_short_kernel<32, false>), dim3((M + 31) / 32, (N + TILE - 1) / TILE), dim3(256), kGemmLds, st, C, ldc,
                               A, lda, B, ldb, M, N, K, fast_ok, nullptr);
        else
            hipLaunchKernelGGL((nt_update_short_kernel<64, false>), dim3((M + 63) / 64, (N + TILE - 1) / TILE), dim3(256), kGemmLds, st, C, ldc,
                               A, lda, B, ldb, M, N, K, fast_ok, nullptr);
        KKT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    hipLaunchKernelGGL(nt_update_kernel<false>, dim3((M + TILE - 1) / TILE, (N + TILE - 1) / TILE, nbatch), dim3(256),
                       kGemmLds, st, C, ldc, A, lda, B, ldb, M, N, K, fast_ok, bstride, nullptr);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}


// ===================================================================================================
// Congruence of a large 's' block, all columns of G at once:  Y_j = R' X_j R  (X_j = mat of column j of the block's rows,
// symmetric, given by its lower triangle; R = rti_k), written as packed lower triangles with the off-diagonals scaled by
// sqrt(2) -- misc.scale(trans = 'T', inverse = 'I') + misc.pack of the reference's kkt factories (misc_solvers.c:187-240,
// :412-550) for blocks too large for the LDS-resident kernel of cone_scale.hip.  Two batched FP64-MFMA products with the
// tile machinery of the "NT" update above (C = A B', A = R' for both):
//     pass 1:  Tt_j = R' X_j          (B = X_j read through its lower triangle)
//     pass 2:  Y_j  = R' Tt_j'        (B = Tt_j; lower triangle packed in the epilogue)
// blockIdx.z = column j.
// ===================================================================================================
template <bool SYMB>
__device__ __forceinline__ void cg_load(const double* __restrict__ X, int ld, int row0, int n, int k0, int tid, double (&reg)[8]) {
    const int ip = (tid & 63) * 2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int k = k0 + (tid >> 6) + 4 * r;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = row0 + ip + h;
            double v = 0.0;
            if (k < n && row < n) {
                if (SYMB && row < k) v = X[k + (int64_t)row * ld];
                else v = X[row + (int64_t)k * ld];
            }
            reg[2 * r + h] = v;
        }
    }
}

template <int PASS>
__global__ __launch_bounds__(256, 2) void sdp_congruence_kernel(const double* __restrict__ RT, int m,
                                                                const double* __restrict__ Bbase, int64_t bstride,
                                                                double* __restrict__ Cbase, int64_t cstride, double extra) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int ti = blockIdx.x, tj = blockIdx.y;
    const int i0 = ti * TILE, j0 = tj * TILE;
    if (PASS == 2 && i0 + TILE <= j0) return;                  // tile strictly above the diagonal: not stored
    const double* __restrict__ B = Bbase + (int64_t)blockIdx.z * bstride;
    double* __restrict__ C = Cbase + (int64_t)blockIdx.z * cstride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wj = wave >> 1, wi = wave & 1;
    auto sJ = [&](int s) -> double* { return smem + s * 2 * STAGE_DOUBLES; };
    auto sI = [&](int s) -> double* { return smem + s * 2 * STAGE_DOUBLES + STAGE_DOUBLES; };
    const int nkt = (m + BK - 1) / BK;
    double rJ[8], rI[8];
    auto fetch = [&](int kt) {
        cg_load<PASS == 1>(B, m, j0, m, kt * BK, tid, rJ);
        cg_load<false>(RT, m, i0, m, kt * BK, tid, rI);
    };
    auto stash = [&](int s) {
        nt_store(sJ(s), tid, rJ);
        nt_store(sI(s), tid, rI);
    };
    d4 acc[4][4];
    zero_acc(acc);
    fetch(0);
    stash(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) fetch(kt + 1);
        wave_mma<1, LDT_M>(sJ(cur) + wj * 64, sI(cur) + wi * 64, acc, lane);
        if (kt + 1 < nkt) stash(cur ^ 1);
        __syncthreads();
    }
    const int li = lane & 15, lq = lane >> 4;
    const double r2 = 1.4142135623730951;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi * 64 + u * 16 + li;
                const int j = j0 + wj * 64 + t * 16 + lq + 4 * r;
                if (i >= m || j >= m) continue;
                if (PASS == 1) C[i + (int64_t)j * m] = acc[t][u][r];
                else if (i >= j) {
                    const int64_t idx = (int64_t)j * m - ((int64_t)j * (j - 1)) / 2 + (i - j);
                    C[idx] = extra * ((i == j) ? acc[t][u][r] : r2 * acc[t][u][r]);
                }
            }
}

__global__ void cg_transpose_kernel(const double* __restrict__ R, double* __restrict__ RT, int m) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < m * m) RT[(e / m) + (int64_t)(e % m) * m] = R[e];
}

// in: rows of this block in column 0 of the input (column stride ldi); out: its packed rows in column 0 of the output (ldo).
// scratch: m*m doubles for R' plus m*m per column of a chunk.
int launch_sdp_congruence(const double* d_rti, int m, const double* in, int64_t ldi, double* out, int64_t ldo, int ncols,
                          double extra, double* scratch, size_t scratch_doubles, hipStream_t st) {
    const size_t mm = (size_t)m * m;
    if (scratch_doubles < 2 * mm) return -1;
    static bool attr = false;
    if (!attr) {
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sdp_congruence_kernel<1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sdp_congruence_kernel<2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGemmLds));
        attr = true;
    }
    double* RT = scratch;
    double* Tt = scratch + mm;
    hipLaunchKernelGGL(cg_transpose_kernel, dim3((unsigned)((mm + 255) / 256)), dim3(256), 0, st, d_rti, RT, m);
    const int chunk = (int)std::min<size_t>((scratch_doubles - mm) / mm, 65535);
    const int nt = (m + TILE - 1) / TILE;
    for (int c0 = 0; c0 < ncols; c0 += chunk) {
        const int nc = std::min(chunk, ncols - c0);
        hipLaunchKernelGGL(sdp_congruence_kernel<1>, dim3(nt, nt, nc), dim3(256), kGemmLds, st, RT, m, in + (int64_t)c0 * ldi, ldi, Tt,
                           (int64_t)mm, 1.0);
        hipLaunchKernelGGL(sdp_congruence_kernel<2>, dim3(nt, nt, nc), dim3(256), kGemmLds, st, RT, m, Tt, (int64_t)mm,
                           out + (int64_t)c0 * ldo, ldo, extra);
    }
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Microbenchmark: issue-bound v_mfma_f64_16x16x4_f64 rate (8 independent accumulators per wave,
// 2 waves per SIMD).  Confirms the FP64 matrix peak that bench.py's roofline line divides by.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void mfma_f64_peak_kernel(double* out, int iters) {
    d4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = MFMA_F64(a, b, acc[i]);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;   // keep the chain alive
}

int run_mfma_f64_peak(int iters, int num_cus, float* tflops) {
    double* d = nullptr;
    KKT_HIP_CHECK(DEV_ALLOC(&d, 8));
    hipEvent_t a, b;
    KKT_HIP_CHECK(hipEventCreate(&a));
    KKT_HIP_CHECK(hipEventCreate(&b));
    const int blocks = num_cus * 2;
    hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, nullptr, d, 16);   // warm-up
    KKT_HIP_CHECK(hipEventRecord(a, nullptr));
    hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, nullptr, d, iters);
    KKT_HIP_CHECK(hipEventRecord(b, nullptr));
    KKT_HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    KKT_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    const double flops = (double)blocks * 4 /*waves*/ * iters * 8.0 * 2048.0;
    if (tflops) *tflops = (float)(flops / (ms * 1e-3) / 1e12);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    (void)dev_free(d);
    return 0;
}

}  // namespace mi355kkt

// mi355kkt -- round 6: triangular solves in 512-row hops over ALL compute units.
//
// Replaces, for one right-hand side and a factor whose 128 x 128 diagonal-block inverses are known (potrf_tiles_kernel), the
// reference's blas.trsv / lapack.potrs (/root/reference/src/C/blas.c:1806, lapack.c:1553) inside kkt_chol2's solve
// (/root/reference/src/python/misc.py:1529, :1555).
//
// What rounds 3-5 measured: with one workgroup per 128-row block row a hop costs max(strip, hand-off + arithmetic)
// = 3.9 us, 64 hops at n = 8192, and the owner of the last block row streams 8 MB on its own at ~45 GB/s.  Here
//   * a hop is 512 rows: 16 hops at n = 8192, 4 at n = 2048;
//   * a block row belongs to 512 / R workgroups of R rows each (R = 32 at n = 8192: all 256 compute units stream L, every byte
//     of it exactly once per solve, 1 MB per workgroup on average);
//   * the 512 x 512 diagonal block is applied as an explicit inverse (block_inverse512, below: formed from the 128 x 128 inverses
//     with four small products per factorisation) -- one exchange of the block's 512 right-hand-side entries among its
//     workgroups, one product from LDS;
//   * accuracy as in trsv_pair_kernel (round 4): the explicit inverse makes the block solve only conditionally stable, so the
//     whole solve gets ONE step of fixed-precision iterative refinement, run as a second sweep one hop behind the first:
//         sweep 1:  t_j = rhs_j - sum_{i<j} L_ji x0_i,   x0_j = M_j t_j,   e_j = t_j - L_jj x0_j
//         sweep 2:  b_j = e_j   - sum_{i<j} L_ji d_i,    d_j  = M_j b_j,   x_j = x0_j + d_j
//     (e = rhs - L x0 exactly; Skeel 1980, Higham Thm 12.3: backward stable while eps cond(L_jj) << 1.)
//   * BOTH sweeps use one pass over L: a workgroup keeps the strip L(rows, block i) in registers until x0_i AND d_i have arrived
//     (d_i follows x0_i by one hop), except for the last strip in front of its own block, where x0 is consumed at once (that is
//     the critical path of sweep 1) and d when it comes.  Two register buffers: one strip held, the next one in flight.
// Hand-offs are the data-tagged 8-byte granules of blas2.hip ({epoch, half a double}, relaxed agent-scope stores / polls): four
// sets of 1024 granules per block -- x0, d, and the two intra-block exchanges t, b.  Dependencies only point to workgroups with
// a smaller blockIdx, every spin is bounded, a timeout sets *err.  Fixed summation order: bit-reproducible.
#include "kkt_common.h"

#include <algorithm>

namespace mi355kkt {

typedef unsigned int u32;
typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int WB = 512;          // rows per hop
// developer aid (-DMI355KKT_DEBUG builds only): 16 stamps per workgroup (s_memrealtime, 100 MHz, one clock for all compute units)
#ifdef MI355KKT_DEBUG
__device__ long long* g_wide_ts = nullptr;
#define WIDE_TS(k_) do { if (g_wide_ts && tid == 0) g_wide_ts[(int64_t)w_ts * 16 + (k_)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WIDE_TS(k_) do { } while (0)
#endif

// R rows per workgroup, T threads: T / R threads per row, each with CPT = WB R / T columns of a strip in registers.  LDS: my R rows
// of the block inverse (one padding column per column group: the threads of a wave that share a row index hit different banks),
// four 512-entry blocks of right-hand-side data (padded likewise), the reduction buffers
template <int R>
struct WideGeom {
    static constexpr int T = 512;          // (256 threads -- measured -- pull the strips half as fast: 190 against 123 us at n = 8192)
    static constexpr int CG = T / R;        // column groups (threads per row)
    static constexpr int CPT = WB / CG;     // columns per thread and strip
    static constexpr int WG4 = 2 * WB / T;  // granules each thread polls per block
    static constexpr int XP = WB + CG;      // padded length of a block of right-hand-side data in LDS
    static constexpr size_t lds_bytes = sizeof(double) * ((size_t)(WB + CG) * R + 4 * XP + 2 * T);
};

// More slices of R rows than compute units (n / 16 > #CUs): the grid is #CUs workgroups and a workgroup takes slices blockIdx,
// blockIdx + gridDim, ... one after the other -- the second one the moment its first is done (a second wave of workgroups trickles
// in instead, as the dispatcher finds slots: measured, 45 us of stall at n = 8192).  Dependencies still only point to smaller slices.
template <int R, bool TRANS>
__global__ __launch_bounds__(WideGeom<R>::T) void trsv_wide_kernel(const double* __restrict__ L, int64_t ldl, int n, double* x, u32 epoch,
                                                        int* err, u64* gran, const double* __restrict__ m512, unsigned lbytes) {
    constexpr int CG = WideGeom<R>::CG, CPT = WideGeom<R>::CPT, T = WideGeom<R>::T, WG4 = WideGeom<R>::WG4, XP = WideGeom<R>::XP;
    extern __shared__ double smem[];
    double* Ms = smem;                 // [WB + CG][R]: my R rows of M_j (forward) / M_j' (backward), column by column
    double* xb = Ms + (WB + CG) * R;   // [4][XP]: blocks of x0 / d / t / b as they arrive (see `wait`)
    double* red = xb + 4 * XP;         // [2][T]: partial sums of the column groups
    const int tid = threadIdx.x, r = tid % R, g = tid / R;
    const int NBk = (n + WB - 1) / WB;
    bool timeout = false;
    const int nsl = (n + R - 1) / R;                       // slices of R rows; the last one may be partial (any order n is served)
    for (int w = blockIdx.x; w < nsl; w += gridDim.x) {
    const int w_ts = w;
    (void)w_ts;
    WIDE_TS(0);
    const int row0 = (TRANS ? nsl - 1 - w : w) * R;        // slice order = dependency order
    const int j = row0 / WB, j0 = j * WB;
    const int idx = row0 + r;                              // my row of L (forward) / of L' (backward)
    const int rb = row0 - j0;
    u64* gX = gran;
    u64* gD = gran + (int64_t)NBk * 2 * WB;
    u64* gT = gran + (int64_t)2 * NBk * 2 * WB;
    u64* gB = gran + (int64_t)3 * NBk * 2 * WB;
    const int nsteps = TRANS ? NBk - 1 - j : j;
    const int c0 = g * CPT;
    const int cp = c0 + g;                                 // my first column in the padded LDS layouts

    // ---- strips: q < nsteps: block (j, blk(q)); the diagonal block (triangle only) after them.
    // Buffer loads: ONE descriptor for all of L (scalar registers), one 32-bit per-thread byte offset, the column as a scalar
    // offset -- 64-bit per-load vector addresses cost two registers per element of a strip.  Out-of-range offsets read zeros.
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(L), 0, (int)lbytes, 0x00020000);
    const unsigned voffb = (unsigned)((idx + (int64_t)c0 * ldl) * 8);
    const unsigned ldb = (unsigned)(ldl * 8);
    auto ldg = [&](unsigned soff) -> double {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsL, voffb, soff, 0));
    };
    // (conditional loads into loop-carried register arrays multiply the compiler's register demand, so the loop below only ever
    //  issues UNCONDITIONAL loads of whole off-diagonal blocks; the ragged last block and the diagonal strips are loaded outside it)
    auto load_off = [&](double (&buf)[CPT], int q) {           // strip q < nsteps of a WHOLE block
        const unsigned sb0 = (unsigned)((TRANS ? NBk - 1 - q : q) * WB) * ldb;
#pragma unroll
        for (int c = 0; c < CPT; ++c) buf[c] = ldg(sb0 + (unsigned)c * ldb);              // (backward: the mirrored L')
    };
    auto load_first = [&](double (&buf)[CPT]) {                // strip 0: backward, it is the (possibly ragged) last block --
        const unsigned sb0 = (unsigned)((TRANS ? NBk - 1 : 0) * WB) * ldb;      // columns beyond n are out of the descriptor's range: zeros
#pragma unroll
        for (int c = 0; c < CPT; ++c) buf[c] = ldg(sb0 + (unsigned)c * ldb);
    };
    auto load_diag = [&](double (&buf)[CPT]) {                 // my rows of L_jj / L_jj': whole rows, the triangle is cut out by
        const unsigned sb0 = (unsigned)j0 * ldb;               // dot_diag (a select here would wait for the loads)
#pragma unroll
        for (int c = 0; c < CPT; ++c) buf[c] = ldg(sb0 + (unsigned)c * ldb);
    };
    // Products in chunks of 8 columns with a scheduling barrier between chunks: left alone, the compiler hoists all LDS operand
    // loads of a product above its FMAs
    auto dot = [&](const double (&buf)[CPT], const double* v) -> double {      // four independent chains
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const double* vv = v + cp;
#pragma unroll
        for (int c = 0; c < CPT; c += 8) {
            a0 = fma(buf[c], vv[c], a0);
            a1 = fma(buf[c + 1], vv[c + 1], a1);
            a2 = fma(buf[c + 2], vv[c + 2], a2);
            a3 = fma(buf[c + 3], vv[c + 3], a3);
            a0 = fma(buf[c + 4], vv[c + 4], a0);
            a1 = fma(buf[c + 5], vv[c + 5], a1);
            a2 = fma(buf[c + 6], vv[c + 6], a2);
            a3 = fma(buf[c + 7], vv[c + 7], a3);
            __builtin_amdgcn_sched_barrier(0);
        }
        return (a0 + a1) + (a2 + a3);
    };
    auto dot_diag = [&](const double (&buf)[CPT], const double* v) -> double {  // the same for my rows of the diagonal block
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const double* vv = v + cp;
        const int lim = TRANS ? j0 + c0 - idx : idx - (j0 + c0);            // forward: columns c <= lim; backward: c >= -lim... see below
#pragma unroll
        for (int c = 0; c < CPT; c += 4) {
            a0 = fma((TRANS ? c >= -lim : c <= lim) ? buf[c] : 0.0, vv[c], a0);
            a1 = fma((TRANS ? c + 1 >= -lim : c + 1 <= lim) ? buf[c + 1] : 0.0, vv[c + 1], a1);
            a2 = fma((TRANS ? c + 2 >= -lim : c + 2 <= lim) ? buf[c + 2] : 0.0, vv[c + 2], a2);
            a3 = fma((TRANS ? c + 3 >= -lim : c + 3 <= lim) ? buf[c + 3] : 0.0, vv[c + 3], a3);
            if ((c & 4) != 0) __builtin_amdgcn_sched_barrier(0);
        }
        return (a0 + a1) + (a2 + a3);
    };
    auto dot_m = [&](const double* v) -> double {                              // my part of row r of M_j times v
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const double* m = Ms + (int64_t)cp * R + r;
        const double* vv = v + cp;
#pragma unroll
        for (int c = 0; c < CPT; c += 8) {
            a0 = fma(m[(c) * R], vv[c], a0);
            a1 = fma(m[(c + 1) * R], vv[c + 1], a1);
            a2 = fma(m[(c + 2) * R], vv[c + 2], a2);
            a3 = fma(m[(c + 3) * R], vv[c + 3], a3);
            a0 = fma(m[(c + 4) * R], vv[c + 4], a0);
            a1 = fma(m[(c + 5) * R], vv[c + 5], a1);
            a2 = fma(m[(c + 6) * R], vv[c + 6], a2);
            a3 = fma(m[(c + 7) * R], vv[c + 7], a3);
            __builtin_amdgcn_sched_barrier(0);
        }
        return (a0 + a1) + (a2 + a3);
    };
    // sum over the column groups, in a fixed order; the result is meaningful in the threads of group 0.  `which` alternates so
    // that a buffer is never rewritten before a barrier has separated it from its readers
    auto reduce = [&](double part, int which) -> double {
        double* buf = red + which * T;
        buf[tid] = part;                                       // = buf[g * R + r]
        __syncthreads();
        double s = 0.0;
        if (g == 0) {
#pragma unroll
            for (int gg = 0; gg < CG; ++gg) s += buf[gg * R + r];
        }
        return s;
    };
    // Waits.  The data of wait number k lands in LDS pair k & 1 (xb + 2 (k & 1) XP: two blocks) and is consumed before the next
    // wait starts; every wait ends with a barrier, so a pair is rewritten only after a barrier has separated it from its readers.
    int nwait = 0;
    // the 2 * bw granules of block b of a set -> bw doubles (zeros beyond bw)
    auto wait_block = [&](const u64* gbase, int b) -> const double* {
        double* dst = xb + 2 * (nwait++ & 1) * XP;
        if (timeout) return dst;                               // (uniform: set from __syncthreads_or)
        const int ng = 2 * min(WB, n - b * WB);
        const u64* gp = gbase + (int64_t)b * 2 * WB + tid;
        u64 v[WG4];
        bool ok[WG4];
#pragma unroll
        for (int k = 0; k < WG4; ++k) { v[k] = 0; ok[k] = tid + T * k >= ng; }
        bool got = false;
        for (unsigned spins = 0; spins < (1u << 22); ++spins) {
#pragma unroll
            for (int k = 0; k < WG4; ++k)
                if (!ok[k]) {
                    v[k] = __hip_atomic_load(gp + T * k, RLX_AGENT);
                    ok[k] = (u32)(v[k] >> 32) == epoch;
                }
            bool all = true;
#pragma unroll
            for (int k = 0; k < WG4; ++k) all = all && ok[k];
            if (all) { got = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int k = 0; k < WG4; ++k) {                        // granule 2i / 2i+1 = low / high word of entry i
            const int wi = tid + T * k, i = wi >> 1;
            reinterpret_cast<u32*>(dst)[2 * (i + i / CPT) + (wi & 1)] = (u32)v[k];
        }
        if (__syncthreads_or(got ? 0 : 1)) timeout = true;
        return dst;
    };
    // two blocks in one polling loop: block ba of set ga at the returned pointer, block bb of set gb at + XP
    auto wait2 = [&](const u64* ga, int ba, const u64* gb, int bb) -> const double* {
        double* dst = xb + 2 * (nwait++ & 1) * XP;
        if (timeout) return dst;
        const int nga = 2 * min(WB, n - ba * WB), ngb = 2 * min(WB, n - bb * WB);
        const u64* gp = ga + (int64_t)ba * 2 * WB + tid;
        const u64* gq = gb + (int64_t)bb * 2 * WB + tid;
        u64 v[2 * WG4];
        bool ok[2 * WG4];
#pragma unroll
        for (int k = 0; k < 2 * WG4; ++k) { v[k] = 0; ok[k] = tid + T * (k % WG4) >= (k < WG4 ? nga : ngb); }
        bool got = false;
        for (unsigned spins = 0; spins < (1u << 22); ++spins) {
#pragma unroll
            for (int k = 0; k < 2 * WG4; ++k)
                if (!ok[k]) {
                    v[k] = __hip_atomic_load((k < WG4 ? gp : gq) + T * (k % WG4), RLX_AGENT);
                    ok[k] = (u32)(v[k] >> 32) == epoch;
                }
            bool all = true;
#pragma unroll
            for (int k = 0; k < 2 * WG4; ++k) all = all && ok[k];
            if (all) { got = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int k = 0; k < WG4; ++k) {
            const int wi = tid + T * k, i = wi >> 1, o = 2 * (i + i / CPT) + (wi & 1);
            reinterpret_cast<u32*>(dst)[o] = (u32)v[k];
            reinterpret_cast<u32*>(dst + XP)[o] = (u32)v[WG4 + k];
        }
        if (__syncthreads_or(got ? 0 : 1)) timeout = true;
        return dst;
    };
    auto publish = [&](u64* gbase, double v) {                 // threads of group 0: entry rb + r of block j
        u64* gp = gbase + (int64_t)j * 2 * WB + 2 * (rb + r);
        const u64 tag = (u64)epoch << 32;
        __hip_atomic_store(gp, tag | (u32)__double2loint(v), RLX_AGENT);
        __hip_atomic_store(gp + 1, tag | (u32)__double2hiint(v), RLX_AGENT);
    };

    // ---- start: strip 0 is requested first, then my rows of the block inverse travel through the second buffer into LDS, then
    // strip 1
    // (rows beyond n in a partial slice: their strips hold whatever lies behind the column in memory -- finite, never stored, never
    //  polled by anybody: the granules of a ragged block end at n; their rows of the block inverse are zero)
    double accA = (g == 0 && idx < n) ? x[idx] : 0.0, accB = 0.0;
    // the first block row's t_0 is its right-hand side: it goes out before anything else is requested, so that the exchange runs
    // while the rows of the block inverse are still on their way (3 us at the head of every solve's critical path)
    const bool first = nsteps == 0;
    if (R < 16 && first && g == 0) publish(gT, accA);
    double sa[CPT], sb[CPT];
    if (nsteps > 0) load_first(sa);
    {
        const double* Mj = m512 + (int64_t)j * 2 * WB * WB + (TRANS ? (int64_t)WB * WB : 0);
        const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(Mj), 0, WB * WB * 8, 0x00020000);
        const unsigned vm = (unsigned)(((rb + r) + c0 * WB) * 8);
#pragma unroll
        for (int c = 0; c < CPT; ++c)
            sb[c] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsM, vm, (unsigned)(c * WB * 8), 0));
#pragma unroll
        for (int c = 0; c < CPT; ++c) Ms[(int64_t)(cp + c) * R + r] = sb[c];
    }
    if (nsteps > 1) load_off(sb, 1);
    __syncthreads();                                           // Ms complete
    WIDE_TS(1);

    // ---- far field (strips in front of the last two): not on any critical path; x0 and d of a block are consumed together, when
    // both are there (d_q follows x0_q by about two hops).  Two strip buffers: one waiting for its block, one in flight.
    // (Measured and not kept: consuming x0_q at once and d_{q-1} with it -- three rotating buffers -- frees sweep 1 from sweep 2
    //  on paper, but the third strip spills at 16 rows per workgroup, and a spill reload is a vector memory operation that returns
    //  behind the strip requested in front of it: slower at every order.)
    auto blk = [&](int q) { return TRANS ? NBk - 1 - q : q; };
    auto pair_step = [&](double (&buf)[CPT], int q) {
        const double* xs = wait2(gX, blk(q), gD, blk(q));
        accA -= dot(buf, xs);
        accB -= dot(buf, xs + XP);
    };
    // ---- the last two strips and my own block.  Sweep 1 never waits for sweep 2 in here: x0 of the two blocks in front of mine
    // is consumed as it arrives, their d afterwards (the strips stay in registers); my rows of the diagonal block have a buffer
    // of their own, requested first
    auto tail = [&](double (&prev)[CPT], double (&last)[CPT]) {
        const int bl = blk(nsteps - 1), bp = blk(nsteps - 2);
        double dg[CPT];
        load_diag(dg);
        WIDE_TS(2);
        if (nsteps > 1) {
            const double* xs = wait2(gX, bp, gX, bl);
            accA -= dot(prev, xs);
            accA -= dot(last, xs + XP);
        } else if (nsteps > 0) {
            accA -= dot(last, wait_block(gX, bl));
        }
        WIDE_TS(3);
        // t_j complete: exchange it inside the block row, apply M_j, publish x0_j            (critical path of sweep 1)
        const double tsum = (R < 16 && first) ? accA : reduce(accA, 0);   // (first block row, 8 rows: published at the start)
        if (!(R < 16 && first) && g == 0) publish(gT, tsum);
        WIDE_TS(4);
        const double* ts = wait_block(gT, j);
        WIDE_TS(5);
        const double x0 = reduce(dot_m(ts), 1);
        if (g == 0) publish(gX, x0);
        WIDE_TS(6);
        // sweep 2: b_j = e_j - sum L_ji d_i, e_j = t_j - L_jj x0_j.  d of the block before last arrives about when x0_j is complete
        double part = accB;
        if (nsteps > 1) {
            const double* xs = wait2(gD, bp, gX, j);
            part -= dot(prev, xs);
            part -= dot_diag(dg, xs + XP);
        } else {
            part -= dot_diag(dg, wait_block(gX, j));
        }
        WIDE_TS(7);
        if (nsteps > 0) part -= dot(last, wait_block(gD, bl));
        WIDE_TS(8);
        const double bs = reduce(part, 0);
        if (g == 0) publish(gB, tsum + bs);                    //                                      (critical path of sweep 2)
        WIDE_TS(9);
        const double* bv = wait_block(gB, j);
        WIDE_TS(10);
        const double dj = reduce(dot_m(bv), 1);
        if (g == 0 && !timeout) {
            publish(gD, dj);
            if (idx < n) x[idx] = x0 + dj;
        }
        WIDE_TS(11);
    };
    // strips q and q + 1 are in sa / sb (loaded or in flight) at the top of every iteration; every load in here is unconditional
    const int F = nsteps - 2;                                  // far-field steps
    int q = 0;
    for (; q + 2 <= F; q += 2) {
        pair_step(sa, q);
        load_off(sa, q + 2);
        pair_step(sb, q + 1);
        load_off(sb, q + 3);
    }
    if (q < F) {                                               // F odd: one more, then prev = sb, last = sa
        pair_step(sa, q);
        load_off(sa, q + 2);
        tail(sb, sa);
    } else if (nsteps == 1) {
        tail(sb, sa);                                          // the only strip is in sa
    } else {
        tail(sa, sb);
    }
    }   // slices
    if (timeout && tid == 0) atomicExch(err, 1);
}

// ===================================================================================================
// 512 x 512 inverses of the diagonal blocks of L from the 128 x 128 ones (minv: per 128-block M then M', potrf_tiles_kernel):
//     M = [M_a 0; -M_b L_ba M_a  M_b]  applied twice (128 -> 256 -> 512), four stages of small products, 64 x 64 output tiles.
// m512: per 512-block 512 x 512 M (lower triangular, zeros above: cleared once at allocation) then M' -- both column-major, so the
// forward solve reads rows of M and the backward solve rows of M' coalesced.  A ragged last block (n a multiple of 128) keeps its
// missing rows / columns zero.
// ===================================================================================================
// C(64 x 64) = alpha A(64 x k0..k1) B(k0..k1 x 64), column-major; Ct (optional): the transposed tile, element (j, i) at
// Ct[j + i ldct].  Four waves, one 32 x 32 quarter each, v_mfma_f64_16x16x4 straight from global memory (the operands are a few
// hundred KB that the factorisation has just left in the caches): lane (li, lk) feeds A[i0 + li][k + lk] and B[k + lk][j0 + li] (see Bt below),
// register r of the result is C[i0 + li][j0 + (lane >> 4) + 4 r].
typedef double d4w __attribute__((ext_vector_type(4)));
// B arrives TRANSPOSED (Bt[j + k ldbt] = B[k][j]): both operands are then read with 16 consecutive lanes on 128 consecutive bytes
// (every B of the four stages has a transposed twin anyway: M', and the T products are stored both ways)
// arows: rows of the A tile that exist (an operand taken from L below a ragged last block: the rest reads as zero)
__device__ __forceinline__ void mfma_tile64(const double* __restrict__ A, int64_t lda, const double* __restrict__ Bt, int64_t ldbt,
                                            int k0, int k1, double alpha, double* C, int64_t ldc, double* Ct, int64_t ldct, int tid,
                                            int arows = 64, const double* zero = nullptr) {
    const int wv = tid >> 6, lane = tid & 63, li = lane & 15, lk = lane >> 4;
    const int i0 = 32 * (wv & 1), j0 = 32 * (wv >> 1);
    d4w acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[u][t] = d4w{0.0, 0.0, 0.0, 0.0};
    // rows of A that do not exist read a location that holds zero (`zero`), through the ADDRESS: a select on the loaded value would
    // make every request wait for its data
    const bool ra0 = i0 + li < arows, ra1 = i0 + 16 + li < arows;
    const double* Ap0 = ra0 ? A + (i0 + li) + (int64_t)lk * lda : zero;
    const double* Ap1 = ra1 ? A + (i0 + 16 + li) + (int64_t)lk * lda : zero;
    const int64_t lda0 = ra0 ? lda : 0, lda1 = ra1 ? lda : 0;
    const double* Bp = Bt + (j0 + li) + (int64_t)lk * ldbt;
    // (k0, k1: multiples of 64) sixteen MFMA steps' operands at a time, the NEXT sixteen requested before the matrix cores start on
    // the current ones: a product is one memory latency + its matrix-core time, not (k1 - k0) / 64 latencies.  The request past the
    // end re-reads the last chunk (unconditional loads: no select on a loop-carried register array)
    struct Ops { double a0[16], a1[16], b0[16], b1[16]; };
    auto load = [&](Ops& o, int k) {
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            o.a0[s4] = Ap0[(int64_t)(k + 4 * s4) * lda0];
            o.a1[s4] = Ap1[(int64_t)(k + 4 * s4) * lda1];
            o.b0[s4] = Bp[(int64_t)(k + 4 * s4) * ldbt];
            o.b1[s4] = Bp[16 + (int64_t)(k + 4 * s4) * ldbt];
        }
    };
    auto mma = [&](const Ops& o) {
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.b0[s4], o.a0[s4], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.b1[s4], o.a0[s4], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.b0[s4], o.a1[s4], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.b1[s4], o.a1[s4], acc[1][1], 0, 0, 0);
        }
    };
    Ops oa, ob;
    const int klast = k1 - 64;
    int k = k0;
    if (k < k1) load(oa, k);
    while (k < k1) {
        load(ob, min(k + 64, klast));
        mma(oa);
        k += 64;
        if (k >= k1) break;
        load(oa, min(k + 64, klast));
        mma(ob);
        k += 64;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = alpha * acc[u][t][r];
                const int ri = i0 + 16 * u + li, cj = j0 + 16 * t + lk + 4 * r;
                C[ri + (int64_t)cj * ldc] = v;
                if (Ct) Ct[cj + (int64_t)ri * ldct] = v;
            }
}

// ONE launch, 16 workgroups per 512-block, four stages separated by a barrier among the block's workgroups (cnt[b]: a counter that
// only ever grows -- `base` = what earlier launches have added to it):
// stage 1: T_p = L(2p+1, 2p) M_2p (p = 0, 1) + copies of the four 128 x 128 diagonal inverses into M / M'
// stage 2: M(2p+1, 2p) = -M_{2p+1} T_p
// stage 3: T2 = L(rows 256.., cols 0..255) M(0..255, 0..255)
// stage 4: M(256.., 0..255) = -M(256.., 256..) T2
// (the triangular factor of every product limits its k range)
__global__ __launch_bounds__(256) void block_inverse512_kernel(const double* __restrict__ L, int64_t ldl, int n,
                                                               const double* __restrict__ minv, double* __restrict__ m512,
                                                               double* __restrict__ scratch, u32* cnt, u32 base) {
    const int tid = threadIdx.x, t = blockIdx.x, b = blockIdx.y;
    const int o = b * WB;
    const int nbk = min(WB, n - o);                            // rows of this 512-block
    const int qn = (nbk + 127) / 128;                          // its 128-blocks (1..4), the last one possibly ragged
    double* M = m512 + (int64_t)b * 2 * WB * WB;
    double* Mt = M + (int64_t)WB * WB;
    double* T = scratch + (int64_t)b * 2 * 256 * 256;         // the transposed products (read again) ...
    double* T2 = T + 256 * 256;                               // ... and their plain copies (never read: mfma_tile64 writes both)
    constexpr int NB2 = 128 * 128;
    // (one cache operation per workgroup and direction: a release / acquire fence in every wave -- measured -- is what the kernel
    //  then consists of: 114 us at 64 workgroups, 280 us at 256)
    auto block_sync = [&](u32 target) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every wave drains its own stores
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // ... and they are written back before anybody is told
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(cnt + b, 1u, RLX_AGENT);
            for (unsigned spins = 0; spins < (1u << 24); ++spins) {
                if ((int)(__hip_atomic_load(cnt + b, RLX_AGENT) - target) >= 0) break;
                __builtin_amdgcn_s_sleep(2);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // what the others wrote is read from memory, not from a stale line
        }
        __syncthreads();
    };
    {   // stage 1
        if (t >= 12) {
            const int d = t - 12;                               // copy diagonal block d
            if (d < qn) {
                const double* src = minv + (int64_t)(4 * b + d) * 2 * NB2;
                // (sixteen elements of each in flight per thread: one at a time, the copy is a chain of 64 memory latencies -- measured,
                //  ~100 us, the whole formation)
                for (int e0 = tid; e0 < NB2; e0 += 16 * 256) {
                    double va[16], vb[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) { va[u] = src[e0 + 256 * u]; vb[u] = src[NB2 + e0 + 256 * u]; }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int e = e0 + 256 * u, i = e & 127, jc = e >> 7;
                        M[(128 * d + i) + (int64_t)(128 * d + jc) * WB] = va[u];
                        Mt[(128 * d + i) + (int64_t)(128 * d + jc) * WB] = vb[u];
                    }
                }
            }
        } else if (t < 8) {
            const int p = t >> 2, ti = (t >> 1) & 1, tj = t & 1;
            if (2 * p + 1 < qn) {
                const double* A = L + (o + 128 * (2 * p + 1) + 64 * ti) + (int64_t)(o + 128 * 2 * p) * ldl;
                const double* Bt = minv + (int64_t)(4 * b + 2 * p) * 2 * NB2 + NB2 + 64 * tj;       // M_2p' (rows 64 tj..)
                // only T' is needed (the B operand of stage 2); it lives in the first half of the block's scratch
                double* Tt = T + (int64_t)p * NB2;
                mfma_tile64(A, ldl, Bt, 128, 64 * tj, 128, 1.0, T + 2 * NB2 + (int64_t)p * NB2 + 64 * ti + (int64_t)(64 * tj) * 128, 128,
                            Tt + 64 * tj + (int64_t)(64 * ti) * 128, 128, tid, nbk - 128 * (2 * p + 1) - 64 * ti, M + WB);
            }
        }
    }
    block_sync(base + 16);
    if (t < 8) {   // stage 2
        const int p = t >> 2, ti = (t >> 1) & 1, tj = t & 1;
        if (2 * p + 1 < qn) {
            const double* A = minv + (int64_t)(4 * b + 2 * p + 1) * 2 * NB2 + 64 * ti;
            const double* Bt = T + (int64_t)p * NB2 + 64 * tj;                                      // T_p' (rows 64 tj..)
            const int ri = 128 * (2 * p + 1) + 64 * ti, cj = 128 * 2 * p + 64 * tj;
            mfma_tile64(A, 128, Bt, 128, 0, 64 * ti + 64, -1.0, M + ri + (int64_t)cj * WB, WB, Mt + cj + (int64_t)ri * WB, WB, tid);
        }
    }
    block_sync(base + 32);
    const int rows2 = 128 * (qn - 2);                          // (whole 128-blocks of storage; rows beyond nbk are zero)
    const int ti = t >> 2, tj = t & 3;
    const bool work = rows2 > 0 && 64 * ti < rows2;
    if (work) {   // stage 3
        const double* A = L + (o + 256 + 64 * ti) + (int64_t)o * ldl;
        const double* Bt = Mt + 64 * tj;                                                             // M(0..255, 0..255)' (rows 64 tj..)
        // T2 is 256 x 256 at most and only its transpose is read again: T2'[j + i 256]; the plain copy goes to the second scratch
        mfma_tile64(A, ldl, Bt, WB, 64 * tj, 256, 1.0, T2 + 64 * ti + (int64_t)(64 * tj) * 256, 256,
                    T + 64 * tj + (int64_t)(64 * ti) * 256, 256, tid, nbk - 256 - 64 * ti, M + WB);
    }
    block_sync(base + 48);
    if (work) {   // stage 4
        const double* A = M + (256 + 64 * ti) + (int64_t)256 * WB;
        const double* Bt = T + 64 * tj;                                                              // T2' (rows 64 tj..)
        const int ri = 256 + 64 * ti, cj = 64 * tj;
        mfma_tile64(A, WB, Bt, 256, 0, 64 * ti + 64, -1.0, M + ri + (int64_t)cj * WB, WB, Mt + cj + (int64_t)ri * WB, WB, tid);
    }
}

#ifdef MI355KKT_DEBUG
int set_wide_ts(long long* dptr) { return hipMemcpyToSymbol(HIP_SYMBOL(g_wide_ts), &dptr, sizeof(dptr)) == hipSuccess ? 0 : -2; }
#endif

int trsv_wide_rows(int n, int num_cus, bool any_order) {   // rows per workgroup, 0: this order is not served
    if (n < 1024 || (!any_order && n % 128) || (int64_t)n * n * 8 >= ((int64_t)1 << 31)) return 0;
    // the 32 (16 rows) or 64 (8 rows) workgroups of a block row exchange with each other: they must be DIFFERENT workgroups of the
    // launch, all co-resident -- a device (or partition) with fewer compute units than that keeps the round-4 kernels
    if (num_cus < 64) return 0;
    return (n + 7) / 8 <= num_cus ? 8 : 16;
}

// state: w.d_m512 (2 x 512 x 512 doubles per 512-block), w.d_m512_scratch, w.d_gran512 (4 sets x 1024 granules per block)
int launch_block_inverse512(const double* L, int64_t ldl, int n, PotrfWork& w, hipStream_t st) {
    if (n <= 0 || w.minv_n != n || w.minv_of != L || !w.d_minv) return -1;
    const int NBk = (n + WB - 1) / WB;
    if (w.m512_blocks < NBk) {
        if (w.d_m512) (void)dev_free(w.d_m512);
        if (w.d_m512_scratch) (void)dev_free(w.d_m512_scratch);
        if (w.d_gran512) (void)dev_free(w.d_gran512);
        w.d_m512 = w.d_m512_scratch = nullptr;
        w.d_gran512 = nullptr;
        w.m512_blocks = 0;
        w.m512_n = 0;
        const size_t mb = sizeof(double) * 2 * WB * WB * (size_t)NBk, gb = sizeof(u64) * 4 * 2 * WB * (size_t)NBk;
        KKT_HIP_CHECK(DEV_ALLOC(&w.d_m512, mb));
        KKT_HIP_CHECK(DEV_ALLOC(&w.d_m512_scratch, sizeof(double) * (2 * 256 * 256 * (size_t)NBk + NBk)));       // + the stage counters
        KKT_HIP_CHECK(hipMemsetAsync(w.d_m512_scratch + 2 * 256 * 256 * (size_t)NBk, 0, sizeof(double) * NBk, st));
        w.m512_launches = 0;
        KKT_HIP_CHECK(DEV_ALLOC(&w.d_gran512, gb));
        // the blocks above the diagonal of M (below it in M') are never written: zeros for good; granule tags start at epoch 0
        KKT_HIP_CHECK(hipMemsetAsync(w.d_m512, 0, mb, st));
        KKT_HIP_CHECK(hipMemsetAsync(w.d_gran512, 0, gb, st));
        w.m512_blocks = NBk;
        w.m512_shape_n = n;
    } else if (w.m512_shape_n != n) {
        // another order in the same storage: what an earlier, larger or differently ragged factor left must not survive
        // (m512_n = 0 only says "stale": a refactorisation of the same order rewrites exactly what it wrote before)
        KKT_HIP_CHECK(hipMemsetAsync(w.d_m512, 0, sizeof(double) * 2 * WB * WB * (size_t)NBk, st));
        w.m512_shape_n = n;
    }
    u32* cnt = reinterpret_cast<u32*>(w.d_m512_scratch + 2 * 256 * 256 * (size_t)w.m512_blocks);
    hipLaunchKernelGGL(block_inverse512_kernel, dim3(16, NBk), dim3(256), 0, st, L, ldl, n, w.d_minv, w.d_m512, w.d_m512_scratch, cnt,
                       48u * w.m512_launches++);
    KKT_HIP_CHECK(hipGetLastError());
    w.m512_n = n;
    w.m512_of = L;
    return 0;
}

template <int R>
static int launch_wide_r(const double* L, int64_t ldl, int n, double* x, int trans, u32 epoch, int* err, hipStream_t st, u64* gran,
                         const double* m512, int num_cus) {
    static bool attr_set = false;
    constexpr size_t lds = WideGeom<R>::lds_bytes;
    if (!attr_set) {
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trsv_wide_kernel<R, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(trsv_wide_kernel<R, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    // (never more workgroups than compute units: a workgroup that has gone on to its second slice waits for slices of workgroups that
    //  must be running already -- 8 rows at n = 4096 / 8192 with two workgroups per compute unit were measured: 58 us against 49 at
    //  4096, and a hand-off timeout at 8192, where the second 256 were not co-resident)
    const dim3 g(std::min((n + R - 1) / R, num_cus)), b(WideGeom<R>::T);
    const unsigned lbytes = (unsigned)(((int64_t)ldl * (n - 1) + n) * 8);      // (< 2 GB: checked by launch_trsv_wide)
    if (trans)
        hipLaunchKernelGGL((trsv_wide_kernel<R, true>), g, b, lds, st, L, ldl, n, x, epoch, err, gran, m512, lbytes);
    else
        hipLaunchKernelGGL((trsv_wide_kernel<R, false>), g, b, lds, st, L, ldl, n, x, epoch, err, gran, m512, lbytes);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

// x := L^-1 x (trans = 0) / L^-T x (trans != 0: needs the mirrored upper triangle); rows = trsv_wide_rows(n, #CUs) != 0
int launch_trsv_wide(const double* L, int64_t ldl, int n, double* x, int trans, unsigned int epoch, int* err, hipStream_t st,
                     PotrfWork& w, int rows, int num_cus) {
    if (w.m512_n != n || w.m512_of != L || !w.d_m512 || !w.d_gran512) return -1;
    if ((int64_t)ldl * n * 8 >= ((int64_t)1 << 31)) return -1;       // 32-bit buffer offsets (trsv_wide_kernel)
    if (rows == 8) return launch_wide_r<8>(L, ldl, n, x, trans, epoch, err, st, w.d_gran512, w.d_m512, num_cus);
    if (rows == 16) return launch_wide_r<16>(L, ldl, n, x, trans, epoch, err, st, w.d_gran512, w.d_m512, num_cus);
    return -1;
}

}  // namespace mi355kkt

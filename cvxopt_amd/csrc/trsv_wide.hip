// Round 5: triangular solves with one right-hand side as 256-row hops spread over eight compute units each
// (the device replacement of blas.trsv / lapack.potrs inside the hook's solve(): reference src/C/blas.c:1806, src/C/lapack.c:1553,
// called from misc.py:1527-1558).
//
// trsv_pair_kernel (blas2.hip, round 4) moves the solution forward 128 rows per hop, one workgroup per block row; a hop is
// hand-off + 64 FMAs + exchange + one 128 x 128 matrix-vector stage = 3.3 us, 16 hops at n = 2048 (60 us for 16.8 MB of L that
// sits in the caches), 64 at n = 8192 (0.25 ms: 1.08 TB/s).  What bounds it is the NUMBER of hops, not bytes.  Here
//
//   * the diagonal blocks are 256 x 256 and applied through their explicit inverses M2_k, formed once per factorisation from the
//     128 x 128 inverses the tile Cholesky already leaves behind (pair_inverse_kernel: [[M_a, 0], [-M_b L_ba M_a, M_b]]),
//   * a block row is owned by EIGHT workgroups of 32 rows each (8 threads per row, 32 columns per thread and step): the strips a
//     workgroup needs -- 32 x 256 of L per consumed block, 32 x 256 of M2 -- are 32 doubles per thread and are in registers
//     before the data they multiply arrives,
//   * a hop is: gather the solved block (hand-off) -> 32 FMAs per thread -> publish the slab's right-hand side -> gather the
//     block's right-hand side from the seven block mates (hand-off) -> 32 FMAs + an 8-lane reduction -> publish the solution:
//     two hand-offs per 256 rows instead of two per 128, with half the arithmetic behind each,
//   * accuracy is that of trsv_pair_kernel: a second sweep, one block behind, solves L d = e for the residual e = b1 - L_kk x0
//     of the first and x = x0 + d -- one step of fixed-precision iterative refinement of the whole triangular solve (Skeel
//     1980; Higham, Accuracy and Stability, Thm 12.3), so M2 only has to be a reasonable inverse.
//
// Hand-offs are the data-tagged granules of blas2.hip ({epoch, 32-bit half of a double}, one relaxed agent-scope store each, no
// fences).  Granule vectors: X0 | B1 | E | D | B2, two granules per entry.  Workgroup ids follow the dependency order and a
// workgroup only ever waits for ids below its own or for its block mates (at most 15 ids above it), so the launch makes progress
// under any residency of 16 workgroups or more; every spin is bounded and a timeout sets *err.  Deterministic: fixed summation
// order everywhere.
#include "kkt_common.h"

namespace mi355kkt {

typedef unsigned int u32;
typedef unsigned long long u64;
typedef double d4w __attribute__((ext_vector_type(4)));
#define RLX_AGENT_W __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define MFMA_F64_W(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

constexpr int WB = 256;      // order of a diagonal block / rows a hop advances
constexpr int WR = 32;       // rows of one workgroup
constexpr int WSL = WB / WR; // workgroups per block row
constexpr int PIL16 = 18;    // LDS leading dimension of the 128 x 16 intermediate of pair_inverse_kernel

// ---------------------------------------------------------------------------------------------------------------------
// M2 = inv(L[256 k .. 256 k + 255]^2) for every block k, column-major 256 x 256, followed by its transpose (the backward solve
// reads rows of M2'): out + k * 2 * 256 * 256.  minv: per 128-block M then M' (2 x 16384 doubles) from the tile Cholesky.
// Eight 256-thread workgroups per block, one per 16-column stripe of the lower-left quadrant: C1 = L_ba M_a (my columns) on the
// matrix cores with operands straight from memory (L2 holds them: the factorisation has just written them), through LDS,
// W = -M_b C1, then my columns of the four quadrants of M2 and M2' are written.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_inverse_kernel(const double* __restrict__ L, int64_t ldl, const double* __restrict__ minv,
                                                            double* __restrict__ out) {
    __shared__ double c1s[128 * PIL16];                    // C1[k][j] (j inside my stripe) at c1s[k * PIL16 + j]
    const int k = blockIdx.x, tj = blockIdx.y;             // block pair, 16-column stripe of the lower-left quadrant
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const double* Ma = minv + (int64_t)(2 * k) * (2 * 128 * 128);       // M_a, then M_a'
    const double* Mb = minv + (int64_t)(2 * k + 1) * (2 * 128 * 128);
    const double* Mat = Ma + 128 * 128;
    const double* Lba = L + (int64_t)(256 * k + 128) + (int64_t)(256 * k) * ldl;
    // ---- C1 = L_ba M_a, my 16 columns: tile ti of 16 x 16, D[i][j] = sum_k A[i][k] B[k][j]; lane (li, lq): A[i = li][4 s + lq],
    //      B[4 s + lq][j = li] = M_a'[j][k]; M_a is lower triangular: only k >= 16 tj contributes (its upper triangle is stored as
    //      zeros, so whole groups of eight steps are skipped, nothing is masked).  Two tiles per wave, all operands of a group of
    //      eight steps in flight together.
    for (int ti = wave; ti < 8; ti += 4) {
        d4w acc = d4w{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s8 = 0; s8 < 32; s8 += 8) {
            if (s8 + 8 <= 4 * tj) continue;
            double a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = Lba[16 * ti + li + (int64_t)(4 * (s8 + u) + lq) * ldl];
                b[u] = Mat[16 * tj + li + (4 * (s8 + u) + lq) * 128];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = MFMA_F64_W(a[u], b[u], acc);
        }
        // D layout: lane (li, lq) register r holds D[i = lq + 4 r][j = li]
#pragma unroll
        for (int r = 0; r < 4; ++r) c1s[(16 * ti + lq + 4 * r) * PIL16 + li] = acc[r];
    }
    __syncthreads();
    double* M2 = out + (int64_t)k * (2 * WB * WB);
    double* M2t = M2 + WB * WB;
    // ---- W = -M_b C1: A[i][k] = M_b[i][k] (zero for k > i: groups beyond the tile's rows are skipped), B[k][j] = C1[k][j]
    for (int ti = wave; ti < 8; ti += 4) {
        d4w acc = d4w{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s8 = 0; s8 < 32; s8 += 8) {
            if (s8 >= 4 * (ti + 1)) continue;
            double a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a[u] = Mb[16 * ti + li + (4 * (s8 + u) + lq) * 128];
                b[u] = c1s[(4 * (s8 + u) + lq) * PIL16 + li];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = MFMA_F64_W(a[u], b[u], acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * ti + lq + 4 * r, j = 16 * tj + li;
            const double v = -acc[r];
            M2[(128 + i) + (int64_t)j * WB] = v;
            M2t[j + (int64_t)(128 + i) * WB] = v;
        }
    }
    // ---- my 16 columns of the other three quadrants: M_a, M_b on the diagonal (their stored transposes for M2'), zeros elsewhere
    for (int e = tid; e < 128 * 16; e += 256) {
        const int i = e & 127, j = 16 * tj + (e >> 7);
        const int src = i + j * 128;
        M2[i + (int64_t)j * WB] = Ma[src];
        M2[(128 + i) + (int64_t)(128 + j) * WB] = Mb[src];
        M2[i + (int64_t)(128 + j) * WB] = 0.0;
        M2t[i + (int64_t)j * WB] = Mat[src];
        M2t[(128 + i) + (int64_t)(128 + j) * WB] = Mb[128 * 128 + src];
        M2t[(128 + i) + (int64_t)j * WB] = 0.0;
    }
}

int launch_pair_inverse(const double* L, int64_t ldl, int n, const double* minv, double* out, hipStream_t st) {
    if (n <= 0 || n % WB || !minv || !out) return -1;
    hipLaunchKernelGGL(pair_inverse_kernel, dim3(n / WB, 8), dim3(256), 0, st, L, ldl, minv, out);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

// sum over the 8 lanes that share a row (lanes 8 r .. 8 r + 7), result in all of them; fixed order
__device__ __forceinline__ double reduce8(double v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}

template <bool TRANS>
__global__ __launch_bounds__(256, 2) void trsv_wide_kernel(const double* __restrict__ L, int64_t ldl, int n, double* x, u32 epoch,
                                                           int* err, u64* gran, const double* __restrict__ minv2) {
    __shared__ double xs2[2][WB];
    __shared__ double vb[WB];
    __shared__ double mine[WR];
    const int tid = threadIdx.x;
    const int cg = tid & 7, r = tid >> 3;                   // column group (32 columns) and row of the slab
    const int nslab = n / WR, nblk = n / WB;
    const int role = (int)blockIdx.x & 1;                   // 0: sweep 1, 1: sweep 2
    const int pos = (int)blockIdx.x >> 1;
    const int s = TRANS ? nslab - 1 - pos : pos;            // my slab
    const int k = s / WSL;                                  // its block
    const int idx = WR * s + r;                             // my row (forward) / my column (backward)
    const int rin = idx - WB * k;                           // ... inside the block
    u64* gX0 = gran;
    u64* gB1 = gran + (int64_t)2 * n;
    u64* gE = gran + (int64_t)4 * n;
    u64* gD = gran + (int64_t)6 * n;
    u64* gB2 = gran + (int64_t)8 * n;
    const u64* gin = role == 0 ? gX0 : gD;                  // the solved blocks my far-field products consume
    u64* gbout = role == 0 ? gB1 : gB2;                     // my slab's right-hand side, for the block mates
    u64* gxout = role == 0 ? gX0 : gD;                      // my slab of the block's solution
    double acc = (role == 0 && cg == 0) ? x[idx] : 0.0;
    // my 32 columns of my row of M2 (forward) / M2' (backward): issued now, used at the end
    const double* Mk = minv2 + (int64_t)k * (2 * WB * WB) + (TRANS ? WB * WB : 0);
    double ra[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) ra[c] = Mk[rin + (int64_t)(32 * cg + c) * WB];
    // all 256 threads: the 512 granules of block j of a granule vector -> 256 doubles at dst; false on a timeout (err is set).
    // dist: blocks between the producer and the front of my own work -- far behind the front the poll is slow (the data is needed
    // many hops from now; 500 workgroups polling one block at full rate would be a terabyte per second of traffic)
    auto wait_block = [&](const u64* gbase, int j, double* dst, int dist) -> bool {
        const u64* g = gbase + (int64_t)j * (2 * WB) + tid;
        u64 v0 = 0, v1 = 0;
        bool got0 = false, got1 = false;
        for (unsigned spins = 0; spins < (1u << 21); ++spins) {
            if (!got0) {
                v0 = __hip_atomic_load(g, RLX_AGENT_W);
                got0 = (u32)(v0 >> 32) == epoch;
            }
            if (!got1) {
                v1 = __hip_atomic_load(g + 256, RLX_AGENT_W);
                got1 = (u32)(v1 >> 32) == epoch;
            }
            if (got0 && got1) break;
            if (dist > 8) __builtin_amdgcn_s_sleep(127);         // (~4 us: far behind the front)
            else if (dist > 4) __builtin_amdgcn_s_sleep(64);
            else if (dist > 1) __builtin_amdgcn_s_sleep(16);
            else __builtin_amdgcn_s_sleep(1);
        }
        reinterpret_cast<u32*>(dst)[tid] = (u32)v0;          // little endian: granule 2 i / 2 i + 1 = low / high word of entry i
        reinterpret_cast<u32*>(dst)[tid + 256] = (u32)v1;
        if (__syncthreads_or((got0 && got1) ? 0 : 1)) {
            if (tid == 0) atomicExch(err, 1);
            return false;
        }
        return true;
    };
    // threads 0..63: the 64 granules of my own slab of a granule vector -> mine[0..31]
    auto wait_mine = [&](const u64* gbase) -> bool {
        bool got = true;
        if (tid < 2 * WR) {
            const u64* g = gbase + (int64_t)2 * WR * s + tid;
            u64 v = 0;
            got = false;
            for (unsigned spins = 0; spins < (1u << 21); ++spins) {
                v = __hip_atomic_load(g, RLX_AGENT_W);
                if ((u32)(v >> 32) == epoch) { got = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            reinterpret_cast<u32*>(mine)[tid] = (u32)v;
        }
        if (__syncthreads_or(got ? 0 : 1)) {
            if (tid == 0) atomicExch(err, 1);
            return false;
        }
        return true;
    };
    auto publish = [&](u64* gbase, double v) {                // lanes cg == 0 / 1: low / high word of entry idx
        if (cg < 2) {
            const u32 w = cg == 0 ? (u32)__double2loint(v) : (u32)__double2hiint(v);
            __hip_atomic_store(gbase + (int64_t)2 * idx + cg, ((u64)epoch << 32) | w, RLX_AGENT_W);
        }
    };
    // 32-term dot product of a register strip with my 32 entries of an LDS vector, as four independent chains
    auto dot32 = [&](const double (&a)[32], const double* v) {
        double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
#pragma unroll
        for (int c = 0; c < 32; c += 4) {
            d0 = fma(a[c], v[32 * cg + c], d0);
            d1 = fma(a[c + 1], v[32 * cg + c + 1], d1);
            d2 = fma(a[c + 2], v[32 * cg + c + 2], d2);
            d3 = fma(a[c + 3], v[32 * cg + c + 3], d3);
        }
        return (d0 + d1) + (d2 + d3);
    };
    // ---- far field: the solved blocks in dependency order
    const int nsteps = TRANS ? (nblk - 1 - k) : k;
    for (int st = 0; st < nsteps; ++st) {
        const int j = TRANS ? (nblk - 1 - st) : st;
        double l0[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) l0[c] = L[idx + (int64_t)(WB * j + 32 * cg + c) * ldl];     // (backward: the mirrored L')
        double* xs = xs2[st & 1];
        if (!wait_block(gin, j, xs, nsteps - st)) return;
        acc -= dot32(l0, xs);
    }
    double* xs = xs2[nsteps & 1];
    const double far = reduce8(acc);                          // rhs_r - sum_j L_rj x_j, in all eight lanes of the row
    if (role == 0) {
        // sweep 1: b1 -> block mates; x0 = M2 b1 (my rows); e = b1 - L_kk x0 (my rows) -> sweep 2
        publish(gbout, far);
        if (!wait_block(gbout, k, xs, 0)) return;
        const double x0 = reduce8(dot32(ra, xs));
        publish(gxout, x0);
        // (off the chain from here) my row of the diagonal block of L itself, then the block's x0
        double rb[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const int col = WB * k + 32 * cg + c;
            const bool in = TRANS ? col >= idx : col <= idx;  // (the other triangle holds the mirrored copy)
            rb[c] = in ? L[idx + (int64_t)col * ldl] : 0.0;
        }
        if (!wait_block(gX0, k, vb, 0)) return;
        const double q = reduce8(dot32(rb, vb));
        publish(gE, far - q);
        return;
    }
    // sweep 2: b2 = e + far (my rows) -> block mates; d = M2 b2; x = x0 + d
    if (!wait_mine(gE)) return;
    publish(gbout, far + mine[r]);
    if (!wait_block(gbout, k, xs, 0)) return;
    const double dk = reduce8(dot32(ra, xs));
    publish(gxout, dk);
    if (!wait_mine(gX0)) return;                              // (published by sweep 1 before e: there since long)
    if (cg == 0) x[idx] = mine[r] + dk;
}

// x := L^-1 x (trans = 0) or L^-T x (trans = 1; needs the mirrored upper triangle): n a multiple of 256, minv2 from
// launch_pair_inverse, gran with room for 10 n granules (zeroed once, epochs never repeat)
int launch_trsv_wide(const double* L, int64_t ldl, int n, double* x, int trans, unsigned int epoch, int* err, hipStream_t st,
                     unsigned long long* gran, const double* minv2) {
    if (n <= 0 || n % WB || !gran || !minv2) return -1;
    const dim3 g(2 * (n / WR)), b(256);
    if (trans)
        hipLaunchKernelGGL((trsv_wide_kernel<true>), g, b, 0, st, L, ldl, n, x, epoch, err, gran, minv2);
    else
        hipLaunchKernelGGL((trsv_wide_kernel<false>), g, b, 0, st, L, ldl, n, x, epoch, err, gran, minv2);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace mi355kkt

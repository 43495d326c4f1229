// Nesterov-Todd scaling W^-T applied to a block of columns, on the device (HBM-bound):
// the replacement of misc_solvers.scale(x, W, trans='T', inverse='I') (reference
// src/C/misc_solvers.c:85-244) for the 'l' and 'q' cones, as used by the kktsolvers
// (misc.py:1093, :1116, :1271, :1306, :1513).
//   'l' rows:   y = di .* x                                        (misc_solvers.c:132-141)
//   'q' cone k: y = (1/beta) (2 Jv (Jv)'x - J x),  J = diag(1,-1,...,-1)   (misc_solvers.c:144-183)
// W_k^-1 is symmetric, so the same kernel serves trans = 'N' and 'T'.
// Where the reference makes one dgemv + one dger + ncols dscal calls PER CONE (2 M BLAS-1 calls for
// 1024 cones x 2048 columns), this is one launch: each (cone, column) pair is one coalesced read and
// one coalesced write of the cone's rows.
#include "kkt_common.h"

namespace mi355kkt {

// out[:, j] = W^-T in[:, j] for the l rows: plain diagonal scaling
__global__ __launch_bounds__(256) void scale_l_kernel(const double* __restrict__ in, int64_t ldi,
                                                      double* __restrict__ out, int64_t ldo, int ml, int ncols,
                                                      const double* __restrict__ di, double extra) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (i < ml && j < ncols) out[i + (int64_t)j * ldo] = extra * di[i] * in[i + (int64_t)j * ldi];
}

// small cones (dimension <= 32): one thread per (cone, column); cone rows are contiguous in the column.
// grid: (ceil(ncones/64), ncols); cone descriptors: off[k] (row offset), dim[k], voff[k] (offset into v)
__global__ __launch_bounds__(64) void scale_q_small_kernel(const double* __restrict__ in, int64_t ldi,
                                                           double* __restrict__ out, int64_t ldo, int ncones,
                                                           const int* __restrict__ off, const int* __restrict__ dim,
                                                           const int* __restrict__ voff, const double* __restrict__ v,
                                                           const double* __restrict__ beta, int ncols, double extra) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y;
    if (k >= ncones || j >= ncols) return;
    const int m = dim[k];
    const double* __restrict__ x = in + off[k] + (int64_t)j * ldi;
    double* __restrict__ y = out + off[k] + (int64_t)j * ldo;
    const double* __restrict__ vk = v + voff[k];
    // cones of dimension 8 or 4 on 16-byte boundaries (the SOCP classes of BASELINE configs[2]): 16-byte requests, no
    // per-row predicates.  (Staging the rows through LDS with fully coalesced 512-byte requests was measured SLOWER: one wave
    // per block and two barriers per column hide less latency than this kernel's many independent threads.)
    if ((m == 8 || m == 4) && ((off[k] | voff[k]) & 1) == 0 && ((ldi | ldo) & 1) == 0 &&
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        typedef double dv2 __attribute__((ext_vector_type(2)));
        const dv2* __restrict__ x2 = reinterpret_cast<const dv2*>(x);
        const dv2* __restrict__ v2 = reinterpret_cast<const dv2*>(vk);
        dv2* __restrict__ y2 = reinterpret_cast<dv2*>(y);
        const double s = extra / beta[k];
        if (m == 8) {
            const dv2 a0 = x2[0], a1 = x2[1], a2 = x2[2], a3 = x2[3];
            const dv2 b0 = v2[0], b1 = v2[1], b2 = v2[2], b3 = v2[3];
            const double w = b0.x * a0.x - b0.y * a0.y - (b1.x * a1.x + b1.y * a1.y) - (b2.x * a2.x + b2.y * a2.y) -
                             (b3.x * a3.x + b3.y * a3.y);
            y2[0] = dv2{s * (2.0 * b0.x * w - a0.x), s * (-2.0 * b0.y * w + a0.y)};
            y2[1] = dv2{s * (-2.0 * b1.x * w + a1.x), s * (-2.0 * b1.y * w + a1.y)};
            y2[2] = dv2{s * (-2.0 * b2.x * w + a2.x), s * (-2.0 * b2.y * w + a2.y)};
            y2[3] = dv2{s * (-2.0 * b3.x * w + a3.x), s * (-2.0 * b3.y * w + a3.y)};
        } else {
            const dv2 a0 = x2[0], a1 = x2[1];
            const dv2 b0 = v2[0], b1 = v2[1];
            const double w = b0.x * a0.x - b0.y * a0.y - (b1.x * a1.x + b1.y * a1.y);
            y2[0] = dv2{s * (2.0 * b0.x * w - a0.x), s * (-2.0 * b0.y * w + a0.y)};
            y2[1] = dv2{s * (-2.0 * b1.x * w + a1.x), s * (-2.0 * b1.y * w + a1.y)};
        }
        return;
    }
    double xv[32];
    double w = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (i < m) {
            xv[i] = x[i];
            w += (i == 0 ? vk[0] : -vk[i]) * xv[i];       // w = (Jv)' x
        }
    }
    const double s = extra / beta[k];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (i < m) {
            const double jv = (i == 0 ? vk[0] : -vk[i]);
            const double jx = (i == 0 ? xv[0] : -xv[i]);
            y[i] = s * (2.0 * jv * w - jx);
        }
    }
}

// large cones: one wave per (cone, SQL_COLS columns), lanes stride the cone's rows; the loads of all its columns are in flight
// before the first butterfly (one column per wave left the kernel latency-bound at 2.2 TB/s)
constexpr int SQL_COLS = 4;
__global__ __launch_bounds__(256) void scale_q_large_kernel(const double* __restrict__ in, int64_t ldi,
                                                            double* __restrict__ out, int64_t ldo, int ncones,
                                                            const int* __restrict__ cone_ids,
                                                            const int* __restrict__ off, const int* __restrict__ dim,
                                                            const int* __restrict__ voff, const double* __restrict__ v,
                                                            const double* __restrict__ beta, int ncols, double extra) {
    const int lane = threadIdx.x & 63;
    const int kk = blockIdx.x;
    const int j0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * SQL_COLS;
    if (kk >= ncones || j0 >= ncols) return;
    const int k = cone_ids[kk];
    const int m = dim[k];
    const double* __restrict__ vk = v + voff[k];
    const double s = extra / beta[k];
    const int nc = min(SQL_COLS, ncols - j0);
    if (m <= 128 && nc == SQL_COLS) {                    // up to two rows per lane: everything in registers
        const int i0 = lane, i1 = lane + 64;
        const double jv0 = (i0 < m) ? (i0 == 0 ? vk[0] : -vk[i0]) : 0.0;
        const double jv1 = (i1 < m) ? -vk[i1] : 0.0;
        double xa[SQL_COLS], xb[SQL_COLS], w[SQL_COLS];
#pragma unroll
        for (int c = 0; c < SQL_COLS; ++c) {
            const double* __restrict__ x = in + off[k] + (int64_t)(j0 + c) * ldi;
            xa[c] = (i0 < m) ? x[i0] : 0.0;
            xb[c] = (i1 < m) ? x[i1] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < SQL_COLS; ++c) w[c] = jv0 * xa[c] + jv1 * xb[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int c = 0; c < SQL_COLS; ++c) w[c] += __shfl_xor(w[c], o, 64);
#pragma unroll
        for (int c = 0; c < SQL_COLS; ++c) {
            double* __restrict__ y = out + off[k] + (int64_t)(j0 + c) * ldo;
            if (i0 < m) y[i0] = s * (2.0 * jv0 * w[c] - (i0 == 0 ? xa[c] : -xa[c]));
            if (i1 < m) y[i1] = s * (2.0 * jv1 * w[c] + xb[c]);
        }
        return;
    }
    for (int c = 0; c < nc; ++c) {
        const double* __restrict__ x = in + off[k] + (int64_t)(j0 + c) * ldi;
        double* __restrict__ y = out + off[k] + (int64_t)(j0 + c) * ldo;
        double w = 0.0;
        for (int i = lane; i < m; i += 64) w += (i == 0 ? vk[0] : -vk[i]) * x[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
        for (int i = lane; i < m; i += 64) {
            const double jv = (i == 0 ? vk[0] : -vk[i]);
            const double jx = (i == 0 ? x[0] : -x[i]);
            y[i] = s * (2.0 * jv * w - jx);
        }
    }
}

// ---- 's' (semidefinite) blocks: x_k := vec(rti' mat(x_k) rti) (misc_solvers.c:187-240, trans='T', inverse='I'), written
//      straight into packed-lower storage with off-diagonals scaled by sqrt(2) (misc_solvers.c:412-550, pack/pack2),
//      which is the row space the SYRK and the GEMVs work in.  One workgroup per (block, column).
constexpr int SDP_MAXN = 80;
__global__ __launch_bounds__(256) void sdp_scale_pack_kernel(const double* __restrict__ in, int64_t ldi,
                                                             double* __restrict__ out, int64_t ldo, const int* __restrict__ sdim,
                                                             const int* __restrict__ soff, const int* __restrict__ spoff,
                                                             const int* __restrict__ sroff, const double* __restrict__ rti,
                                                             double extra) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int k = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    const int nk = sdim[k];
    if (nk > SDP_MAXN) return;             // large blocks of a mixed problem go through launch_sdp_congruence
    double* X = sm;
    double* T = sm + nk * nk;
    const double* __restrict__ x = in + soff[k] + (int64_t)j * ldi;
    const double* __restrict__ Rm = rti + sroff[k];
    for (int e = tid; e < nk * nk; e += 256) {
        const int a = e % nk, b = e / nk;
        if (a >= b) {                      // only the lower triangle of mat(x_k) is referenced
            const double v = x[e];
            X[a * nk + b] = v;
            X[b * nk + a] = v;
        }
    }
    __syncthreads();
    for (int e = tid; e < nk * nk; e += 256) {      // T = X R
        const int a = e % nk, b = e / nk;
        double s = 0.0;
        for (int c = 0; c < nk; ++c) s += X[a * nk + c] * Rm[c + b * nk];
        T[a + b * nk] = s;
    }
    __syncthreads();
    double* __restrict__ y = out + spoff[k] + (int64_t)j * ldo;
    const double r2 = 1.4142135623730951;
    for (int e = tid; e < nk * nk; e += 256) {      // Y = R' T, lower triangle, packed
        const int a = e % nk, b = e / nk;
        if (a < b) continue;
        double s = 0.0;
        for (int c = 0; c < nk; ++c) s += Rm[c + a * nk] * T[c + b * nk];
        const int idx = b * nk - (b * (b - 1)) / 2 + (a - b);
        y[idx] = extra * ((a == b) ? s : r2 * s);
    }
}

// Blocks too large for the LDS-resident kernel above: Y = R' X R in 16-column panels of R, T = X R[:, panel] staged in LDS
// (nk x 16), X read from global memory through its lower triangle.  Plain FP64 FMAs: a functional path for the occasional
// large block (mcsdp-style problems), not a tuned one.
constexpr int SDP_PANEL = 16;
__global__ __launch_bounds__(256) void sdp_scale_pack_big_kernel(const double* __restrict__ in, int64_t ldi,
                                                                 double* __restrict__ out, int64_t ldo,
                                                                 const int* __restrict__ sdim, const int* __restrict__ soff,
                                                                 const int* __restrict__ spoff, const int* __restrict__ sroff,
                                                                 const double* __restrict__ rti, double extra) {
    extern __shared__ __attribute__((aligned(16))) double sm[];     // T: nk x SDP_PANEL
    const int k = blockIdx.x, j = blockIdx.y, tid = threadIdx.x;
    const int nk = sdim[k];
    const double* __restrict__ x = in + soff[k] + (int64_t)j * ldi;
    const double* __restrict__ Rm = rti + sroff[k];
    double* __restrict__ y = out + spoff[k] + (int64_t)j * ldo;
    const double r2 = 1.4142135623730951;
    // blockIdx.z strides over the panels: a single vector (the right-hand side of a solve) still spreads over nk / 16 CUs
    for (int b0 = blockIdx.z * SDP_PANEL; b0 < nk; b0 += gridDim.z * SDP_PANEL) {
        const int pw = min(SDP_PANEL, nk - b0);
        for (int e = tid; e < nk * pw; e += 256) {      // T[a][bb] = sum_c X[a][c] R[c][b0 + bb], X symmetric from its lower part
            const int a = e % nk, bb = e / nk;
            const double* __restrict__ rc = Rm + (int64_t)(b0 + bb) * nk;
            double s = 0.0;
            for (int c = 0; c < a; ++c) s += x[a + (int64_t)c * nk] * rc[c];
            for (int c = a; c < nk; ++c) s += x[c + (int64_t)a * nk] * rc[c];
            sm[a + bb * nk] = s;
        }
        __syncthreads();
        for (int e = tid; e < nk * pw; e += 256) {      // Y[a][b0 + bb] = sum_c R[c][a] T[c][bb], lower triangle, packed
            const int a = e % nk, bb = e / nk, b = b0 + bb;
            if (a < b) continue;
            const double* __restrict__ ra = Rm + (int64_t)a * nk;
            double s = 0.0;
            for (int c = 0; c < nk; ++c) s += ra[c] * sm[c + bb * nk];
            const int64_t idx = (int64_t)b * nk - ((int64_t)b * (b - 1)) / 2 + (a - b);
            y[idx] = extra * ((a == b) ? s : r2 * s);
        }
        __syncthreads();
    }
}

// packed -> unpacked for the 's' part of a single vector (misc_solvers.c:552-601): lower triangle only
__global__ __launch_bounds__(256) void sdp_unpack_kernel(const double* __restrict__ packed, double* __restrict__ out,
                                                         const int* __restrict__ sdim, const int* __restrict__ soff,
                                                         const int* __restrict__ spoff) {
    const int k = blockIdx.x, nk = sdim[k];
    const double ir2 = 0.70710678118654752;
    for (int e = threadIdx.x; e < nk * nk; e += 256) {
        const int a = e % nk, b = e / nk;
        if (a < b) continue;
        const int idx = b * nk - (b * (b - 1)) / 2 + (a - b);
        const double v = packed[spoff[k] + idx];
        out[soff[k] + e] = (a == b) ? v : ir2 * v;
    }
}

int cone_layout_build_s(ConeLayout& cl, int lq_rows, const std::vector<int>& s) {
    cl.ns = (int)s.size();
    cl.lq_rows = lq_rows;
    cl.cdim_packed = lq_rows;
    cl.rlen = 0;
    cl.s_maxn = 0;
    if (cl.ns == 0) return 0;
    std::vector<int> sdim(s), soff(cl.ns), spoff(cl.ns), sroff(cl.ns);
    int o = lq_rows, po = lq_rows, ro = 0;
    for (int k = 0; k < cl.ns; ++k) {
        soff[k] = o;
        spoff[k] = po;
        sroff[k] = ro;
        o += s[k] * s[k];
        po += s[k] * (s[k] + 1) / 2;
        ro += s[k] * s[k];
        cl.s_maxn = std::max(cl.s_maxn, s[k]);
    }
    cl.cdim_packed = po;
    cl.rlen = ro;
    auto up = [&](int** d, const std::vector<int>& h) -> int {
        KKT_HIP_CHECK(DEV_ALLOC(d, sizeof(int) * (h.size() ? h.size() : 1)));
        if (!h.empty()) KKT_HIP_CHECK(memcpy_sync(*d, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
        return 0;
    };
    cl.h_sdim = sdim; cl.h_soff = soff; cl.h_spoff = spoff; cl.h_sroff = sroff;
    if (int e = up(&cl.d_sdim, sdim)) return e;
    if (int e = up(&cl.d_soff, soff)) return e;
    if (int e = up(&cl.d_spoff, spoff)) return e;
    if (int e = up(&cl.d_sroff, sroff)) return e;
    return 0;
}

int launch_sdp_scale_pack(const ConeLayout& cl, const double* in, int64_t ldi, double* out, int64_t ldo, int ncols,
                          const double* d_rti, double extra, hipStream_t st) {
    if (cl.ns == 0 || ncols <= 0) return 0;
    const bool no_mfma = dev_knob("MI355KKT_SDP_NO_MFMA") != nullptr;
    if (cl.s_maxn > SDP_MAXN && ncols >= 16 && !no_mfma) {
        // a whole matrix of columns (Gs = W^-T G): blocks > 80 as two batched FP64-MFMA products each (gemm_f64.hip),
        // the smaller ones through the LDS-resident kernel
        int small_max = 0;
        size_t need = 0;
        for (int k = 0; k < cl.ns; ++k) {
            const int nk = cl.h_sdim[k];
            if (nk <= SDP_MAXN) small_max = std::max(small_max, nk);
            else need = std::max(need, (size_t)nk * nk * (1 + (size_t)std::min(ncols, std::max(1, (int)((size_t)(32 << 20) / ((size_t)nk * nk))))));
        }
        if (need > cl.cong_doubles) {
            if (cl.d_cong) (void)dev_free(cl.d_cong);
            cl.d_cong = nullptr;
            cl.cong_doubles = 0;
            KKT_HIP_CHECK(DEV_ALLOC(&cl.d_cong, sizeof(double) * need));
            cl.cong_doubles = need;
        }
        if (small_max > 0) {
            static bool attr_s = false;
            if (!attr_s) {
                KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sdp_scale_pack_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)(sizeof(double) * 2 * SDP_MAXN * SDP_MAXN)));
                attr_s = true;
            }
            hipLaunchKernelGGL(sdp_scale_pack_kernel, dim3(cl.ns, ncols), dim3(256), sizeof(double) * 2 * small_max * small_max, st,
                               in, ldi, out, ldo, cl.d_sdim, cl.d_soff, cl.d_spoff, cl.d_sroff, d_rti, extra);
        }
        for (int k = 0; k < cl.ns; ++k) {
            const int nk = cl.h_sdim[k];
            if (nk <= SDP_MAXN) continue;
            if (int e = launch_sdp_congruence(d_rti + cl.h_sroff[k], nk, in + cl.h_soff[k], ldi, out + cl.h_spoff[k], ldo, ncols, extra,
                                              cl.d_cong, cl.cong_doubles, st))
                return e;
        }
        KKT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (cl.s_maxn > SDP_MAXN) {
        if ((size_t)cl.s_maxn * SDP_PANEL * sizeof(double) > 150 * 1024) {
            set_last_error("semidefinite blocks larger than %d x %d are not supported on the device", 150 * 1024 / 8 / SDP_PANEL,
                           150 * 1024 / 8 / SDP_PANEL);
            return -4;
        }
        static bool attr_big = false;
        if (!attr_big) {
            KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sdp_scale_pack_big_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            attr_big = true;
        }
        const int npanels = (cl.s_maxn + SDP_PANEL - 1) / SDP_PANEL;
        const int gz = std::max(1, std::min(npanels, 1024 / std::max(1, cl.ns * ncols)));
        hipLaunchKernelGGL(sdp_scale_pack_big_kernel, dim3(cl.ns, ncols, gz), dim3(256), sizeof(double) * cl.s_maxn * SDP_PANEL, st, in,
                           ldi, out, ldo, cl.d_sdim, cl.d_soff, cl.d_spoff, cl.d_sroff, d_rti, extra);
        KKT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    static bool attr = false;
    const size_t lds = sizeof(double) * 2 * SDP_MAXN * SDP_MAXN;
    if (!attr) {
        KKT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sdp_scale_pack_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL(sdp_scale_pack_kernel, dim3(cl.ns, ncols), dim3(256), sizeof(double) * 2 * cl.s_maxn * cl.s_maxn, st, in,
                       ldi, out, ldo, cl.d_sdim, cl.d_soff, cl.d_spoff, cl.d_sroff, d_rti, extra);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_sdp_unpack(const ConeLayout& cl, const double* packed, double* out, hipStream_t st) {
    if (cl.ns == 0) return 0;
    hipLaunchKernelGGL(sdp_unpack_kernel, dim3(cl.ns), dim3(256), 0, st, packed, out, cl.d_sdim, cl.d_soff, cl.d_spoff);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

int cone_layout_build(ConeLayout& cl, int ml, const std::vector<int>& q) {
    cl.ml = ml;
    cl.nq = (int)q.size();
    if (cl.nq == 0) return 0;
    std::vector<int> off(cl.nq), dim(q), voff(cl.nq), small_ids, large_ids;
    int o = ml, vo = 0;
    for (int k = 0; k < cl.nq; ++k) {
        off[k] = o;
        voff[k] = vo;
        o += q[k];
        vo += q[k];
        (q[k] <= 32 ? small_ids : large_ids).push_back(k);
    }
    cl.vlen = vo;
    // the small-cone kernel indexes cones densely: build a compacted descriptor set for it
    std::vector<int> s_off, s_dim, s_voff, s_beta_idx;
    for (int k : small_ids) {
        s_off.push_back(off[k]);
        s_dim.push_back(dim[k]);
        s_voff.push_back(voff[k]);
    }
    cl.n_small = (int)small_ids.size();
    cl.n_large = (int)large_ids.size();
    auto up = [&](int** d, const std::vector<int>& h) -> int {
        KKT_HIP_CHECK(DEV_ALLOC(d, sizeof(int) * (h.size() ? h.size() : 1)));
        if (!h.empty()) KKT_HIP_CHECK(memcpy_sync(*d, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
        return 0;
    };
    if (int e = up(&cl.d_off, off)) return e;
    if (int e = up(&cl.d_dim, dim)) return e;
    if (int e = up(&cl.d_voff, voff)) return e;
    if (int e = up(&cl.d_small_ids, small_ids)) return e;
    if (int e = up(&cl.d_large_ids, large_ids)) return e;
    if (int e = up(&cl.d_s_off, s_off)) return e;
    if (int e = up(&cl.d_s_dim, s_dim)) return e;
    if (int e = up(&cl.d_s_voff, s_voff)) return e;
    // beta for the compacted small set is gathered on the device at factor time (beta changes per factor)
    KKT_HIP_CHECK(DEV_ALLOC(&cl.d_s_beta, sizeof(double) * (cl.n_small ? cl.n_small : 1)));
    return 0;
}

void cone_layout_free(ConeLayout& cl) {
    int* ip[] = {cl.d_off, cl.d_dim, cl.d_voff, cl.d_small_ids, cl.d_large_ids, cl.d_s_off, cl.d_s_dim, cl.d_s_voff};
    for (int* p : ip)
        if (p) (void)dev_free(p);
    if (cl.d_s_beta) (void)dev_free(cl.d_s_beta);
    int* sp[] = {cl.d_sdim, cl.d_soff, cl.d_spoff, cl.d_sroff};
    for (int* p : sp)
        if (p) (void)dev_free(p);
    if (cl.d_cong) (void)dev_free(cl.d_cong);
    cl = ConeLayout();
}

__global__ void gather_beta_kernel(const double* __restrict__ beta, const int* __restrict__ ids, double* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = beta[ids[i]];
}

int cone_layout_set_beta(ConeLayout& cl, const double* d_beta, hipStream_t st) {
    if (cl.n_small > 0) {
        hipLaunchKernelGGL(gather_beta_kernel, dim3((cl.n_small + 255) / 256), dim3(256), 0, st, d_beta, cl.d_small_ids,
                           cl.d_s_beta, cl.n_small);
        KKT_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

// out(:, 0:ncols) = extra * W^-T in(:, 0:ncols)   for the l + q rows (in and out may alias)
int launch_cone_scale(const ConeLayout& cl, const double* in, int64_t ldi, double* out, int64_t ldo, int ncols,
                      const double* d_di, const double* d_v, const double* d_beta, double extra, hipStream_t st) {
    if (ncols <= 0) return 0;
    if (cl.ml > 0) {
        hipLaunchKernelGGL(scale_l_kernel, dim3((cl.ml + 255) / 256, ncols), dim3(256), 0, st, in, ldi, out, ldo, cl.ml,
                           ncols, d_di, extra);
        KKT_HIP_CHECK(hipGetLastError());
    }
    if (cl.n_small > 0) {
        hipLaunchKernelGGL(scale_q_small_kernel, dim3((cl.n_small + 63) / 64, ncols), dim3(64), 0, st, in, ldi, out, ldo,
                           cl.n_small, cl.d_s_off, cl.d_s_dim, cl.d_s_voff, d_v, cl.d_s_beta, ncols, extra);
        KKT_HIP_CHECK(hipGetLastError());
    }
    if (cl.n_large > 0) {
        hipLaunchKernelGGL(scale_q_large_kernel, dim3(cl.n_large, (ncols + 4 * SQL_COLS - 1) / (4 * SQL_COLS)), dim3(256), 0, st, in, ldi, out, ldo,
                           cl.n_large, cl.d_large_ids, cl.d_off, cl.d_dim, cl.d_voff, d_v, d_beta, ncols, extra);
        KKT_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

// ---- batched problems (mi355kkt_batch_* with second-order cones): every problem has its own (di, v, beta) ----------------
// cones of dimension <= 32: one thread per (cone, column, problem), the cone's rows in registers
__global__ __launch_bounds__(64) void batch_scale_q_small_kernel(const double* in, int64_t ldi, int64_t sIn,
                                                                 double* out, int64_t ldo, int64_t sOut, int nq,
                                                                 const int* __restrict__ qoff, const int* __restrict__ qdim,
                                                                 int ml, int vstride, int bstride,
                                                                 const double* __restrict__ v, const double* __restrict__ beta,
                                                                 int ncols) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    const int j = blockIdx.y;
    const int64_t b = blockIdx.z;
    if (k >= nq || j >= ncols) return;
    const int m = qdim[k];
    if (m > 32) return;                                  // (batch_scale_wave_kernel)
    const double* x = in + b * sIn + qoff[k] + (int64_t)j * ldi;
    double* y = out + b * sOut + qoff[k] + (int64_t)j * ldo;
    const double* __restrict__ vk = v + b * vstride + (qoff[k] - ml);
    // cones of 8 or 4 rows on 16-byte boundaries: 16-byte requests like scale_q_small_kernel (1.7 -> TB/s class of the
    // single-problem kernel); in == out is fine, every thread reads its cone before it writes it
    if ((m == 8 || m == 4) && ((qoff[k] | (qoff[k] - ml) | vstride) & 1) == 0 && ((ldi | ldo | sIn | sOut) & 1) == 0 &&
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        typedef double dv2 __attribute__((ext_vector_type(2)));
        const dv2* x2 = reinterpret_cast<const dv2*>(x);
        const dv2* __restrict__ v2 = reinterpret_cast<const dv2*>(vk);
        dv2* y2 = reinterpret_cast<dv2*>(y);
        const double s = 1.0 / beta[b * bstride + k];
        if (m == 8) {
            const dv2 a0 = x2[0], a1 = x2[1], a2 = x2[2], a3 = x2[3];
            const dv2 b0 = v2[0], b1 = v2[1], b2 = v2[2], b3 = v2[3];
            const double w = b0.x * a0.x - b0.y * a0.y - (b1.x * a1.x + b1.y * a1.y) - (b2.x * a2.x + b2.y * a2.y) -
                             (b3.x * a3.x + b3.y * a3.y);
            y2[0] = dv2{s * (2.0 * b0.x * w - a0.x), s * (-2.0 * b0.y * w + a0.y)};
            y2[1] = dv2{s * (-2.0 * b1.x * w + a1.x), s * (-2.0 * b1.y * w + a1.y)};
            y2[2] = dv2{s * (-2.0 * b2.x * w + a2.x), s * (-2.0 * b2.y * w + a2.y)};
            y2[3] = dv2{s * (-2.0 * b3.x * w + a3.x), s * (-2.0 * b3.y * w + a3.y)};
        } else {
            const dv2 a0 = x2[0], a1 = x2[1];
            const dv2 b0 = v2[0], b1 = v2[1];
            const double w = b0.x * a0.x - b0.y * a0.y - (b1.x * a1.x + b1.y * a1.y);
            y2[0] = dv2{s * (2.0 * b0.x * w - a0.x), s * (-2.0 * b0.y * w + a0.y)};
            y2[1] = dv2{s * (-2.0 * b1.x * w + a1.x), s * (-2.0 * b1.y * w + a1.y)};
        }
        return;
    }
    double xv[32];
    double w = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (i < m) {
            xv[i] = x[i];
            w += (i == 0 ? vk[0] : -vk[i]) * xv[i];       // w = (Jv)' x
        }
    }
    const double s = 1.0 / beta[b * bstride + k];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        if (i < m) {
            const double jv = (i == 0 ? vk[0] : -vk[i]);
            const double jx = (i == 0 ? xv[0] : -xv[i]);
            y[i] = s * (2.0 * jv * w - jx);
        }
    }
}
// one wave per (unit, column, problem): units 0 .. nlarge-1 are the cones of dimension > 32 (large_ids; lanes stride the
// rows), the units after them 64-row pieces of the 'l' block
__global__ __launch_bounds__(256) void batch_scale_wave_kernel(const double* in, int64_t ldi, int64_t sIn,
                                                              double* out, int64_t ldo, int64_t sOut, int nlarge,
                                                              const int* __restrict__ large_ids,
                                                              const int* __restrict__ qoff, const int* __restrict__ qdim,
                                                              int ml, int cdim, int vstride, int bstride,
                                                              const double* __restrict__ di, const double* __restrict__ v,
                                                              const double* __restrict__ beta, int ncols) {
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x;
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int64_t b = blockIdx.z;
    if (j >= ncols) return;
    const double* x = in + b * sIn + (int64_t)j * ldi;
    double* y = out + b * sOut + (int64_t)j * ldo;
    if (u >= nlarge) {
        const int i = (u - nlarge) * 64 + lane;
        if (i < ml) y[i] = di[b * cdim + i] * x[i];
        return;
    }
    const int k = large_ids[u];
    const int m = qdim[k];
    x += qoff[k];
    y += qoff[k];
    const double* __restrict__ vk = v + b * vstride + (qoff[k] - ml);
    double w = 0.0;
    for (int i = lane; i < m; i += 64) w += (i == 0 ? vk[0] : -vk[i]) * x[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
    const double s = 1.0 / beta[b * bstride + k];
    for (int i = lane; i < m; i += 64) {
        const double jv = (i == 0 ? vk[0] : -vk[i]);
        const double jx = (i == 0 ? x[0] : -x[i]);
        y[i] = s * (2.0 * jv * w - jx);
    }
}

int launch_batch_cone_scale(const double* in, int64_t ldi, int64_t sIn, double* out, int64_t ldo, int64_t sOut, int ncols,
                            int nbatch, int cdim, int ml, int nq, int sumq, const int* d_qoff, const int* d_qdim,
                            const int* d_large_ids, int nlarge, const double* d_di, const double* d_v, const double* d_beta,
                            hipStream_t st) {
    if (ncols <= 0 || nbatch <= 0 || cdim <= 0) return 0;
    const int vstride = sumq > 0 ? sumq : 1, bstride = nq > 0 ? nq : 1;
    if (nq > nlarge) {
        hipLaunchKernelGGL(batch_scale_q_small_kernel, dim3((nq + 63) / 64, ncols, nbatch), dim3(64), 0, st, in, ldi, sIn, out, ldo,
                           sOut, nq, d_qoff, d_qdim, ml, vstride, bstride, d_v, d_beta, ncols);
        KKT_HIP_CHECK(hipGetLastError());
    }
    const int units = nlarge + (ml + 63) / 64;
    if (units > 0) {
        hipLaunchKernelGGL(batch_scale_wave_kernel, dim3(units, (ncols + 3) / 4, nbatch), dim3(256), 0, st, in, ldi, sIn, out, ldo,
                           sOut, nlarge, d_large_ids, d_qoff, d_qdim, ml, cdim, vstride, bstride, d_di, d_v, d_beta, ncols);
        KKT_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

}  // namespace mi355kkt

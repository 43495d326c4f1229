// C ABI of libmi355kkt (see include/mi355kkt.h): solver handle, factor(), solve(), stand-alone ops.
//
// Device engine = reduced ("Schur complement") form shared by every factory flavour
// (reference misc.py:1352-1567 is the closest restatement; Appendix A of SURVEY.md):
//     S = H + Gs' Gs                Gs = W^-T G            -> syrk_tn_kernel (scaling fused)
//     S = L L'                                             -> launch_potrf
//     Asct = L^-1 A',  K = Asct' Asct = Lk Lk'  (p > 0)    -> trsm_lower, syrk_tn_kernel, launch_potrf
//   solve:  zs = W^-T bz;  x = L^-1 (bx + Gs' zs [+ A' by]);  y = K^-1 (Asct' x - by);
//           x = L^-T (x - Asct y);  z = Gs x - zs
// G and A stay resident in HBM; per factor() only W (O(cdim) doubles) crosses PCIe, per solve() only
// x, y, z.
#include <unistd.h>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <mutex>
#include <new>
#include <stdexcept>

#include "../../include/mi355kkt.h"
#include "kkt_common.h"
#include <dlfcn.h>
#include "ordering.h"
#include "cone_ops.h"
#include <functional>

namespace mi355kkt {

static thread_local char g_err[512] = "";

void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// No C++ exception may cross the C ABI (an uncaught std::bad_alloc from a host-side std::vector, a std::system_error ... would
// std::terminate the caller's process -- the Python interpreter).  Every non-trivial entry point below is a function-try-block
// that ends here: the exception becomes an error code and a message for mi355kkt_last_error().
int kkt_catch(const char* what) noexcept {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        set_last_error("%s: out of host memory", what);
        return MI355KKT_ENOMEM;
    } catch (const std::exception& e) {
        set_last_error("%s: C++ exception: %s", what, e.what());
        return MI355KKT_EHIP;
    } catch (...) {
        set_last_error("%s: unknown C++ exception", what);
        return MI355KKT_EHIP;
    }
}

// ---- tiny element-wise helpers -------------------------------------------------------------------
// per-problem info of the batched engine: a failing pivot of K_b (rank(A_b) < p) is reported as n + pivot
__global__ void batch_merge_info_kernel(int* __restrict__ info, const int* __restrict__ infoK, int n, int nbatch) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nbatch && info[i] <= 0 && infoK[i] > 0) info[i] = n + infoK[i];
}
__global__ void scal_kernel(double* x, int n, double a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] *= a;
}
__global__ void diag_add_kernel(double* A, int64_t lda, int n, double a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) A[i + (int64_t)i * lda] += a;
}
__global__ void scaled_copy_kernel(const double* x, double* y, int n, double a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = a * x[i];
}
// mirror the lower triangle of nbatch n x n matrices into their upper triangles (batched along blockIdx.z)
__global__ void fill_kernel(double* x, double a, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = a;
}
__global__ void mul_kernel(double* out, const double* a, const double* b, double alpha, int n) {   // out = alpha a .* b  (b may be null)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = alpha * a[i] * (b ? b[i] : 1.0);
}
__global__ void add3_kernel(double* out, const double* a, const double* b, double beta, int n) {   // out = a + beta b
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] + beta * b[i];
}
__global__ void axpby_kernel(double* y, const double* x, double a, int64_t n) {   // y = a x
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a * x[i];
}
__global__ void row_gather_kernel(const double* __restrict__ row, int64_t ld, int n, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;       // out[i] = A[j][i] for the row whose first entry is `row`
    if (i < n) out[i] = row[(int64_t)i * ld];
}
__global__ void symmetrize_kernel(double* A, int n, int64_t bstride) {
    A += (int64_t)blockIdx.z * bstride;
    const int i = blockIdx.x * 16 + threadIdx.x, j = blockIdx.y * 16 + threadIdx.y;
    if (i < n && j < n && i > j) A[j + (int64_t)i * n] = A[i + (int64_t)j * n];
}
// out (n x p, ld n) = A' where A is p x n (lda)
__global__ void transpose_kernel(const double* __restrict__ A, int64_t lda, int p, int n, double* __restrict__ out,
                                 int64_t sA = 0, int64_t sOut = 0) {      // blockIdx.z: batched problems
    __shared__ double t[32][33];
    A += (int64_t)blockIdx.z * sA;
    out += (int64_t)blockIdx.z * sOut;
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // bx over n (cols of A), by over p (rows of A)
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = by + threadIdx.x, j = bx + r;   // A[i, j]
        t[r][threadIdx.x] = (i < p && j < n) ? A[i + (int64_t)j * lda] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int j = bx + threadIdx.x, i = by + r;   // out[j, i]
        if (j < n && i < p) out[j + (int64_t)i * n] = t[threadIdx.x][r];
    }
}

static inline dim3 g1(int n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace mi355kkt

using namespace mi355kkt;

// state of the device-resident interior-point loop (batch_ipm.hip) for B problems of one shape
struct IpmWork {
    IpmState S;
    double* f64 = nullptr;
    int* i32 = nullptr;       // active | status | iters | freeze | nactive | info
    int* pinned = nullptr;
    int B = 0;
};
static void ipm_free(IpmWork& w) {
    if (w.f64) (void)dev_free(w.f64);
    if (w.i32) (void)dev_free(w.i32);
    if (w.pinned) (void)hipHostFree(w.pinned);
    w = IpmWork();
}
static int ipm_alloc(IpmWork& w, int nbatch, int n, int m, int np = 0) {
    if (w.f64) return 0;
    const size_t B = nbatch, N = n, M = m ? m : 1, Pq = np;
    const size_t nd = B * (8 * N + 13 * M + 6 * Pq + 9);
    // all or nothing: a partially allocated state must not look complete to the next call
    if (DEV_ALLOC(&w.f64, sizeof(double) * nd) != hipSuccess) { ipm_free(w); return MI355KKT_ENOMEM; }
    if (DEV_ALLOC(&w.i32, sizeof(int) * (5 * B + 1)) != hipSuccess) { ipm_free(w); return MI355KKT_ENOMEM; }
    if (hipHostMalloc(&w.pinned, sizeof(int) * 4) != hipSuccess) { ipm_free(w); return MI355KKT_ENOMEM; }
    w.B = nbatch;
    IpmState& S = w.S;
    S.n = n; S.m = m; S.p = np;
    double* p = w.f64;
    auto take = [&](size_t k) { double* r = p; p += k; return r; };
    S.q = take(B * N); S.x = take(B * N); S.rx = take(B * N); S.dx = take(B * N);
    S.GTz = take(B * N); S.Px = take(B * N); S.x_out = take(B * N);
    S.h = take(B * M); S.s = take(B * M); S.z = take(B * M); S.rz = take(B * M); S.dz = take(B * M); S.ds = take(B * M);
    S.lmbda = take(B * M); S.d = take(B * M); S.di = take(B * M); S.ws3 = take(B * M); S.Gx = take(B * M);
    S.s_out = take(B * M); S.z_out = take(B * M);
    S.gap = take(B); S.resx0 = take(B); S.resz0 = take(B); S.step = take(B); S.sigma = take(B);
    S.pcost = take(B); S.dcost = take(B); S.gap_out = take(B); S.resy0 = take(B);
    S.ATy = take(B * N);
    if (np > 0) {
        S.b = take(B * Pq); S.y = take(B * Pq); S.ry = take(B * Pq); S.dy = take(B * Pq); S.Ax = take(B * Pq);
        S.y_out = take(B * Pq);
    }
    int* q = w.i32;
    S.active = q; S.status = q + B; S.iters = q + 2 * B; S.freeze = q + 3 * B; S.nactive = q + 4 * B;
    return 0;
}
static int* ipm_info_words(IpmWork& w) { return w.i32 + 4 * (size_t)w.B + 1; }

struct LpWork {
    LpState S;
    double* f64 = nullptr;
    int* i32 = nullptr;       // active | status | iters | init_optimal | nactive | info | qoff[nq] | qdim[nq]
    int* pinned = nullptr;
};
static void lp_free(LpWork& w) {
    if (w.f64) (void)dev_free(w.f64);
    if (w.i32) (void)dev_free(w.i32);
    if (w.pinned) (void)hipHostFree(w.pinned);
    w = LpWork();
}
// 's' blocks of the device loops: sizes of the extra state (r, rti, three scratch matrices per block, the Jacobi scratch,
// sigs, sigz) and the int descriptors sdim | soff | sloff
struct SBlocks {
    int ns = 0, sums = 0, sums2 = 0, maxs = 0, mins = 0;
    explicit SBlocks(const std::vector<int>& s) : ns((int)s.size()) {
        mins = ns ? s[0] : 0;
        for (int k : s) { sums += k; sums2 += k * k; maxs = std::max(maxs, k); mins = std::min(mins, k); }
    }
    size_t jw_doubles() const { return s_jw_doubles(maxs, 1024) + 16 * s_jw_doubles(16, 64); }   // workgroup team + 16 wave teams
    size_t doubles() const { return ns ? 5 * (size_t)sums2 + 2 * (size_t)sums + jw_doubles() + 8 : 8; }
};
template <class ST>
static int sblocks_bind(ST& S, const SBlocks& sb, const std::vector<int>& s, int lq, double*& p, int* di) {
    S.ns = sb.ns; S.lq = lq; S.ldim = lq + sb.sums;
    // with 's' blocks: 1024 threads (16 waves rotate 16 column pairs of a Jacobi round at a time) and as much of the 160 KB
    // LDS as G and V of the largest block need
    S.nthreads = sb.ns ? 1024 : 256;
    S.lds_doubles = sb.ns ? (int)std::min<size_t>(20416, 2 * (size_t)sb.maxs * sb.maxs) : 0;
    auto take = [&](size_t k) { double* r = p; p += (k ? k : 1); return r; };
    S.r = take(sb.sums2); S.rti = take(sb.sums2); S.sw1 = take(sb.sums2); S.sw2 = take(sb.sums2); S.sw3 = take(sb.sums2);
    S.sigs = take(sb.sums); S.sigz = take(sb.sums); S.jw = take(sb.ns ? sb.jw_doubles() : 1);
    S.jww = S.jw + (sb.ns ? s_jw_doubles(sb.maxs, 1024) : 0);
    S.smin = sb.mins; S.smax = sb.maxs;
    S.swmax = dev_knob("MI355KKT_SDP_WAVE_MAX") ? std::min(16, std::max(-16, atoi(dev_knob("MI355KKT_SDP_WAVE_MAX")))) : 16;
    S.sdim = di; S.soff = di + sb.ns; S.sloff = di + 2 * sb.ns;
    if (sb.ns) {
        std::vector<int> h(3 * (size_t)sb.ns);
        int off = lq, loff = lq;
        for (int k = 0; k < sb.ns; ++k) {
            h[k] = s[k]; h[sb.ns + k] = off; h[2 * sb.ns + k] = loff;
            off += s[k] * s[k]; loff += s[k];
        }
        if (memcpy_sync(di, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice) != hipSuccess) return MI355KKT_EHIP;
    }
    return 0;
}

static int lp_alloc(LpWork& w, int n, int ml, const std::vector<int>& q, const std::vector<int>& sd, int np) {
    if (w.f64) return 0;
    int sumq = 0;
    for (int k : q) sumq += k;
    const SBlocks sb(sd);
    const int m = ml + sumq + sb.sums2, nq = (int)q.size();
    const size_t N = n ? n : 1, M = m ? m : 1, Pq = np ? np : 1;
    const size_t nd = 10 * N + 9 * Pq + 23 * M + (size_t)sumq + nq + LP_NSC + 8 + sb.doubles();
    // all or nothing: a partially allocated state must not look complete to the next call
    if (DEV_ALLOC(&w.f64, sizeof(double) * nd) != hipSuccess) { lp_free(w); return MI355KKT_ENOMEM; }
    if (DEV_ALLOC(&w.i32, sizeof(int) * (8 + 2 * (size_t)nq + 3 * (size_t)sb.ns)) != hipSuccess) { lp_free(w); return MI355KKT_ENOMEM; }
    if (hipHostMalloc(&w.pinned, sizeof(int) * 4) != hipSuccess) { lp_free(w); return MI355KKT_ENOMEM; }
    LpState& S = w.S;
    S.n = n; S.m = m; S.p = np; S.ml = ml; S.nq = nq;
    double* p = w.f64;
    if (int e = sblocks_bind(S, sb, sd, ml + sumq, p, w.i32 + 8 + 2 * nq)) { lp_free(w); return e; }
    auto take = [&](size_t k) { double* r = p; p += k; return r; };
    S.c = take(N); S.x = take(N); S.dx = take(N); S.rx = take(N); S.x1 = take(N); S.GTz = take(N); S.ATy = take(N); S.x_out = take(N);
    S.wx = take(N); S.wx2 = take(N);
    S.b = take(Pq); S.y = take(Pq); S.dy = take(Pq); S.ry = take(Pq); S.y1 = take(Pq); S.Ax = take(Pq); S.y_out = take(Pq);
    S.wy = take(Pq); S.wy2 = take(Pq);
    S.h = take(M); S.s = take(M); S.z = take(M); S.ds = take(M); S.dz = take(M); S.rz = take(M); S.z1 = take(M); S.th = take(M);
    S.lmbda = take(M); S.lmbdasq = take(M); S.d = take(M); S.di = take(M); S.ws3 = take(M); S.Gx = take(M); S.s_out = take(M);
    S.z_out = take(M); S.t1 = take(M); S.t2 = take(M); S.wz3 = take(M); S.ws = take(M); S.wz = take(M); S.ws2 = take(M);
    S.wz2 = take(M);
    S.v = take(sumq ? sumq : 1); S.beta = take(nq ? nq : 1);
    S.sc = take(LP_NSC);
    int* qi = w.i32;
    S.active = qi; S.status = qi + 1; S.iters = qi + 2; S.init_optimal = qi + 3; S.nactive = qi + 4;
    if (nq) {
        std::vector<int> hq(2 * (size_t)nq);
        int off = ml;
        for (int k = 0; k < nq; ++k) { hq[k] = off; hq[nq + k] = q[k]; off += q[k]; }
        if (memcpy_sync(qi + 8, hq.data(), sizeof(int) * 2 * nq, hipMemcpyHostToDevice) != hipSuccess) return MI355KKT_EHIP;
    }
    S.qoff = qi + 8; S.qdim = qi + 8 + nq;
    return 0;
}

struct QpWork {
    QpState S;
    double* f64 = nullptr;
    int* i32 = nullptr;       // active | status | iters | nactive | info | - | - | - | qoff[nq] | qdim[nq]
    int* pinned = nullptr;
};
static void qp_free(QpWork& w) {
    if (w.f64) (void)dev_free(w.f64);
    if (w.i32) (void)dev_free(w.i32);
    if (w.pinned) (void)hipHostFree(w.pinned);
    w = QpWork();
}
static int qp_alloc(QpWork& w, int n, int ml, const std::vector<int>& q, const std::vector<int>& sd, int np, int nbatch = 1) {
    if (w.f64) return 0;
    int sumq = 0;
    for (int k : q) sumq += k;
    const SBlocks sb(sd);
    const int m = ml + sumq + sb.sums2, nq = (int)q.size();
    const size_t N = n ? n : 1, M = m ? m : 1, Pq = np ? np : 1;
    const size_t B = nbatch > 1 ? nbatch : 1;          // batched mode: every per-problem array holds B slices (QpState::nbatch)
    if (B > 1 && sb.ns > 0) return MI355KKT_ENOTIMPL;
    const size_t nd = B * (10 * N + 8 * Pq + 21 * M + (size_t)(sumq ? sumq : 1) + (nq ? nq : 1) + QP_NSC) + 8 + sb.doubles();
    // all or nothing: a partially allocated state must not look complete to the next call
    if (DEV_ALLOC(&w.f64, sizeof(double) * nd) != hipSuccess) { qp_free(w); return MI355KKT_ENOMEM; }
    if (DEV_ALLOC(&w.i32, sizeof(int) * (8 + 2 * (size_t)nq + 3 * (size_t)sb.ns + 4 * B)) != hipSuccess) { qp_free(w); return MI355KKT_ENOMEM; }
    if (hipHostMalloc(&w.pinned, sizeof(int) * 4) != hipSuccess) { qp_free(w); return MI355KKT_ENOMEM; }
    QpState& S = w.S;
    S.n = n; S.m = m; S.p = np; S.ml = ml; S.nq = nq; S.nbatch = (int)B;
    double* p = w.f64;
    if (int e = sblocks_bind(S, sb, sd, ml + sumq, p, w.i32 + 8 + 2 * nq)) { qp_free(w); return e; }
    auto take = [&](size_t k) { double* r = p; p += B * k; return r; };
    S.q = take(N); S.x = take(N); S.dx = take(N); S.rx = take(N); S.Px = take(N); S.GTz = take(N); S.ATy = take(N); S.x_out = take(N);
    S.wx = take(N); S.wx2 = take(N);
    S.b = take(Pq); S.y = take(Pq); S.dy = take(Pq); S.ry = take(Pq); S.Ax = take(Pq); S.y_out = take(Pq); S.wy = take(Pq);
    S.wy2 = take(Pq);
    S.h = take(M); S.s = take(M); S.z = take(M); S.ds = take(M); S.dz = take(M); S.rz = take(M); S.lmbda = take(M);
    S.lmbdasq = take(M); S.d = take(M); S.di = take(M); S.ws3 = take(M); S.Gx = take(M); S.s_out = take(M); S.z_out = take(M);
    S.t1 = take(M); S.t2 = take(M); S.wz3 = take(M); S.ws = take(M); S.wz = take(M); S.ws2 = take(M); S.wz2 = take(M);
    S.v = take(sumq ? sumq : 1); S.beta = take(nq ? nq : 1);
    S.sc = take(QP_NSC);
    int* qi = w.i32;
    S.active = qi; S.status = qi + 1; S.iters = qi + 2; S.nactive = qi + 3;
    if (B > 1) {          // active | status | iters | info, B words each, behind the cone descriptors
        int* bi = qi + 8 + 2 * (size_t)nq + 3 * (size_t)sb.ns;
        S.active = bi; S.status = bi + B; S.iters = bi + 2 * B;
    }
    if (nq) {
        std::vector<int> hq(2 * (size_t)nq);
        int off = ml;
        for (int k = 0; k < nq; ++k) { hq[k] = off; hq[nq + k] = q[k]; off += q[k]; }
        if (memcpy_sync(qi + 8, hq.data(), sizeof(int) * 2 * nq, hipMemcpyHostToDevice) != hipSuccess) return MI355KKT_EHIP;
    }
    S.qoff = qi + 8; S.qdim = qi + 8 + nq;
    return 0;
}

constexpr bool TRSV_WIDE_DEFAULT = true;      // knob MI355KKT_TRSV_WIDE=0: the round-4 pair kernel instead of the 512-row hops (A/B, tests)
constexpr bool TRSV_PAIR_DEFAULT = true;      // knob MI355KKT_TRSV_PAIR=0 selects the one-sweep kernel (A/B measurements, tests)

struct mi355kkt_solver {
    int device = 0, kind = 0;
    int n = 0, p = 0, ml = 0, cdim = 0;
    std::vector<int> q, s;
    int num_cus = 256;
    hipStream_t st = nullptr;
    hipEvent_t ev[8] = {};   // 0-3 factor phases, 4-5 solve, 6-7 around the dominant SYRK kernel
    // constants
    const double* dG = nullptr;  int64_t ldG = 0;  double* G_owned = nullptr;
    const double* dA = nullptr;  int64_t ldA = 0;  double* A_owned = nullptr;
    const double* dH = nullptr;  int64_t ldH = 0;  double* H_owned = nullptr;
    // asynchronous upload of a host H (mi355kkt_set_H_dense_async): own copy stream, the SYRK does not wait for it
    hipStream_t cst = nullptr;
    hipEvent_t ev_h = nullptr;
    bool h_pending = false;
    const void* reg_ptr = nullptr;  size_t reg_bytes = 0;    // host range currently pinned with hipHostRegister
    double kktreg = 0.0;
    // per-factor state
    double* dW = nullptr;      // effective diagonal scaling of the 'l' block (di, possibly / sqrt(1+reg))
    // second-order cones: Gs = W^-T G is materialised (the cone transform is not diagonal)
    ConeLayout cl;
    double* dGs = nullptr;     // cdim x n, only when nq > 0
    double* dV = nullptr;      // concatenated v_k
    double* dBeta = nullptr;   // beta_k
    double* dRti = nullptr;    // concatenated rti_k ('s' cones)
    int krows = 0;             // rows of the scaled constraint matrix the SYRK contracts over (packed for 's' cones)
    double* dWst = nullptr;    // staging for host-side W (di | v | beta)
    // persistent triangular solves: hand-off flags (one word per 128-block), launch epoch, timeout word
    unsigned long long* dgran = nullptr;   // data-tagged granules of the solved blocks (256 per 128-block)
    unsigned int epoch = 0;
    int* derr = nullptr;
    int* herr = nullptr;   // pinned
    // sparse mode (config 4): S is factored by the supernodal multifrontal engine instead of the dense one
    bool sparse = false;
    SparseEngine sp;
    int sp_extra = 0;          // > 0: the engine's G carries the p rows of A below the cone rows (S + A'A fallback, misc.py:1433-1447)
    // sparse A (mi355kkt_set_A_csr): CSR for A x and the forward solves, CSC (= CSR of A') for A' y; no dense copy
    bool A_sparse = false;
    int64_t *dArp = nullptr, *dAcp = nullptr;
    int *dAci = nullptr, *dAri = nullptr;
    double *dAv = nullptr, *dAvc = nullptr;
    double* dS = nullptr;      // n x n: S then its Cholesky factor L
    double* dAsct = nullptr;   // n x p
    double* dK = nullptr;      // p x p
    bool firstcall = true, singular = false, factored = false;
    // workspaces
    double *dx = nullptr, *dy = nullptr, *dz = nullptr, *dzs = nullptr, *dtn = nullptr, *dtp = nullptr, *dwork = nullptr;
    double* hbuf = nullptr;    // pinned: max(n + p + cdim, W size)
    size_t hbuf_doubles = 0;
    SyrkPlan planS, planAtA, planK;
    PotrfWork pw;
    float t_syrk = 0, t_potrf = 0, t_schur = 0, t_factor = 0, t_solve = 0, t_syrk_kernel = 0;
    IpmWork ipm;               // device-resident coneqp loop (mi355kkt_coneqp_lp), allocated on first use
    LpWork lp;                 // device-resident conelp loop (mi355kkt_conelp)
    QpWork qp;                 // device-resident coneqp loop with second-order cones (mi355kkt_coneqp)
    double* dRef = nullptr;    // iterative refinement of the ldl flavours: bx0 | by0 | zs0 | rx | ry | rz | t  (2 n + 2 p + 3 krows doubles)
    double* dHsym = nullptr;   // full symmetric copy of H for the residual product P x
    bool hsym_valid = false;
    double* dIpmWork = nullptr;
    double* dSpWork = nullptr;  // GEMV workspace of the sparse engine's Schur-complement step (p > 0)
    // options['show_progress'] of the reference drivers: called once per iteration of the device-resident loops
    mi355kkt_progress_fn progress = nullptr;
    void* progress_user = nullptr;
    // mi355kkt_set_option
    int use_correction = 1;    // options['use_correction'] of solvers.coneqp (coneprog.py:1781)
    int ldl_refine = 2;        // refinement steps of the ldl / ldl2 flavours against the 3 x 3 system
    // round 6, the 'qr' mapping (misc.kkt_qr -> this engine): the same refinement, but only for factorisations whose reduced matrix
    // is ill conditioned -- (max L_ii / min L_ii)^2, a lower bound of cond(S), read back with the info word, above QR_REFINE_COND
    int qr_refine = 0;         // steps (0: off; mi355kkt_set_option "qr_refinement")
    bool qr_active = false;    // this factorisation's solves are refined
    double* d_cond = nullptr;  // {min, max} of diag(L), device / pinned host
    double* h_cond = nullptr;
    // ... and beyond QR2_COND the factor itself is repaired (CholeskyQR2, Yamamoto et al. 2015): with L1 = chol(Gs'Gs) -- inaccurate
    // there, but it exists -- Q1 = Gs L1^-T is nearly orthonormal, L2 = chol(Q1'Q1) is accurate, and S = L1 (L2 L2') L1' holds to
    // working precision column by column: the accuracy of a QR factor of Gs (what misc.kkt_qr computes) from three products the
    // engine already has (transpose, triangular solve with many right-hand sides, SYRK, Cholesky).  conelp without H only.
    // When even chol(Gs'Gs) breaks down (cond(Gs) beyond ~6e7), L1 = chol(Gs'Gs + sigma I) with sigma = 11 (rows n + n (n + 1)) eps
    // x (an upper bound of ||Gs||^2) always exists, cond(Q1) = sqrt(sigma) / sigma_min(Gs) is small, and TWO more passes give
    // S = L1 L2 (L3 L3') L2' L1' (shifted CholeskyQR3, Fukaya et al. 2020).
    int qr_extra = 0;          // factors this factorisation carries beyond L1 (0: none, 1: CholeskyQR2, 2: shifted CholeskyQR3)
    bool pwx_ready[2] = {false, false};
    double* dQt = nullptr;     // n x rows: Gs', then Q_k' = L_k^-1 Q_{k-1}'
    double* dQ = nullptr;      // rows x n: (LP cone: Gs first,) then Q_k
    double* dSx[2] = {nullptr, nullptr};     // n x n: Q_k'Q_k, then L_{k+1}
    PotrfWork pwx[2];
};

// per-iteration report of a device-resident loop: the scalar block comes back with one small copy (the loop has just
// synchronised on the "still active" word anyway)
static int report_progress(mi355kkt_solver* h, int it, const double* d_sc, int nsc, const int* idx, int nidx, int itau, int ikappa) {
    if (!h->progress) return 0;
    double sc[64];
    if (nsc > 64) nsc = 64;
    KKT_HIP_CHECK(memcpy_sync(sc, d_sc, sizeof(double) * nsc, hipMemcpyDeviceToHost));
    double vals[8];
    for (int k = 0; k < nidx; ++k) vals[k] = sc[idx[k]];
    int nv = nidx;
    if (itau >= 0) vals[nv++] = sc[ikappa] / sc[itau];
    h->progress(it, nv, vals, h->progress_user);
    return 0;
}

// ---- roctx ranges (tracing): the phases of factor() / solve() show up as named ranges in rocprofv3
// --marker-trace timelines.  libroctx64.so is looked up at run time on the first use with $MI355KKT_ROCTX=1 -- the library itself
// keeps linking only the HIP runtime.
struct RoctxApi { int (*push)(const char*) = nullptr; int (*pop)() = nullptr; bool tried = false; };
static RoctxApi& roctx_api() {
    static RoctxApi a;
    if (!a.tried) {
        a.tried = true;
        const char* e = getenv("MI355KKT_ROCTX");
        if (e && atoi(e) != 0) {
            void* lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (!lib) lib = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (lib) {
                a.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
                a.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
            }
        }
    }
    return a;
}
struct RoctxRange {
    bool on;
    explicit RoctxRange(const char* name) : on(false) {
        RoctxApi& a = roctx_api();
        if (a.push && a.pop) { a.push(name); on = true; }
    }
    void next(const char* name) {                      // close the current range, open the next one
        if (on) { roctx_api().pop(); roctx_api().push(name); }
    }
    ~RoctxRange() { if (on) roctx_api().pop(); }
};

static int bind(const mi355kkt_solver* h) {
    KKT_HIP_CHECK(hipSetDevice(h->device));
    return 0;
}

extern "C" {

int mi355kkt_version(void) { return 100; }
const char* mi355kkt_last_error(void) { return g_err; }

int mi355kkt_device_count(void) try {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
} catch (...) { return kkt_catch("mi355kkt_device_count"); }

int mi355kkt_device_info(int device, char* name, int len, int* num_cus, size_t* mem_bytes) try {
    hipDeviceProp_t prop;
    KKT_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (name && len > 0) {
        snprintf(name, (size_t)len, "%s (%s)", prop.name, prop.gcnArchName);
    }
    if (num_cus) *num_cus = prop.multiProcessorCount;
    if (mem_bytes) *mem_bytes = prop.totalGlobalMem;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_device_info"); }

int mi355kkt_dev_malloc(void** ptr, size_t bytes) try {
    KKT_HIP_CHECK(DEV_ALLOC(ptr, bytes ? bytes : 8));
    return 0;
} catch (...) { return kkt_catch("mi355kkt_dev_malloc"); }
int mi355kkt_dev_free(void* ptr) try {
    if (ptr) KKT_HIP_CHECK(dev_free(ptr));
    return 0;
} catch (...) { return kkt_catch("mi355kkt_dev_free"); }
int mi355kkt_memcpy_h2d(void* dst, const void* src, size_t bytes) try {
    if (bytes) KKT_HIP_CHECK(memcpy_sync(dst, src, bytes, hipMemcpyHostToDevice));
    return 0;
} catch (...) { return kkt_catch("mi355kkt_memcpy_h2d"); }
int mi355kkt_memcpy_d2h(void* dst, const void* src, size_t bytes) try {
    if (bytes) KKT_HIP_CHECK(memcpy_sync(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
} catch (...) { return kkt_catch("mi355kkt_memcpy_d2h"); }
int mi355kkt_memcpy_d2d(void* dst, const void* src, size_t bytes) try {
    if (bytes) KKT_HIP_CHECK(memcpy_sync(dst, src, bytes, hipMemcpyDeviceToDevice));
    return 0;
} catch (...) { return kkt_catch("mi355kkt_memcpy_d2d"); }
int mi355kkt_device_synchronize(void) try {
    KKT_HIP_CHECK(hipDeviceSynchronize());
    return 0;
} catch (...) { return kkt_catch("mi355kkt_device_synchronize"); }
// device memory shared between the processes of one node (batch scatter / gather without a collective)
int mi355kkt_ipc_export(const void* dptr, void* handle64, int64_t* offset, int64_t* alloc_bytes) try {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI promises a 64-byte handle");
    if (!dptr || !handle64 || !offset) { set_last_error("mi355kkt_ipc_export: null argument"); return MI355KKT_EINVAL; }
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    KKT_HIP_CHECK(hipMemGetAddressRange(&base, &size, const_cast<void*>(dptr)));
    hipIpcMemHandle_t hd;
    KKT_HIP_CHECK(hipIpcGetMemHandle(&hd, base));
    memcpy(handle64, &hd, sizeof(hd));
    *offset = (int64_t)((const char*)dptr - (const char*)base);
    if (alloc_bytes) *alloc_bytes = (int64_t)size;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_ipc_export"); }
int mi355kkt_ipc_open(const void* handle64, void** base) try {
    if (!handle64 || !base) { set_last_error("mi355kkt_ipc_open: null argument"); return MI355KKT_EINVAL; }
    hipIpcMemHandle_t hd;
    memcpy(&hd, handle64, sizeof(hd));
    void* p = nullptr;
    KKT_HIP_CHECK(hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess));
    *base = p;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_ipc_open"); }
int mi355kkt_ipc_close(void* base) try {
    if (base) KKT_HIP_CHECK(hipIpcCloseMemHandle(base));
    return 0;
} catch (...) { return kkt_catch("mi355kkt_ipc_close"); }

static size_t dmax(size_t a, size_t b) { return a > b ? a : b; }

int mi355kkt_create(mi355kkt_solver** out, int device, int kind, int n, int p, int ml, int nq, const int* q, int ns,
                    const int* s) try {
    if (!out || n < 0 || p < 0 || ml < 0 || nq < 0 || ns < 0 || kind < 0 || kind > 3) {
        set_last_error("mi355kkt_create: invalid argument");
        return MI355KKT_EINVAL;
    }
    if (mi355kkt_device_count() <= device) {
        set_last_error("mi355kkt_create: HIP device %d not available (count=%d)", device, mi355kkt_device_count());
        return MI355KKT_EHIP;
    }
    mi355kkt_solver* h = new mi355kkt_solver();
    h->device = device;
    h->kind = kind;
    h->n = n;
    h->p = p;
    h->ml = ml;
    int64_t cdim = ml;
    for (int k = 0; k < nq; ++k) {
        if (q[k] < 1) { delete h; set_last_error("q[%d] < 1", k); return MI355KKT_EINVAL; }
        h->q.push_back(q[k]);
        cdim += q[k];
    }
    for (int k = 0; k < ns; ++k) {
        if (s[k] < 0) { delete h; set_last_error("s[%d] < 0", k); return MI355KKT_EINVAL; }
        h->s.push_back(s[k]);
        cdim += (int64_t)s[k] * s[k];
    }
    if (cdim > INT32_MAX) { delete h; set_last_error("cdim overflows int"); return MI355KKT_EINVAL; }
    h->cdim = (int)cdim;
    int rc = 0;
    auto fail = [&](int code) { mi355kkt_destroy(h); return code; };
    if (hipSetDevice(device) != hipSuccess) return fail(MI355KKT_EHIP);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(MI355KKT_EHIP);
    h->num_cus = prop.multiProcessorCount;
    {   // main stream at the highest priority: its panel kernels overtake the low-priority bulk updates
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&h->st, hipStreamNonBlocking, greatest) != hipSuccess &&
            hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) != hipSuccess)
            return fail(MI355KKT_EHIP);
    }
    for (auto& e : h->ev)
        if (hipEventCreate(&e) != hipSuccess) return fail(MI355KKT_EHIP);
    auto alloc = [&](double** ptr, size_t doubles) -> int {
        if (DEV_ALLOC(ptr, sizeof(double) * dmax(doubles, 1)) != hipSuccess) {
            set_last_error("hipMalloc of %zu doubles failed", doubles);
            return MI355KKT_ENOMEM;
        }
        return 0;
    };
    const size_t N = (size_t)n, P = (size_t)p, C = (size_t)h->cdim;
    if ((rc = alloc(&h->dW, C + P))) return fail(rc);          // + p: the unit scaling of the A rows in the S + A'A sparse mode
    if ((rc = alloc(&h->dAsct, N * P))) return fail(rc);
    if ((rc = alloc(&h->dK, P * P))) return fail(rc);
    if ((rc = alloc(&h->dx, N))) return fail(rc);
    if ((rc = alloc(&h->dy, P))) return fail(rc);
    if ((rc = alloc(&h->dz, C + P))) return fail(rc);
    if ((rc = alloc(&h->dzs, C + P))) return fail(rc);
    if ((rc = alloc(&h->dtn, N))) return fail(rc);
    if ((rc = alloc(&h->dtp, P))) return fail(rc);
    if ((rc = alloc(&h->dwork, C + P + 8))) return fail(rc);   // grown to the dense GEMV workspace on first dense use
    if ((rc = alloc(&h->dWst, 2 * C + (size_t)nq + 8))) return fail(rc);
    {
        const size_t nfl = N / 128 + 2 + 64 * TRSV_JOB_STRIDE;    // + room for 64 batched jobs of the sparse engine's wide supernodes
        if (DEV_ALLOC(&h->dgran, sizeof(unsigned long long) * nfl * 256) != hipSuccess) return fail(MI355KKT_ENOMEM);
        if (memset_sync(h->dgran, 0, sizeof(unsigned long long) * nfl * 256) != hipSuccess) return fail(MI355KKT_EHIP);
        if (DEV_ALLOC(&h->derr, sizeof(int)) != hipSuccess) return fail(MI355KKT_ENOMEM);
        if (memset_sync(h->derr, 0, sizeof(int)) != hipSuccess) return fail(MI355KKT_EHIP);
        if (hipHostMalloc(&h->herr, sizeof(int)) != hipSuccess) return fail(MI355KKT_ENOMEM);
        *h->herr = 0;
    }
    h->krows = h->cdim;
    if (nq > 0 || ns > 0) {
        if ((rc = alloc(&h->dGs, C * N))) return fail(rc);
        if ((rc = alloc(&h->dV, C))) return fail(rc);
        if ((rc = alloc(&h->dBeta, (size_t)nq))) return fail(rc);
        if ((rc = alloc(&h->dRti, C))) return fail(rc);
        if ((rc = cone_layout_build(h->cl, ml, h->q))) return fail(rc);
        int lq = ml;
        for (int v : h->q) lq += v;
        if ((rc = cone_layout_build_s(h->cl, lq, h->s))) return fail(rc);
        h->krows = h->cl.cdim_packed;
    }
    h->hbuf_doubles = dmax(N + P + C, 2 * C + (size_t)nq) + C + 8;
    if (hipHostMalloc(&h->hbuf, sizeof(double) * h->hbuf_doubles) != hipSuccess) return fail(MI355KKT_ENOMEM);
    if ((rc = potrf_work_init(h->pw))) return fail(rc);
    if (p > 0) {
        if ((rc = build_syrk_plan(h->planAtA, n, p, h->num_cus))) return fail(rc);
        if ((rc = build_syrk_plan(h->planK, p, n, h->num_cus))) return fail(rc);
    }
    *out = h;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_create"); }

static void h_unregister(mi355kkt_solver* h);
static int ensure_hsym(mi355kkt_solver* hs);
static int ensure_gemv_work(mi355kkt_solver* hs);
void mi355kkt_destroy(mi355kkt_solver* h) try {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->st) (void)hipStreamSynchronize(h->st);
    double* bufs[] = {h->G_owned, h->A_owned, h->H_owned, h->dW, h->dS, h->dAsct, h->dK,
                      h->dx, h->dy, h->dz, h->dzs, h->dtn, h->dtp, h->dwork, h->dWst, h->dGs, h->dV, h->dBeta, h->dRti};
    cone_layout_free(h->cl);
    sparse_engine_free(h->sp);
    ipm_free(h->ipm);
    lp_free(h->lp);
    qp_free(h->qp);
    if (h->dHsym) (void)dev_free(h->dHsym);
    if (h->dIpmWork) (void)dev_free(h->dIpmWork);
    if (h->dSpWork) (void)dev_free(h->dSpWork);
    h_unregister(h);
    {
        void* ap[] = {h->dArp, h->dAcp, h->dAci, h->dAri, h->dAv, h->dAvc};
        for (void* q : ap) if (q) (void)dev_free(q);
    }
    if (h->cst) (void)hipStreamDestroy(h->cst);
    if (h->ev_h) (void)hipEventDestroy(h->ev_h);
    if (h->dgran) (void)dev_free(h->dgran);
    if (h->dRef) (void)dev_free(h->dRef);
    if (h->d_cond) (void)dev_free(h->d_cond);
    if (h->h_cond) (void)hipHostFree(h->h_cond);
    if (h->dQt) (void)dev_free(h->dQt);
    if (h->dQ) (void)dev_free(h->dQ);
    for (int k = 0; k < 2; ++k) {
        if (h->dSx[k]) (void)dev_free(h->dSx[k]);
        if (h->pwx_ready[k]) potrf_work_free(h->pwx[k]);
    }
    if (h->derr) (void)dev_free(h->derr);
    if (h->herr) (void)hipHostFree(h->herr);
    for (double* b : bufs)
        if (b) (void)dev_free(b);
    if (h->hbuf) (void)hipHostFree(h->hbuf);
    potrf_work_free(h->pw);
    free_syrk_plan(h->planS);
    free_syrk_plan(h->planAtA);
    free_syrk_plan(h->planK);
    for (auto& e : h->ev)
        if (e) (void)hipEventDestroy(e);
    if (h->st) (void)hipStreamDestroy(h->st);
    delete h;
} catch (...) { (void)kkt_catch("mi355kkt_destroy"); }

static int upload_dense(double** owned, const double* src, int64_t ld, int rows, int cols, hipStream_t st) {
    if (*owned) { (void)dev_free(*owned); *owned = nullptr; }
    const size_t bytes = sizeof(double) * dmax((size_t)rows * cols, 1);
    KKT_HIP_CHECK(DEV_ALLOC(owned, bytes));
    if (rows > 0 && cols > 0)
        KKT_HIP_CHECK(memcpy2d_sync(*owned, sizeof(double) * rows, src, sizeof(double) * ld, sizeof(double) * rows, cols,
                                  hipMemcpyHostToDevice));
    return 0;
}

int mi355kkt_set_G_dense(mi355kkt_solver* h, const double* G, int64_t ldG) try {
    if (!h || (!G && h->cdim > 0 && h->n > 0) || ldG < (h->cdim > 1 ? h->cdim : 1)) {
        set_last_error("set_G_dense: invalid argument");
        return MI355KKT_EINVAL;
    }
    if (int e = bind(h)) return e;
    if (int e = upload_dense(&h->G_owned, G, ldG, h->cdim, h->n, h->st)) return e;
    h->dG = h->G_owned;
    h->ldG = h->cdim > 1 ? h->cdim : 1;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_set_G_dense"); }

/* Overwrites rows [row0, row0 + nrows) of the dense G held by the handle (host source, column-major, ld = ldsrc): the
 * Jacobian block Df of the nonlinear constraints that cvxprog.cp / cpl prepend to G at every iteration
 * (misc.py:1265-1266 `Gs[:mnl,:] = Df`). */
int mi355kkt_set_G_rows(mi355kkt_solver* h, int row0, int nrows, const double* src, int64_t ldsrc) try {
    if (!h || nrows < 0 || row0 < 0 || row0 + nrows > h->cdim || (nrows > 0 && (!src || ldsrc < nrows))) {
        set_last_error("set_G_rows: invalid argument");
        return MI355KKT_EINVAL;
    }
    if (!h->G_owned || h->dG != h->G_owned) { set_last_error("set_G_rows: G must have been set with set_G_dense / set_G_csc"); return MI355KKT_EINVAL; }
    if (int e = bind(h)) return e;
    if (nrows > 0 && h->n > 0)
        KKT_HIP_CHECK(memcpy2d_sync(h->G_owned + row0, sizeof(double) * h->ldG, src, sizeof(double) * ldsrc, sizeof(double) * nrows,
                                  h->n, hipMemcpyHostToDevice));
    return 0;
} catch (...) { return kkt_catch("mi355kkt_set_G_rows"); }

int mi355kkt_set_G_csc(mi355kkt_solver* h, const int64_t* colptr, const int64_t* rowind, const double* values) try {
    if (!h || !colptr) { set_last_error("set_G_csc: invalid argument"); return MI355KKT_EINVAL; }
    std::vector<double> dense((size_t)h->cdim * h->n, 0.0);
    for (int j = 0; j < h->n; ++j)
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            if (rowind[k] < 0 || rowind[k] >= h->cdim) { set_last_error("set_G_csc: row index out of range"); return MI355KKT_EINVAL; }
            dense[(size_t)j * h->cdim + rowind[k]] += values[k];
        }
    return mi355kkt_set_G_dense(h, dense.data(), h->cdim > 1 ? h->cdim : 1);
} catch (...) { return kkt_catch("mi355kkt_set_G_csc"); }

/* A (p x n) in CSR: rowptr[p + 1], colind[nnz] (int64, like the CCS arrays of cvxopt), values[nnz].  The handle keeps A sparse
 * on the device (CSR + its transpose) -- reference misc.py:1483-1487 / cholmod.spsolve keep A' sparse too.  Sparse engine only. */
int mi355kkt_set_A_csr(mi355kkt_solver* h, const int64_t* rowptr_in, const int64_t* colind_in, const double* values_in) try {
    if (!h || (h->p > 0 && !rowptr_in)) { set_last_error("set_A_csr: null argument"); return MI355KKT_EINVAL; }
    const int p = h->p, n = h->n;
    if (p == 0) { h->A_sparse = true; return 0; }
    // validation first (no device needed): rowptr[0] == 0 and monotone, column indices in range
    if (rowptr_in[0] != 0) { set_last_error("set_A_csr: rowptr[0] must be 0"); return MI355KKT_EINVAL; }
    for (int r = 0; r < p; ++r)
        if (rowptr_in[r + 1] < rowptr_in[r]) { set_last_error("set_A_csr: rowptr must be non-decreasing"); return MI355KKT_EINVAL; }
    const int64_t nnz_in = rowptr_in[p];
    if (nnz_in > 0 && (!colind_in || !values_in)) { set_last_error("set_A_csr: null argument"); return MI355KKT_EINVAL; }
    for (int64_t k = 0; k < nnz_in; ++k)
        if (colind_in[k] < 0 || colind_in[k] >= n) { set_last_error("set_A_csr: column index out of range"); return MI355KKT_EINVAL; }
    // canonical form: columns ascending within a row, repeated (row, column) entries summed in input order -- what the dense
    // upload does with repeated triplets; the scatter kernels of the sparse engine assume one entry per (row, column)
    std::vector<int64_t> rp((size_t)p + 1, 0), colind;
    std::vector<double> values;
    colind.reserve((size_t)nnz_in);
    values.reserve((size_t)nnz_in);
    {
        std::vector<int64_t> ord;
        for (int r = 0; r < p; ++r) {
            const int64_t a = rowptr_in[r], b = rowptr_in[r + 1];
            ord.resize((size_t)(b - a));
            for (int64_t k = a; k < b; ++k) ord[(size_t)(k - a)] = k;
            std::stable_sort(ord.begin(), ord.end(), [&](int64_t x, int64_t y) { return colind_in[x] < colind_in[y]; });
            for (size_t t = 0; t < ord.size(); ++t) {
                const int64_t k = ord[t];
                if (t > 0 && colind_in[k] == colind.back() && (int64_t)colind.size() > rp[r]) values.back() += values_in[k];
                else { colind.push_back(colind_in[k]); values.push_back(values_in[k]); }
            }
            rp[(size_t)r + 1] = (int64_t)colind.size();
        }
    }
    const int64_t* rowptr = rp.data();
    const int64_t nnz = rp[p];
    if (int e = bind(h)) return e;
    std::vector<int> ci((size_t)nnz), ri((size_t)nnz);
    std::vector<int64_t> cp((size_t)n + 1, 0);
    std::vector<double> vc((size_t)nnz);
    for (int64_t k = 0; k < nnz; ++k) {
        ci[k] = (int)colind[k];
        ++cp[colind[k] + 1];
    }
    for (int j = 0; j < n; ++j) cp[j + 1] += cp[j];
    {
        std::vector<int64_t> nxt(cp.begin(), cp.end() - 1);
        for (int r = 0; r < p; ++r)
            for (int64_t k = rowptr[r]; k < rowptr[r + 1]; ++k) {
                const int64_t q = nxt[colind[k]]++;
                ri[q] = r;
                vc[q] = values[k];
            }
    }
    void* old[] = {h->dArp, h->dAcp, h->dAci, h->dAri, h->dAv, h->dAvc};
    for (void* q : old) if (q) (void)dev_free(q);
    h->dArp = h->dAcp = nullptr; h->dAci = h->dAri = nullptr; h->dAv = h->dAvc = nullptr;
    const size_t z = (size_t)(nnz > 0 ? nnz : 1);
    KKT_HIP_CHECK(DEV_ALLOC(&h->dArp, sizeof(int64_t) * (p + 1)));
    KKT_HIP_CHECK(DEV_ALLOC(&h->dAcp, sizeof(int64_t) * ((size_t)n + 1)));
    KKT_HIP_CHECK(DEV_ALLOC(&h->dAci, sizeof(int) * z));
    KKT_HIP_CHECK(DEV_ALLOC(&h->dAri, sizeof(int) * z));
    KKT_HIP_CHECK(DEV_ALLOC(&h->dAv, sizeof(double) * z));
    KKT_HIP_CHECK(DEV_ALLOC(&h->dAvc, sizeof(double) * z));
    KKT_HIP_CHECK(memcpy_sync(h->dArp, rp.data(), sizeof(int64_t) * (p + 1), hipMemcpyHostToDevice));
    KKT_HIP_CHECK(memcpy_sync(h->dAcp, cp.data(), sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice));
    if (nnz > 0) {
        KKT_HIP_CHECK(memcpy_sync(h->dAci, ci.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
        KKT_HIP_CHECK(memcpy_sync(h->dAri, ri.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
        KKT_HIP_CHECK(memcpy_sync(h->dAv, values.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
        KKT_HIP_CHECK(memcpy_sync(h->dAvc, vc.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
    }
    h->A_sparse = true;
    h->dA = nullptr;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_set_A_csr"); }

/* set_sparse_problem whose G carries `extra_rows` more rows below the cdim cone rows: the rows of A with unit scaling, i.e.
 * S = H + G'D^2 G + A'A -- the reference's fallback for a singular S on the first factorisation (misc.py:1433-1447), which in
 * sparse mode needs a new symbolic analysis because the pattern of S grows.  The handle is in "singular" mode afterwards
 * (solve() adds A'by to bx, misc.py:1527).  extra_rows is 0 or p. */
int mi355kkt_set_sparse_problem_aug(mi355kkt_solver* h, const int64_t* gcolptr, const int64_t* growind, const double* gvalues,
                                    const int64_t* hcolptr, const int64_t* hrowind, const double* hvalues, int extra_rows) try {
    if (!h || !gcolptr) { set_last_error("set_sparse_problem: null argument"); return MI355KKT_EINVAL; }
    if (extra_rows != 0 && extra_rows != h->p) { set_last_error("set_sparse_problem_aug: extra_rows must be 0 or p"); return MI355KKT_EINVAL; }
    if (!h->q.empty() || !h->s.empty()) {
        set_last_error("set_sparse_problem: the sparse engine handles LP cones");
        return MI355KKT_ENOTIMPL;
    }
    if (int e = bind(h)) return e;
    const int rows = h->cdim + extra_rows;
    for (int j = 0; j < h->n; ++j)
        for (int64_t k = gcolptr[j]; k < gcolptr[j + 1]; ++k)
            if (growind[k] < 0 || growind[k] >= rows) { set_last_error("set_sparse_problem: G row index out of range"); return MI355KKT_EINVAL; }
    if (int e = sparse_engine_create(h->sp, h->n, rows, gcolptr, growind, gvalues, hcolptr, hrowind, hvalues)) return e;
    h->sp.t_gran = h->dgran; h->sp.t_err = h->derr; h->sp.t_epoch = &h->epoch; h->sp.t_njobs_max = 64; h->sp.t_num_cus = h->num_cus;
    h->sparse = true;
    h->firstcall = true;
    h->sp_extra = extra_rows;
    h->singular = extra_rows > 0;
    h->factored = false;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_set_sparse_problem_aug"); }

int mi355kkt_set_sparse_problem(mi355kkt_solver* h, const int64_t* gcolptr, const int64_t* growind, const double* gvalues,
                                const int64_t* hcolptr, const int64_t* hrowind, const double* hvalues) try {
    return mi355kkt_set_sparse_problem_aug(h, gcolptr, growind, gvalues, hcolptr, hrowind, hvalues, 0);
} catch (...) { return kkt_catch("mi355kkt_set_sparse_problem"); }
int mi355kkt_sparse_stats(const mi355kkt_solver* h, int64_t* nnzL, int* nsupernodes, int* nlevels, double* flops) try {
    if (!h || !h->sparse) return MI355KKT_EINVAL;
    if (nnzL) *nnzL = h->sp.sym.nnzL;
    if (nsupernodes) *nsupernodes = h->sp.sym.ns;
    if (nlevels) *nlevels = h->sp.sym.nlevels;
    if (flops) *flops = h->sp.sym.flops;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_sparse_stats"); }

/* which fill-reducing ordering the symbolic analysis chose: 1 nested dissection, 2 approximate minimum degree */
int mi355kkt_sparse_ordering(const mi355kkt_solver* h) try {
    if (!h || !h->sparse) return MI355KKT_EINVAL;
    return h->sp.sym.order_method;
} catch (...) { return kkt_catch("mi355kkt_sparse_ordering"); }

int mi355kkt_set_A_dense(mi355kkt_solver* h, const double* A, int64_t ldA) try {
    if (!h || (!A && h->p > 0 && h->n > 0) || ldA < (h->p > 1 ? h->p : 1)) {
        set_last_error("set_A_dense: invalid argument");
        return MI355KKT_EINVAL;
    }
    if (int e = bind(h)) return e;
    if (int e = upload_dense(&h->A_owned, A, ldA, h->p, h->n, h->st)) return e;
    h->dA = h->A_owned;
    h->ldA = h->p > 1 ? h->p : 1;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_set_A_dense"); }

int mi355kkt_set_G_device(mi355kkt_solver* h, const double* dG, int64_t ldG) try {
    if (!h || ldG < (h->cdim > 1 ? h->cdim : 1)) { set_last_error("set_G_device: invalid argument"); return MI355KKT_EINVAL; }
    h->dG = dG;
    h->ldG = ldG;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_set_G_device"); }
int mi355kkt_set_A_device(mi355kkt_solver* h, const double* dA, int64_t ldA) try {
    if (!h || ldA < (h->p > 1 ? h->p : 1)) { set_last_error("set_A_device: invalid argument"); return MI355KKT_EINVAL; }
    h->dA = dA;
    h->ldA = ldA;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_set_A_device"); }

int mi355kkt_set_H_dense(mi355kkt_solver* h, const double* H, int64_t ldH) try {
    if (!h) return MI355KKT_EINVAL;
    if (int e = bind(h)) return e;
    h->hsym_valid = false;
    if (h->h_pending) {                                    // an asynchronous upload is still in flight: let it land first
        KKT_HIP_CHECK(hipStreamSynchronize(h->cst));
        h->h_pending = false;
    }
    h_unregister(h);                                       // the caller may free the buffer an earlier async call pinned
    if (!H) {
        h->dH = nullptr;
        return 0;
    }
    if (ldH < (h->n > 1 ? h->n : 1)) { set_last_error("set_H_dense: ldH too small"); return MI355KKT_EINVAL; }
    if (!h->H_owned) KKT_HIP_CHECK(DEV_ALLOC(&h->H_owned, sizeof(double) * dmax((size_t)h->n * h->n, 1)));
    // nothing on the compute stream may still read the old H (a queued solve_device of the ldl flavours reads it for its residual,
    // a device loop for P x): the copy below is only ordered on the legacy stream
    if (h->st) KKT_HIP_CHECK(hipStreamSynchronize(h->st));
    if (h->n > 0)
        KKT_HIP_CHECK(memcpy2d_sync(h->H_owned, sizeof(double) * h->n, H, sizeof(double) * ldH, sizeof(double) * h->n,
                                  h->n, hipMemcpyHostToDevice));
    h->dH = h->H_owned;
    h->ldH = h->n > 1 ? h->n : 1;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_set_H_dense"); }
static void h_unregister(mi355kkt_solver* h) {
    if (h->reg_ptr) {
        if (h->cst) (void)hipStreamSynchronize(h->cst);
        (void)hipHostUnregister(const_cast<void*>(h->reg_ptr));
        h->reg_ptr = nullptr;
        h->reg_bytes = 0;
    }
}
/* Same as set_H_dense, but the copy is enqueued on the handle's copy stream from the caller's buffer pinned in place
 * (hipHostRegister, cached while the same buffer is passed again) and the NEXT factor() only waits for it after the
 * scaled SYRK: S = Gs'Gs runs while H crosses PCIe, then S += tril(H).  The caller keeps H alive and unmodified until
 * that factor() returns, and alive until the next set_H_* call or destroy (the Python mirror holds a reference). */
// Does [a, b) touch the process's brk heap (the "[heap]" line of /proc/self/maps)?  Memory there is recycled by malloc under the
// caller's feet -- chunks are split, merged, trimmed and grown back -- and must never be hipHostRegister'ed: round 4's abort (small
// H, always a heap chunk) and a GPU memory fault in round 6's one-process suite (8-32 MB H from a heap whose mmap threshold glibc had
// raised) both ended on heap addresses.  Read at every call (the heap moves); ~50 us next to a copy of megabytes.
static bool touches_brk_heap(uintptr_t a, uintptr_t b) {
    FILE* f = fopen("/proc/self/maps", "r");
    if (!f) return true;                                   // cannot tell: do not pin
    char line[512];
    bool hit = false;
    while (fgets(line, sizeof line, f)) {
        if (!strstr(line, "[heap]")) continue;
        unsigned long long lo = 0, hi = 0;
        if (sscanf(line, "%llx-%llx", &lo, &hi) == 2 && a < (uintptr_t)hi && b > (uintptr_t)lo) hit = true;
    }
    fclose(f);
    return hit;
}

int mi355kkt_set_H_dense_async(mi355kkt_solver* h, const double* H, int64_t ldH) try {
    if (!h) return MI355KKT_EINVAL;
    if (!H || h->n == 0) { h_unregister(h); h->h_pending = false; return mi355kkt_set_H_dense(h, H, ldH); }
    if (int e = bind(h)) return e;
    if (ldH < (h->n > 1 ? h->n : 1)) { set_last_error("set_H_dense_async: ldH too small"); return MI355KKT_EINVAL; }
    const size_t bytes = sizeof(double) * ((size_t)ldH * (h->n - 1) + h->n);
    // A small H is uploaded synchronously: pinning it in place would lock whole pages of the caller's heap (a 10 x 10 matrix
    // shares its page with unrelated allocations, the caller's and the runtime's) for an overlap that only matters when the copy
    // takes as long as a kernel -- the headline's H is 512 MB.
    // (test knob MI355KKT_PIN_SMALL_H: pin whatever the size -- the behaviour of the build that aborted in round 4)
    const bool pin_any = dev_knob("MI355KKT_PIN_SMALL_H") != nullptr;
    if (bytes < ((size_t)4 << 20) && !pin_any)
        return mi355kkt_set_H_dense(h, H, ldH);                                  // (waits for a pending upload, unpins)
    // Round 5: only WHOLE PAGES THAT BELONG TO H ALONE are ever pinned.  A large array may still be a heap chunk (glibc raises its
    // mmap threshold up to 32 MB as a process ages), and then its first and last page are shared with its neighbours -- the
    // class of input that ended in GPU memory faults for small matrices.  The page-aligned interior is registered
    // and copied asynchronously; the up to two partial pages at the ends (< 8 KB) are copied synchronously from pageable memory.
    // Needs a contiguous H (ldH == n; anything else takes the synchronous path).
    const bool contiguous = ldH == (h->n > 1 ? h->n : 1);
    const uintptr_t b0 = reinterpret_cast<uintptr_t>(H), b1 = b0 + bytes;
    // (the host's page size, not a hard-coded 4096: with 64 KB pages a 4 KB-aligned interior would still share OS pages: ADVICE r5)
    static const uintptr_t pgmask = []() { const long v = sysconf(_SC_PAGESIZE); return (uintptr_t)(v > 0 ? v : 4096) - 1; }();
    uintptr_t p0 = pin_any ? b0 : ((b0 + pgmask) & ~pgmask), p1 = pin_any ? b1 : (b1 & ~pgmask);
    if (!pin_any && (!contiguous || p1 <= p0 || p1 - p0 < ((size_t)2 << 20))) return mi355kkt_set_H_dense(h, H, ldH);
    // Round 6: ... and never pages of the brk heap, whatever the size (see touches_brk_heap): a registration that is already held for
    // exactly this range stays (it was checked when it was made)
    if (!pin_any && !(h->reg_ptr == reinterpret_cast<const void*>(p0) && h->reg_bytes == (size_t)(p1 - p0)) && touches_brk_heap(p0, p1))
        return mi355kkt_set_H_dense(h, H, ldH);
    const void* rptr = reinterpret_cast<const void*>(p0);
    const size_t rbytes = (size_t)(p1 - p0);
    if (h->reg_ptr != rptr || h->reg_bytes != rbytes) {
        h_unregister(h);
        if (hipHostRegister(const_cast<void*>(rptr), rbytes, hipHostRegisterDefault) != hipSuccess) {
            (void)hipGetLastError();                       // not pinnable: plain synchronous upload
            h->h_pending = false;
            return mi355kkt_set_H_dense(h, H, ldH);
        }
        h->reg_ptr = rptr;
        h->reg_bytes = rbytes;
    }
    if (!h->cst) KKT_HIP_CHECK(hipStreamCreateWithFlags(&h->cst, hipStreamNonBlocking));
    if (!h->ev_h) KKT_HIP_CHECK(hipEventCreateWithFlags(&h->ev_h, hipEventDisableTiming));
    if (!h->H_owned) KKT_HIP_CHECK(DEV_ALLOC(&h->H_owned, sizeof(double) * dmax((size_t)h->n * h->n, 1)));
    KKT_HIP_CHECK(hipStreamSynchronize(h->st));            // nothing on the compute stream may still read the old H
    if (pin_any && !contiguous) {
        KKT_HIP_CHECK(hipMemcpy2DAsync(h->H_owned, sizeof(double) * h->n, H, sizeof(double) * ldH, sizeof(double) * h->n, h->n,
                                       hipMemcpyHostToDevice, h->cst));
    } else {
        char* dst = reinterpret_cast<char*>(h->H_owned);
        if (p0 > b0) KKT_HIP_CHECK(memcpy_sync(dst, H, (size_t)(p0 - b0), hipMemcpyHostToDevice));
        if (b1 > p1) KKT_HIP_CHECK(memcpy_sync(dst + (p1 - b0), reinterpret_cast<const void*>(p1), (size_t)(b1 - p1), hipMemcpyHostToDevice));
        KKT_HIP_CHECK(hipMemcpyAsync(dst + (p0 - b0), rptr, rbytes, hipMemcpyHostToDevice, h->cst));
    }
    KKT_HIP_CHECK(hipEventRecord(h->ev_h, h->cst));
    h->dH = h->H_owned;
    h->ldH = h->n > 1 ? h->n : 1;
    h->hsym_valid = false;
    h->h_pending = true;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_set_H_dense_async"); }
int mi355kkt_set_H_device(mi355kkt_solver* h, const double* dH, int64_t ldH) try {
    if (!h) return MI355KKT_EINVAL;
    if (h->h_pending && h->cst) (void)hipStreamSynchronize(h->cst);
    h->h_pending = false;
    h_unregister(h);                                       // header contract: a pinned host H is released by the next set_H_*
    h->dH = dH;
    h->ldH = ldH;
    h->hsym_valid = false;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_set_H_device"); }
int mi355kkt_set_progress(mi355kkt_solver* h, mi355kkt_progress_fn fn, void* user) try {
    if (!h) return MI355KKT_EINVAL;
    h->progress = fn;
    h->progress_user = user;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_set_progress"); }
int mi355kkt_set_kktreg(mi355kkt_solver* h, double reg) try {
    if (!h || !(reg >= 0.0)) { set_last_error("set_kktreg: reg must be >= 0"); return MI355KKT_EINVAL; }
    h->kktreg = reg;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_set_kktreg"); }

int mi355kkt_set_option(mi355kkt_solver* h, const char* name, double value) try {
    if (!h || !name) { set_last_error("set_option: null argument"); return MI355KKT_EINVAL; }
    if (!strcmp(name, "use_correction")) { h->use_correction = value != 0.0 ? 1 : 0; return 0; }
    if (!strcmp(name, "ldl_refinement")) {
        if (!(value >= 0.0) || value > 16.0) { set_last_error("set_option: ldl_refinement must be in 0..16"); return MI355KKT_EINVAL; }
        h->ldl_refine = (int)value;
        return 0;
    }
    if (!strcmp(name, "qr_refinement")) {
        if (!(value >= 0.0) || value > 16.0) { set_last_error("set_option: qr_refinement must be in 0..16"); return MI355KKT_EINVAL; }
        h->qr_refine = (int)value;
        h->qr_active = false;
        return 0;
    }
    set_last_error("set_option: unknown option '%s'", name);
    return MI355KKT_EINVAL;
} catch (...) { return kkt_catch("mi355kkt_set_option"); }

// {min, max} of |diag(L)|: one workgroup, fixed order
constexpr double QR_REFINE_COND = 1e8;       // (max L_ii / min L_ii)^2 from which the 'qr' mapping refines its solves
constexpr double QR2_COND = 1e10;            // ... and from which it repairs the factor by CholeskyQR2
// out(rows x n) = diag(d) G  (the LP cone's Gs = W^-T G, which the SYRK never materialises)
__global__ void qr_scale_rows_kernel(const double* __restrict__ G, int64_t ldg, const double* __restrict__ d, int rows, int n,
                                     double* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i < rows && j < n) out[i + (int64_t)j * rows] = d[i] * G[i + (int64_t)j * ldg];
}
__global__ __launch_bounds__(256) void diag_minmax_kernel(const double* __restrict__ S, int64_t ld, int n, double* __restrict__ out) {
    __shared__ double lo[256], hi[256];
    double a = 1e300, b = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double v = fabs(S[i + (int64_t)i * ld]);
        a = v < a ? v : a;                                     // (a NaN pivot never gets here: info > 0 is returned first)
        b = v > b ? v : b;
    }
    lo[threadIdx.x] = a;
    hi[threadIdx.x] = b;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) {
            lo[threadIdx.x] = fmin(lo[threadIdx.x], lo[threadIdx.x + s2]);
            hi[threadIdx.x] = fmax(hi[threadIdx.x], hi[threadIdx.x + s2]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = lo[0]; out[1] = hi[0]; }
}

// info word -> host (synchronises the stream)
// the all-CU triangular solves (trsv512.hip) serve this handle's S: dense engine, order 1024 and up, not switched off by the test knob
static bool trsv_wide_wanted(const mi355kkt_solver* h) {
    const char* k = dev_knob("MI355KKT_TRSV_WIDE");
    if (!(k ? atoi(k) != 0 : TRSV_WIDE_DEFAULT)) return false;
    return trsv_wide_rows(h->n, h->num_cus, !(k && atoi(k) == 128)) != 0; // (any order >= 1024; knob value 128: multiples of 128 only --
}                                                                          //  ragged orders then take the round-4 one-sweep kernel: A/B)

static int fetch_info(mi355kkt_solver* h, int* info) {
    KKT_HIP_CHECK(hipMemcpyAsync(h->pw.h_info, h->pw.d_info, sizeof(int), hipMemcpyDeviceToHost, h->st));
    KKT_HIP_CHECK(hipStreamSynchronize(h->st));
    *info = *h->pw.h_info;
    if (*info < 0) {          // the persistent tile Cholesky gave up on a hand-off (a workgroup was starved for ~seconds)
        set_last_error("potrf: tile hand-off timeout (info = %d)", *info);
        return MI355KKT_EHIP;
    }
    return 0;
}

// assemble S = H + [reg I] + Gs' Gs [+ A'A]
// out[r] = beta * out[r] + sum_k v[k] x[ci[k]] over row r of a CSR matrix: one wave per row, fixed-order tree reduction
__global__ __launch_bounds__(256) void csr_mul_kernel(const int64_t* __restrict__ rp, const int* __restrict__ ci,
                                                      const double* __restrict__ v, int nrows, const double* __restrict__ x,
                                                      double* __restrict__ out, double beta) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= nrows) return;
    double acc = 0.0;
    for (int64_t k = rp[r] + lane; k < rp[r + 1]; k += 64) acc = fma(v[k], x[ci[k]], acc);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) out[r] = (beta == 0.0 ? 0.0 : beta * out[r]) + acc;
}
// A x -> Ax (p) with the handle's A, dense or sparse
static int A_mul(mi355kkt_solver* hs, const double* x, double* Ax, double* gwork, hipStream_t st) {
    const int n = hs->n, np = hs->p;
    if (np <= 0) return 0;
    if (hs->A_sparse) {
        hipLaunchKernelGGL(csr_mul_kernel, dim3((np + 3) / 4), dim3(256), 0, st, hs->dArp, hs->dAci, hs->dAv, np, x, Ax, 0.0);
        KKT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    return launch_gemv_n_scaled(hs->dA, hs->ldA, np, n, nullptr, x, Ax, Ax, 1.0, 0.0, gwork, st);
}
// out (n) := beta * out + A' y
static int A_mulT(mi355kkt_solver* hs, const double* y, double* out, double beta, double* gwork, hipStream_t st) {
    const int n = hs->n, np = hs->p;
    if (np <= 0 || n <= 0) return 0;
    if (hs->A_sparse) {
        hipLaunchKernelGGL(csr_mul_kernel, dim3((n + 3) / 4), dim3(256), 0, st, hs->dAcp, hs->dAri, hs->dAvc, n, y, out, beta);
        KKT_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (beta == 0.0) KKT_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(double) * n, st));
    return launch_gemv_t_scaled(hs->dA, hs->ldA, np, n, nullptr, y, hs->dtp, out, gwork, st);
}

__global__ void add_lower_kernel(double* __restrict__ S, int64_t lds, const double* __restrict__ H, int64_t ldh, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;      // column j, rows i >= j
    if (i < n && i >= j) S[i + (int64_t)j * lds] += H[i + (int64_t)j * ldh];
}

static int assemble_S(mi355kkt_solver* h, bool add_AtA) {
    // H still crossing PCIe on the copy stream (set_H_dense_async): S = Gs'Gs first, S += tril(H) once it has landed
    const bool late_H = h->h_pending && h->dH != nullptr;
    const double* Hnow = late_H ? nullptr : h->dH;
    struct Late {
        mi355kkt_solver* h; bool on;
        int add() {
            if (!on) return 0;
            KKT_HIP_CHECK(hipStreamWaitEvent(h->st, h->ev_h, 0));
            if (h->n > 0)
                hipLaunchKernelGGL(add_lower_kernel, dim3((h->n + 255) / 256, h->n), dim3(256), 0, h->st, h->dS, (int64_t)h->n, h->dH,
                                   h->ldH, h->n);
            h->h_pending = false;
            return 0;
        }
    } late{h, late_H};
    if (!h->q.empty() || !h->s.empty()) {
        // Gs = W^-T G once (HBM-bound; 's' rows land in packed storage), then the unscaled SYRK on Gs
        const double zs = 1.0 / std::sqrt(1.0 + h->kktreg);
        if (int e = launch_cone_scale(h->cl, h->dG, h->ldG, h->dGs, h->krows, h->n, h->dW, h->dV, h->dBeta, zs, h->st)) return e;
        if (int e = launch_sdp_scale_pack(h->cl, h->dG, h->ldG, h->dGs, h->krows, h->n, h->dRti, zs, h->st)) return e;
        if (int e = launch_syrk_scaled(h->planS, h->dGs, h->krows, nullptr, h->dS, h->n, Hnow, h->ldH, h->st, &h->ev[6]))
            return e;
    } else if (int e = launch_syrk_scaled(h->planS, h->dG, h->ldG, h->ml > 0 ? h->dW : nullptr, h->dS, h->n, Hnow,
                                          h->ldH, h->st, &h->ev[6]))
        return e;
    if (int e = late.add()) return e;
    if (h->kktreg != 0.0 && h->n > 0) hipLaunchKernelGGL(diag_add_kernel, g1(h->n), dim3(256), 0, h->st, h->dS, (int64_t)h->n, h->n, h->kktreg);
    if (add_AtA && h->p > 0)
        if (int e = launch_syrk_scaled(h->planAtA, h->dA, h->ldA, nullptr, h->dS, h->n, h->dS, h->n, h->st)) return e;
    return 0;
}

int mi355kkt_factor_device(mi355kkt_solver* h, const mi355kkt_scaling* W) try {
    if (!h || !W) { set_last_error("factor: null argument"); return MI355KKT_EINVAL; }
    if (!h->s.empty() && !W->rti) { set_last_error("factor: W.rti missing"); return MI355KKT_EINVAL; }
    if (h->ml > 0 && !W->di) { set_last_error("factor: W.di missing"); return MI355KKT_EINVAL; }
    if (!h->q.empty() && (!W->v || !W->beta)) { set_last_error("factor: W.v / W.beta missing"); return MI355KKT_EINVAL; }
    if (h->sparse) {
        if (int e = bind(h)) return e;
        h->factored = false;
        KKT_HIP_CHECK(hipEventRecord(h->ev[0], h->st));
        if (h->ml > 0) hipLaunchKernelGGL(scaled_copy_kernel, g1(h->ml), dim3(256), 0, h->st, W->di, h->dW, h->ml, 1.0);
        if (h->sp_extra > 0)      // S + A'A mode: the A rows below the cone rows carry unit scaling
            hipLaunchKernelGGL(fill_kernel, g1(h->sp_extra), dim3(256), 0, h->st, h->dW + h->ml, 1.0, (int64_t)h->sp_extra);
        int sinfo = 0;
        if (int e = sparse_engine_factor(h->sp, h->dW, h->st, &sinfo)) return e;
        KKT_HIP_CHECK(hipEventRecord(h->ev[3], h->st));
        KKT_HIP_CHECK(hipStreamSynchronize(h->st));
        (void)hipEventElapsedTime(&h->t_factor, h->ev[0], h->ev[3]);
        h->t_syrk = h->t_potrf = h->t_schur = h->t_syrk_kernel = 0;
        h->firstcall = false;
        // singular S: the caller (the Python mirror, on the first factorisation with p > 0) re-creates the sparse problem with
        // the rows of A appended -- mi355kkt_set_sparse_problem_aug, a new symbolic analysis since the pattern of S grows --
        // and factors again: the S + A'A fallback of misc.py:1433-1447
        if (sinfo < 0) { set_last_error("sparse factor: tile hand-off timeout in the root front (info = %d)", sinfo); return MI355KKT_EHIP; }
        if (sinfo > 0) return sinfo;
        if (h->p > 0) {
            // equality constraints (misc.py:1464-1487, sparse branch): Asct = L^-1 P A' with all p right-hand sides in
            // one pass of the supernodal forward solve (kept in the permuted ordering), K = Asct' Asct, dense Cholesky of K
            if (!h->dA && !h->A_sparse) { set_last_error("factor: A not set"); return MI355KKT_EINVAL; }
            if (!h->dSpWork) KKT_HIP_CHECK(DEV_ALLOC(&h->dSpWork, sizeof(double) * gemv_work_doubles(h->n, h->p)));
            if (h->A_sparse) {
                if (int e = sparse_engine_forward_rows_csr(h->sp, h->dArp, h->dAci, h->dAv, h->p, h->dAsct, h->st)) return e;
            } else if (int e = sparse_engine_forward_rows(h->sp, h->dA, h->ldA, h->p, h->dAsct, h->st)) return e;
            KKT_HIP_CHECK(hipMemsetAsync(h->pw.d_info, 0, sizeof(int), h->st));
            if (int e = launch_syrk_scaled(h->planK, h->dAsct, h->n, nullptr, h->dK, h->p, nullptr, 0, h->st)) return e;
            if (h->kktreg != 0.0) hipLaunchKernelGGL(diag_add_kernel, g1(h->p), dim3(256), 0, h->st, h->dK, (int64_t)h->p, h->p, h->kktreg);
            if (int e = launch_potrf(h->dK, h->p, h->p, h->pw, h->st)) return e;
            int kinfo = 0;
            if (int e = fetch_info(h, &kinfo)) return e;
            if (kinfo > 0) return h->n + kinfo;
        }
        h->factored = true;
        return 0;
    }
    if ((h->cdim > 0 && h->n > 0 && !h->dG) || (h->p > 0 && h->n > 0 && !h->dA)) {
        set_last_error("factor: G / A not set");
        return MI355KKT_EINVAL;
    }
    if (int e = bind(h)) return e;
    if (!h->dS) {   // dense engine state is created on first use (a sparse-mode handle never pays for it)
        const size_t N = (size_t)h->n;
        // all or nothing: dS doubles as the "dense state exists" flag, so it is set last
        double *newS = nullptr, *newwork = nullptr;
        // (p x n products with A -- the refinement of the ldl flavours computes by - A ux with this workspace -- need ceil(n / 256) * p:
        //  more than either of the first two when p is large, the cone rows few and n > 256)
        const size_t work_doubles = dmax(dmax(dmax(gemv_work_doubles(h->cdim, h->n), gemv_work_doubles(h->n, h->p)),
                                              gemv_work_doubles(h->p, h->n)), (size_t)h->cdim + 8);
        if (DEV_ALLOC(&newS, sizeof(double) * dmax(N * N, 1)) != hipSuccess ||
            DEV_ALLOC(&newwork, sizeof(double) * work_doubles) != hipSuccess) {
            if (newS) (void)dev_free(newS);
            set_last_error("factor: out of device memory for the %zu x %zu reduced KKT matrix", N, N);
            return MI355KKT_ENOMEM;
        }
        if (int e = build_syrk_plan(h->planS, h->n, h->krows, h->num_cus)) {
            (void)dev_free(newS);
            (void)dev_free(newwork);
            return e;
        }
        (void)dev_free(h->dwork);
        h->dwork = newwork;
        h->dS = newS;
    }
    h->factored = false;
    const double zscale = 1.0 / std::sqrt(1.0 + h->kktreg);   // K[z,z] = -(1+reg): fold into the row scaling
    RoctxRange rr("mi355kkt factor: scale (W^-T G)");
    KKT_HIP_CHECK(hipEventRecord(h->ev[0], h->st));
    if (!h->q.empty() || !h->s.empty()) {
        // cone path: dW keeps the raw di (the 1/sqrt(1+reg) factor is applied by the scaling kernels)
        if (h->ml > 0) hipLaunchKernelGGL(scaled_copy_kernel, g1(h->ml), dim3(256), 0, h->st, W->di, h->dW, h->ml, 1.0);
        if (!h->q.empty()) {
            KKT_HIP_CHECK(hipMemcpyAsync(h->dV, W->v, sizeof(double) * h->cl.vlen, hipMemcpyDeviceToDevice, h->st));
            KKT_HIP_CHECK(hipMemcpyAsync(h->dBeta, W->beta, sizeof(double) * h->q.size(), hipMemcpyDeviceToDevice, h->st));
            if (int e = cone_layout_set_beta(h->cl, h->dBeta, h->st)) return e;
        }
        if (!h->s.empty())
            KKT_HIP_CHECK(hipMemcpyAsync(h->dRti, W->rti, sizeof(double) * h->cl.rlen, hipMemcpyDeviceToDevice, h->st));
    } else if (h->ml > 0) {
        hipLaunchKernelGGL(scaled_copy_kernel, g1(h->ml), dim3(256), 0, h->st, W->di, h->dW, h->ml, zscale);
    }
    rr.next("mi355kkt factor: assemble S = H + Gs'Gs");
    if (int e = assemble_S(h, h->singular)) return e;
    KKT_HIP_CHECK(hipEventRecord(h->ev[1], h->st));
    rr.next("mi355kkt factor: Cholesky");
    if (int e = launch_potrf(h->dS, h->n, h->n, h->pw, h->st)) return e;
    KKT_HIP_CHECK(hipEventRecord(h->ev[2], h->st));
    int info = 0;
    if (int e = fetch_info(h, &info)) return e;
    if (info > 0 && (h->firstcall || h->kind != MI355KKT_CHOL2) && !h->singular && h->p > 0) {
        // reference misc.py:1433-1447: singular S on the first call -> S += A'A for good.  kkt_chol2 only looks at the first
        // call; kkt_ldl / kkt_ldl2 (pivoted LDL' of the whole matrix, lapack.c:2282) and kkt_chol (QR elimination of A,
        // misc.py:1250-1282) accept every nonsingular KKT matrix at every call.  S + A'A is positive definite exactly when
        // the KKT matrix is nonsingular (x'Sx = 0 and Ax = 0 force x = 0), so for those flavours the switch may happen at
        // any factorisation: the same set of systems, without pivoting.
        h->singular = true;
        if (int e = assemble_S(h, true)) return e;
        if (int e = launch_potrf(h->dS, h->n, h->n, h->pw, h->st)) return e;
        KKT_HIP_CHECK(hipEventRecord(h->ev[2], h->st));
        if (int e = fetch_info(h, &info)) return e;
    }
    h->firstcall = false;
    h->qr_active = false;
    h->qr_extra = 0;
    const bool qr_cones = !h->q.empty() || !h->s.empty();
    const int qr_rows = qr_cones ? h->krows : h->ml;
    // (the repair's triangular solve with `rows` right-hand sides is scalar code: it is for the small and medium problems 'qr' is
    //  chosen for -- up to 4e9 multiply-adds, ~10 ms --, larger ones keep the refinement alone and their speed)
    const bool qr_can = h->qr_refine > 0 && !h->dH && !h->singular && h->kktreg == 0.0 && qr_rows >= h->n && h->n > 0 &&
                        (double)h->n * h->n * qr_rows <= 4e9;
    int qr_passes = 0;
    auto cond_word = [&]() -> int {            // {min, max} of |diag| of what h->dS holds -> h->h_cond (synchronises)
        if (!h->d_cond) {
            KKT_HIP_CHECK(DEV_ALLOC(&h->d_cond, 2 * sizeof(double)));
            KKT_HIP_CHECK(hipHostMalloc(&h->h_cond, 2 * sizeof(double)));
        }
        hipLaunchKernelGGL(diag_minmax_kernel, dim3(1), dim3(256), 0, h->st, h->dS, (int64_t)h->n, h->n, h->d_cond);
        KKT_HIP_CHECK(hipMemcpyAsync(h->h_cond, h->d_cond, 2 * sizeof(double), hipMemcpyDeviceToHost, h->st));
        KKT_HIP_CHECK(hipStreamSynchronize(h->st));
        return 0;
    };
    if (info > 0 && qr_can) {
        // the 'qr' mapping: Gs'Gs is numerically singular where the reference's QR of Gs still works.  Shifted Cholesky of the
        // re-assembled matrix, then two repair passes
        rr.next("mi355kkt factor: shifted Cholesky (CholeskyQR3)");
        if (int e = assemble_S(h, false)) return e;
        if (int e = cond_word()) return e;
        const double nrm2 = (double)h->n * h->h_cond[1];       // trace(S) <= n max S_ii, an upper bound of ||Gs||_2^2
        const double sigma = 11.0 * ((double)qr_rows * h->n + (double)h->n * (h->n + 1)) * 2.220446049250313e-16 * nrm2;
        if (sigma > 0.0 && sigma == sigma && sigma < 1e300) {
            hipLaunchKernelGGL(diag_add_kernel, g1(h->n), dim3(256), 0, h->st, h->dS, (int64_t)h->n, h->n, sigma);
            if (int e = launch_potrf(h->dS, h->n, h->n, h->pw, h->st)) return e;
            KKT_HIP_CHECK(hipEventRecord(h->ev[2], h->st));
            if (int e = fetch_info(h, &info)) return e;
            if (info == 0) qr_passes = 2;
        }
    }
    if (info > 0) return info;
    if (h->qr_refine > 0) {
        // the 'qr' mapping: how ill conditioned is the reduced matrix?  (max L_ii / min L_ii)^2 <= cond(S), one tiny kernel and one
        // more 16-byte read-back per factorisation of this mapping only
        if (int e = cond_word()) return e;
        const double r = h->h_cond[1] / h->h_cond[0], c2 = r * r;
        h->qr_active = qr_passes > 0 || !(c2 < QR_REFINE_COND);          // (also for inf / nan)
        if (qr_passes == 0 && !(c2 < QR2_COND) && qr_can) qr_passes = 1;
        if (qr_passes > 0) {
            rr.next("mi355kkt factor: CholeskyQR repair of the factor");
            const int rows = qr_rows;
            const size_t nn = (size_t)h->n, rr_ = (size_t)rows;
            if (!h->dQt) KKT_HIP_CHECK(DEV_ALLOC(&h->dQt, sizeof(double) * nn * rr_));
            if (!h->dQ) KKT_HIP_CHECK(DEV_ALLOC(&h->dQ, sizeof(double) * nn * rr_));
            const double* Gs = h->dGs;
            int64_t ldgs = h->krows;
            if (!qr_cones) {                                   // the LP cone's Gs is never materialised by the SYRK: form it here
                hipLaunchKernelGGL(qr_scale_rows_kernel, dim3((rows + 255) / 256, h->n), dim3(256), 0, h->st, h->dG, h->ldG, h->dW, rows,
                                   h->n, h->dQ);
                Gs = h->dQ;
                ldgs = rows;
            }
            hipLaunchKernelGGL(transpose_kernel, dim3((h->n + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, h->st, Gs, ldgs, rows, h->n,
                               h->dQt);                        // Q_0' = Gs'
            for (int k = 0; k < qr_passes; ++k) {
                if (!h->dSx[k]) KKT_HIP_CHECK(DEV_ALLOC(&h->dSx[k], sizeof(double) * nn * nn));
                if (!h->pwx_ready[k]) {
                    if (int e = potrf_work_init(h->pwx[k])) return e;
                    h->pwx_ready[k] = true;
                }
                // Q_{k+1}' = L_{k+1}^-1 Q_k'  (L_1 = the factor in h->dS), Q_{k+1} in the SYRK's layout, S = Q'Q, next factor
                const double* Lk = k == 0 ? h->dS : h->dSx[k - 1];
                if (int e = launch_trsm_lower(Lk, h->n, h->n, h->dQt, h->n, rows, 0, h->st)) return e;
                hipLaunchKernelGGL(transpose_kernel, dim3((rows + 31) / 32, (h->n + 31) / 32), dim3(32, 8), 0, h->st, h->dQt,
                                   (int64_t)h->n, h->n, rows, h->dQ);
                if (int e = launch_syrk_scaled(h->planS, h->dQ, rows, nullptr, h->dSx[k], h->n, nullptr, 0, h->st)) return e;
                if (int e = launch_potrf(h->dSx[k], h->n, h->n, h->pwx[k], h->st)) return e;
                KKT_HIP_CHECK(hipMemcpyAsync(h->pwx[k].h_info, h->pwx[k].d_info, sizeof(int), hipMemcpyDeviceToHost, h->st));
                KKT_HIP_CHECK(hipStreamSynchronize(h->st));
                const int infok = *h->pwx[k].h_info;
                if (infok < 0) { set_last_error("potrf: tile hand-off timeout (info = %d)", infok); return MI355KKT_EHIP; }
                if (infok > 0) return infok;
            }
            h->qr_extra = qr_passes;
        }
    }
    rr.next("mi355kkt factor: Schur complement / solve preparation");
    if (h->p > 0) {
        // Asct = L^-1 A'
        hipLaunchKernelGGL(transpose_kernel, dim3((h->n + 31) / 32, (h->p + 31) / 32), dim3(32, 8), 0, h->st, h->dA,
                           h->ldA, h->p, h->n, h->dAsct);
        if (int e = launch_trsm_lower(h->dS, h->n, h->n, h->dAsct, h->n, h->p, 0, h->st)) return e;
        for (int k = 0; k < h->qr_extra; ++k)
            if (int e = launch_trsm_lower(h->dSx[k], h->n, h->n, h->dAsct, h->n, h->p, 0, h->st)) return e;
        // K = Asct' Asct [+ reg I]
        if (int e = launch_syrk_scaled(h->planK, h->dAsct, h->n, nullptr, h->dK, h->p, nullptr, 0, h->st)) return e;
        if (h->kktreg != 0.0) hipLaunchKernelGGL(diag_add_kernel, g1(h->p), dim3(256), 0, h->st, h->dK, (int64_t)h->p, h->p, h->kktreg);
        if (int e = launch_potrf(h->dK, h->p, h->p, h->pw, h->st)) return e;
    }
    // L' into the (otherwise unused) upper triangle of S: the transposed persistent solve streams it coalesced
    if ((h->n + 127) / 128 <= h->num_cus)
        if (int e = launch_mirror_lower(h->dS, h->n, h->n, h->st)) return e;
    // round 6: 512 x 512 inverses of the diagonal blocks for the all-CU triangular solves (trsv512.hip), from the 128 x 128 ones the
    // tile Cholesky of S left (not when K went through the tile kernel after it: they are K's then)
    if (trsv_wide_wanted(h) && h->pw.minv_n == h->n && h->pw.minv_of == h->dS)
        if (int e = launch_block_inverse512(h->dS, h->n, h->n, h->pw, h->st)) return e;
    KKT_HIP_CHECK(hipEventRecord(h->ev[3], h->st));
    if (int e = fetch_info(h, &info)) return e;
    (void)hipEventElapsedTime(&h->t_syrk, h->ev[0], h->ev[1]);
    (void)hipEventElapsedTime(&h->t_potrf, h->ev[1], h->ev[2]);
    (void)hipEventElapsedTime(&h->t_schur, h->ev[2], h->ev[3]);
    (void)hipEventElapsedTime(&h->t_factor, h->ev[0], h->ev[3]);
    (void)hipEventElapsedTime(&h->t_syrk_kernel, h->ev[6], h->ev[7]);
    if (info > 0) return info;
    h->factored = true;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_factor_device"); }

int mi355kkt_factor(mi355kkt_solver* h, const mi355kkt_scaling* W) try {
    if (!h || !W) { set_last_error("factor: null argument"); return MI355KKT_EINVAL; }
    if (h->ml > 0 && !W->di) { set_last_error("factor: W.di missing"); return MI355KKT_EINVAL; }
    if (int e = bind(h)) return e;
    mi355kkt_scaling Wd = {};
    const size_t ml = h->ml, vlen = h->cl.vlen, nq = h->q.size(), rlen = h->cl.rlen;
    if (nq > 0 && (!W->v || !W->beta)) { set_last_error("factor: W.v / W.beta missing"); return MI355KKT_EINVAL; }
    if (rlen > 0 && !W->rti) { set_last_error("factor: W.rti missing"); return MI355KKT_EINVAL; }
    if (ml) memcpy(h->hbuf, W->di, sizeof(double) * ml);
    if (nq) {
        memcpy(h->hbuf + ml, W->v, sizeof(double) * vlen);
        memcpy(h->hbuf + ml + vlen, W->beta, sizeof(double) * nq);
    }
    if (rlen) memcpy(h->hbuf + ml + vlen + nq, W->rti, sizeof(double) * rlen);
    if (ml + vlen + nq + rlen)
        KKT_HIP_CHECK(hipMemcpyAsync(h->dWst, h->hbuf, sizeof(double) * (ml + vlen + nq + rlen), hipMemcpyHostToDevice, h->st));
    if (ml) Wd.di = h->dWst;
    if (nq) {
        Wd.v = h->dWst + ml;
        Wd.beta = h->dWst + ml + vlen;
    }
    if (rlen) Wd.rti = h->dWst + ml + vlen + nq;
    return mi355kkt_factor_device(h, &Wd);
} catch (...) { return kkt_catch("mi355kkt_factor"); }

int mi355kkt_solve_device(mi355kkt_solver* h, double* dx, double* dy, double* dz) try {
    if (!h) return MI355KKT_EINVAL;
    if (!h->factored) { set_last_error("solve: no valid factorisation"); return MI355KKT_EINVAL; }
    if (int e = bind(h)) return e;
    RoctxRange rr("mi355kkt solve");
    hipStream_t st = h->st;
    const int n = h->n, p = h->p, m = h->cdim;
    KKT_HIP_CHECK(hipEventRecord(h->ev[4], st));
    if (h->sparse) {                                                // misc.py:1513-1563, sparse branch
        double* zz = dz;
        if (h->sp_extra > 0) {                                      // the engine's cone space has p more (A) rows: zero tail
            zz = h->dz;
            if (dz != h->dz) KKT_HIP_CHECK(hipMemcpyAsync(h->dz, dz, sizeof(double) * m, hipMemcpyDeviceToDevice, st));
            KKT_HIP_CHECK(hipMemsetAsync(h->dz + m, 0, sizeof(double) * h->sp_extra, st));
        }
        if (int e = sparse_engine_gemv_t(h->sp, h->dW, zz, h->dzs, h->dwork, dx, st)) return e;
        if (h->singular && p > 0)                                   // x += A' by  (misc.py:1527)
            if (int e = A_mulT(h, dy, dx, 1.0, nullptr, st)) return e;
        if (p == 0) {
            if (int e = sparse_engine_solve(h->sp, dx, st)) return e;
        } else {
            // x_p := L^-1 P x;  y := K^-1 (Asct' x_p - y);  x_p -= Asct y;  x := P' L^-T x_p       (misc.py:1528-1558)
            if (int e = sparse_engine_forward(h->sp, dx, nullptr, st)) return e;
            double* xp = h->sp.d_xp;
            hipLaunchKernelGGL(scal_kernel, g1(p), dim3(256), 0, st, dy, p, -1.0);
            if (int e = launch_gemv_t_scaled(h->dAsct, n, n, p, nullptr, xp, h->dtn, dy, nullptr, st)) return e;
            if (int e = launch_trsm_lower(h->dK, p, p, dy, p, 1, 0, st)) return e;
            if (int e = launch_trsm_lower(h->dK, p, p, dy, p, 1, 1, st)) return e;
            if (int e = launch_gemv_n_scaled(h->dAsct, n, n, p, nullptr, dy, xp, xp, -1.0, 1.0, h->dSpWork, st)) return e;
            if (int e = sparse_engine_backward(h->sp, dx, st)) return e;
        }
        if (int e = sparse_engine_gemv_n(h->sp, h->dW, dx, h->dzs, zz, st)) return e;
        if (zz != dz) KKT_HIP_CHECK(hipMemcpyAsync(dz, zz, sizeof(double) * m, hipMemcpyDeviceToDevice, st));
        KKT_HIP_CHECK(hipEventRecord(h->ev[5], st));
        return 0;
    }
    // zs = W^-T bz ;  x += Gs' zs                                     (misc.py:1513, :1524)
    const bool cones = !h->q.empty() || !h->s.empty();
    const bool sdp = !h->s.empty();
    const double zscale = 1.0 / std::sqrt(1.0 + h->kktreg);
    const double* Gmat = cones ? h->dGs : h->dG;
    const int64_t ldGm = cones ? (int64_t)h->krows : h->ldG;
    const double* wvec = cones ? nullptr : h->dW;
    const int mk = cones ? h->krows : m;        // rows of the (packed) scaled constraint space
    // The ldl / ldl2 flavours stand for the reference's pivoted LDL' of the WHOLE 3 x 3 matrix (misc.py:1085-1121, lapack.sytrf):
    // users pick them for ill-conditioned scalings, where the reduced (normal-equations) form loses digits -- the late-iteration
    // test measured a 3 x 3 residual of 3e-9 against the reference's 2e-13 with d spanning 1e-3 .. 1e3.  They therefore get one
    // step of iterative refinement against the 3 x 3 system, written in the scaled space the engine works in:
    //     [H A' Gs'; A 0 0; Gs 0 -I] [ux; uy; w] = [bx; by; zs],   Gs = W^-T G,  zs = W^-T bz,  w = W uz.
    // (mi355kkt_set_option(h, "ldl_refinement", steps), 0 switches it off; not applied with kktreg, whose regularised system is the one to be solved.)
    // Two steps by default: one brings d in 1e-3 .. 1e3 from 3e-9 to 2e-15 (the reference: 2e-13), d in 1e-5 .. 1e5 needs the second.
    const bool ldl_kind = h->kind == MI355KKT_LDL || h->kind == MI355KKT_LDL2;
    const int ref_steps = ldl_kind ? h->ldl_refine : (h->qr_active ? h->qr_refine : 0);
    const bool refine = ref_steps > 0 && h->kktreg == 0.0 && mk > 0 && n > 0;
    double *bx0 = nullptr, *by0 = nullptr, *zs0 = nullptr, *rx = nullptr, *ry = nullptr, *rz = nullptr, *tt = nullptr;
    if (refine) {
        if (!h->dRef) KKT_HIP_CHECK(DEV_ALLOC(&h->dRef, sizeof(double) * (2 * (size_t)n + 2 * (size_t)dmax(p, 1) + 3 * (size_t)mk)));
        bx0 = h->dRef; by0 = bx0 + n; zs0 = by0 + dmax(p, 1); rx = zs0 + mk; ry = rx + n; rz = ry + dmax(p, 1); tt = rz + mk;
        KKT_HIP_CHECK(hipMemcpyAsync(bx0, dx, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
        if (p > 0) KKT_HIP_CHECK(hipMemcpyAsync(by0, dy, sizeof(double) * p, hipMemcpyDeviceToDevice, st));
    }
    if (cones) {
        if (int e = launch_cone_scale(h->cl, dz, m, h->dzs, mk, 1, h->dW, h->dV, h->dBeta, zscale, st)) return e;
        if (int e = launch_sdp_scale_pack(h->cl, dz, m, h->dzs, mk, 1, h->dRti, zscale, st)) return e;
        if (int e = launch_gemv_t_scaled(Gmat, ldGm, mk, n, nullptr, h->dzs, h->dzs, dx, h->dwork, st)) return e;
    } else if (int e = launch_gemv_t_scaled(h->dG, h->ldG, m, n, h->dW, dz, h->dzs, dx, h->dwork, st))
        return e;
    if (refine) KKT_HIP_CHECK(hipMemcpyAsync(zs0, h->dzs, sizeof(double) * mk, hipMemcpyDeviceToDevice, st));
    // triangular solves with L: one persistent launch each when every 128-block can own a resident workgroup
    const bool persistent = (n + 127) / 128 <= h->num_cus;      // (larger orders: the multi-kernel blocked solve)
    // round 4: two pipelined sweeps (trsv_pair_kernel) when the tile Cholesky left the 128 x 128 inverses of THIS factor, the order
    // is a multiple of 128 and both chains fit the device
    const bool have_minv = h->pw.minv_n == n && h->pw.minv_of == h->dS;
    const char* pk = dev_knob("MI355KKT_TRSV_PAIR");
    const bool pair = persistent && have_minv && n % 128 == 0 && n >= 256 && 2 * (n / 128) <= h->num_cus && (pk ? atoi(pk) != 0 : TRSV_PAIR_DEFAULT);
    const int wide_rows = (trsv_wide_wanted(h) && h->pw.m512_n == n && h->pw.m512_of == h->dS) ? trsv_wide_rows(n, h->num_cus, true) : 0;
    auto tri_solve1 = [&](int trans, double* xv) -> int {
        if (wide_rows) return launch_trsv_wide(h->dS, n, n, xv, trans, ++h->epoch, h->derr, st, h->pw, wide_rows, h->num_cus);
        if (pair) return launch_trsv_pair(h->dS, n, n, xv, trans, ++h->epoch, h->derr, st, h->dgran, h->pw.d_minv);
        if (persistent) return launch_trsv_persistent(h->dS, n, n, xv, trans, ++h->epoch, h->derr, st, h->dgran,
                                                      (h->pw.minv_n == n && h->pw.minv_of == h->dS) ? h->pw.d_minv : nullptr);
        return launch_trsm_lower(h->dS, n, n, xv, n, 1, trans, st);
    };
    // S = L1 L2 [L3] ([L3'] L2') L1' after a CholeskyQR repair (the 'qr' mapping, ill-conditioned Gs): the factors in order forward,
    // in reverse order backward -- the extra ones through the blocked substitution (this mode buys accuracy, not time)
    auto tri_solve = [&](int trans, double* xv) -> int {
        if (h->qr_extra == 0) return tri_solve1(trans, xv);
        if (!trans) {
            if (int e = tri_solve1(0, xv)) return e;
            for (int k = 0; k < h->qr_extra; ++k)
                if (int e = launch_trsm_lower(h->dSx[k], n, n, xv, n, 1, 0, st)) return e;
            return 0;
        }
        for (int k = h->qr_extra - 1; k >= 0; --k)
            if (int e = launch_trsm_lower(h->dSx[k], n, n, xv, n, 1, 1, st)) return e;
        return tri_solve1(1, xv);
    };
    // the reduced system: xv = bx + Gs' zs on entry, (ux, uy) on exit                       (misc.py:1527-1558)
    auto reduced_solve = [&](double* xv, double* yv) -> int {
        if (h->singular && p > 0)                                   // x += A' by  (:1527)
            if (int e = launch_gemv_t_scaled(h->dA, h->ldA, p, n, nullptr, yv, h->dtp, xv, nullptr, st)) return e;
        if (int e = tri_solve(0, xv)) return e;                                         // :1529
        if (p > 0) {
            // y := K^-1 (Asct' x - y)                                     (:1541-1543)
            hipLaunchKernelGGL(scal_kernel, g1(p), dim3(256), 0, st, yv, p, -1.0);
            if (int e = launch_gemv_t_scaled(h->dAsct, n, n, p, nullptr, xv, h->dtn, yv, nullptr, st)) return e;
            if (int e = launch_trsm_lower(h->dK, p, p, yv, p, 1, 0, st)) return e;
            if (int e = launch_trsm_lower(h->dK, p, p, yv, p, 1, 1, st)) return e;
            // x := x - Asct y                                             (:1553)
            if (int e = launch_gemv_n_scaled(h->dAsct, n, n, p, nullptr, yv, xv, xv, -1.0, 1.0, h->dwork, st)) return e;
        }
        return tri_solve(1, xv);                                                        // :1555
    };
    if (int e = reduced_solve(dx, dy)) return e;
    // w := Gs x - zs   (/ sqrt(1+reg) when the z-block pivot is -(1+reg))       (:1563); packed space: in place for 's' cones
    double* wv = sdp ? h->dzs : dz;
    if (int e = launch_gemv_n_scaled(Gmat, ldGm, mk, n, wvec, dx, h->dzs, wv, zscale, -zscale, h->dwork, st)) return e;
    for (int rstep = 0; refine && rstep < ref_steps; ++rstep) {
        // residual of the 3 x 3 system in the scaled space
        //   rz = zs0 - Gs ux + w
        if (int e = launch_gemv_n_scaled(Gmat, ldGm, mk, n, wvec, dx, zs0, tt, 1.0, -1.0, h->dwork, st)) return e;     // tt = Gs ux - zs0
        hipLaunchKernelGGL(add3_kernel, g1(mk), dim3(256), 0, st, rz, wv, tt, -1.0, mk);                                // rz = w - tt
        //   rx = bx0 - H ux - A' uy - Gs' w
        if (h->dH) {
            if (int e = ensure_hsym(h)) return e;
            if (int e = ensure_gemv_work(h)) return e;              // (workspace sized for an n x n product)
            if (int e = launch_gemv_n_scaled(h->dHsym, n, n, n, nullptr, dx, bx0, rx, -1.0, 1.0, h->dIpmWork, st)) return e;
        } else {
            KKT_HIP_CHECK(hipMemcpyAsync(rx, bx0, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
        }
        hipLaunchKernelGGL(mul_kernel, g1(mk), dim3(256), 0, st, tt, wv, wvec, -1.0, mk);                              // tt = -(di .*) w
        if (int e = launch_gemv_t_scaled(Gmat, ldGm, mk, n, nullptr, tt, tt, rx, h->dwork, st)) return e;               // rx += G' tt
        if (p > 0) {
            hipLaunchKernelGGL(mul_kernel, g1(p), dim3(256), 0, st, ry, dy, (const double*)nullptr, -1.0, p);          // ry = -uy (scratch)
            if (int e = launch_gemv_t_scaled(h->dA, h->ldA, p, n, nullptr, ry, h->dtp, rx, nullptr, st)) return e;      // rx -= A' uy
            //   ry = by0 - A ux
            if (int e = launch_gemv_n_scaled(h->dA, h->ldA, p, n, nullptr, dx, by0, ry, -1.0, 1.0, h->dwork, st)) return e;
        }
        // correction: the same reduced solve with (rx, ry, rz)
        hipLaunchKernelGGL(mul_kernel, g1(mk), dim3(256), 0, st, tt, rz, wvec, 1.0, mk);                               // tt = (di .*) rz
        if (int e = launch_gemv_t_scaled(Gmat, ldGm, mk, n, nullptr, tt, tt, rx, h->dwork, st)) return e;               // rx += Gs' rz
        if (int e = reduced_solve(rx, ry)) return e;
        if (int e = launch_gemv_n_scaled(Gmat, ldGm, mk, n, wvec, rx, rz, tt, 1.0, -1.0, h->dwork, st)) return e;       // dw = Gs dx - rz
        hipLaunchKernelGGL(add3_kernel, g1(n), dim3(256), 0, st, dx, dx, rx, 1.0, n);
        if (p > 0) hipLaunchKernelGGL(add3_kernel, g1(p), dim3(256), 0, st, dy, dy, ry, 1.0, p);
        hipLaunchKernelGGL(add3_kernel, g1(mk), dim3(256), 0, st, wv, wv, tt, 1.0, mk);
    }
    if (sdp) {
        // the l/q rows are copied and the 's' blocks unpacked (lower triangles) into z
        if (h->cl.lq_rows > 0)
            KKT_HIP_CHECK(hipMemcpyAsync(dz, h->dzs, sizeof(double) * h->cl.lq_rows, hipMemcpyDeviceToDevice, st));
        if (int e = launch_sdp_unpack(h->cl, h->dzs, dz, st)) return e;
    }
    KKT_HIP_CHECK(hipEventRecord(h->ev[5], st));
    return 0;
} catch (...) { return kkt_catch("mi355kkt_solve_device"); }

static int check_handoff(mi355kkt_solver* h) {   // stream must be idle
    if (*h->herr) {
        *h->herr = 0;
        (void)memset_sync(h->derr, 0, sizeof(int));
        set_last_error("persistent triangular solve: hand-off timeout (a workgroup was not co-resident?)");
        return MI355KKT_EHIP;
    }
    return 0;
}

// for the device-resident loops: the persistent triangular solves report a hand-off timeout only through derr; read it
// back with the per-iteration word (the stream has just been synchronised) instead of iterating on garbage
static int loop_check_handoff(mi355kkt_solver* h) {
    if (!h->derr) return 0;
    KKT_HIP_CHECK(memcpy_sync(h->herr, h->derr, sizeof(int), hipMemcpyDeviceToHost));
    return check_handoff(h);
}

int mi355kkt_sync(mi355kkt_solver* h) try {
    if (!h) return MI355KKT_EINVAL;
    KKT_HIP_CHECK(hipMemcpyAsync(h->herr, h->derr, sizeof(int), hipMemcpyDeviceToHost, h->st));
    KKT_HIP_CHECK(hipStreamSynchronize(h->st));
    return check_handoff(h);
} catch (...) { return kkt_catch("mi355kkt_sync"); }

int mi355kkt_solve(mi355kkt_solver* h, double* x, double* y, double* z) try {
    if (!h) return MI355KKT_EINVAL;
    if (!h->factored) { set_last_error("solve: no valid factorisation"); return MI355KKT_EINVAL; }
    if (int e = bind(h)) return e;
    const size_t n = h->n, p = h->p, m = h->cdim;
    if ((n && !x) || (p && !y) || (m && !z)) { set_last_error("solve: null vector"); return MI355KKT_EINVAL; }
    double* hb = h->hbuf;
    if (n) memcpy(hb, x, sizeof(double) * n);
    if (p) memcpy(hb + n, y, sizeof(double) * p);
    if (m) memcpy(hb + n + p, z, sizeof(double) * m);
    if (n) KKT_HIP_CHECK(hipMemcpyAsync(h->dx, hb, sizeof(double) * n, hipMemcpyHostToDevice, h->st));
    if (p) KKT_HIP_CHECK(hipMemcpyAsync(h->dy, hb + n, sizeof(double) * p, hipMemcpyHostToDevice, h->st));
    if (m) KKT_HIP_CHECK(hipMemcpyAsync(h->dz, hb + n + p, sizeof(double) * m, hipMemcpyHostToDevice, h->st));
    if (int e = mi355kkt_solve_device(h, h->dx, h->dy, h->dz)) return e;
    if (n) KKT_HIP_CHECK(hipMemcpyAsync(hb, h->dx, sizeof(double) * n, hipMemcpyDeviceToHost, h->st));
    if (p) KKT_HIP_CHECK(hipMemcpyAsync(hb + n, h->dy, sizeof(double) * p, hipMemcpyDeviceToHost, h->st));
    if (m) KKT_HIP_CHECK(hipMemcpyAsync(hb + n + p, h->dz, sizeof(double) * m, hipMemcpyDeviceToHost, h->st));
    KKT_HIP_CHECK(hipMemcpyAsync(h->herr, h->derr, sizeof(int), hipMemcpyDeviceToHost, h->st));
    KKT_HIP_CHECK(hipStreamSynchronize(h->st));
    if (int e = check_handoff(h)) return e;
    if (n) memcpy(x, hb, sizeof(double) * n);
    if (p) memcpy(y, hb + n, sizeof(double) * p);
    if (m) memcpy(z, hb + n + p, sizeof(double) * m);
    return 0;
} catch (...) { return kkt_catch("mi355kkt_solve"); }

int mi355kkt_is_singular_mode(const mi355kkt_solver* h) { return (h && h->singular) ? 1 : 0; }

int mi355kkt_get_timings(mi355kkt_solver* h, float* out, int n) try {
    if (!h || !out) return 0;
    (void)hipStreamSynchronize(h->st);
    if (hipEventQuery(h->ev[5]) == hipSuccess && hipEventQuery(h->ev[4]) == hipSuccess)
        (void)hipEventElapsedTime(&h->t_solve, h->ev[4], h->ev[5]);
    // (events that were never recorded -- timings() before the first solve -- make hipEventElapsedTime fail with "invalid resource
    //  handle"; that must not stay behind as the runtime's sticky last error: the next launch check would report it.  Found in round 4.)
    (void)hipGetLastError();
    const float v[6] = {h->t_syrk, h->t_potrf, h->t_schur, h->t_factor, h->t_solve, h->t_syrk_kernel};
    int k = 0;
    for (; k < n && k < 6; ++k) out[k] = v[k];
    return k;
} catch (...) { return kkt_catch("mi355kkt_get_timings"); }

int mi355kkt_get_factor(mi355kkt_solver* h, double* L, int64_t ldL) try {
    if (!h || !L || ldL < h->n) return MI355KKT_EINVAL;
    if (int e = bind(h)) return e;
    KKT_HIP_CHECK(hipStreamSynchronize(h->st));
    if (h->n > 0)
        KKT_HIP_CHECK(memcpy2d_sync(L, sizeof(double) * ldL, h->dS, sizeof(double) * h->n, sizeof(double) * h->n, h->n,
                                  hipMemcpyDeviceToHost));
    return 0;
} catch (...) { return kkt_catch("mi355kkt_get_factor"); }

// ---- batched LP-cone QP engine (BASELINE config 5: many independent small dense problems) -------------
}  // extern "C"

struct mi355kkt_batch {
    int device = 0, nbatch = 0, n = 0, ml = 0, num_cus = 256;
    // equality constraints A_b x = b_b (p rows each): Asct_b = L_b^-1 A_b', K_b = Asct_b' Asct_b per problem (misc.py:1464-1487)
    int p = 0;
    double *dA = nullptr, *dAsct = nullptr, *dK = nullptr, *dy = nullptr, *dtp = nullptr;
    SyrkPlan planK, planAtA;
    PotrfWork pwK;
    bool singular = false, firstcall = true;      // S + A'A mode, decided for the whole batch at the first factorisation
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {};
    double *dG = nullptr, *dH = nullptr, *dS = nullptr, *dW = nullptr;
    double *dx = nullptr, *dz = nullptr, *dzs = nullptr, *dwork = nullptr, *dt1 = nullptr, *dt2 = nullptr;
    bool hasH = false;
    SyrkPlan plan;
    PotrfWork pw;
    float t_factor = 0;
    bool defer_sync = false;          // set by the device-resident loop: the public calls then only enqueue
    IpmWork ipm;                      // allocated on the first mi355kkt_batch_coneqp call
    // second-order cones (mi355kkt_batch_create_cones): dims = {'l': nl, 'q': q} for every problem; `ml` above is then the
    // number of ROWS of G (cdim = nl + sum(q)), which is all the LP-cone code paths need to know.  Gs_b = W_b^-T G_b is
    // materialised (the cone transform is not diagonal), like the single-problem engine does.
    int nl = 0, sumq = 0;
    std::vector<int> q;
    int *d_qoff = nullptr, *d_qdim = nullptr, *d_large = nullptr;   // cone offsets / dimensions / ids of the cones > 32 rows
    int nlarge = 0;
    double *dGs = nullptr, *dV = nullptr, *dBeta = nullptr;
    bool w_set = false;               // v, beta of the current factorisation are in dV, dBeta
    QpWork qp;                        // state of the device-resident loop with cones (coneqp_ipm.hip, one workgroup per problem)
    int use_correction = 1;           // options['use_correction'] of solvers.coneqp (coneprog.py:1781), mi355kkt_batch_set_option
};

extern "C" {

int mi355kkt_batch_set_option(mi355kkt_batch* b, const char* name, double value) try {
    if (!b || !name) { set_last_error("batch_set_option: null argument"); return MI355KKT_EINVAL; }
    if (!strcmp(name, "use_correction")) { b->use_correction = value != 0.0 ? 1 : 0; return 0; }
    set_last_error("batch_set_option: unknown option '%s'", name);
    return MI355KKT_EINVAL;
} catch (...) { return kkt_catch("mi355kkt_batch_set_option"); }

int mi355kkt_batch_create(mi355kkt_batch** out, int device, int nbatch, int n, int ml) try {
    return mi355kkt_batch_create_eq(out, device, nbatch, n, ml, 0);
} catch (...) { return kkt_catch("mi355kkt_batch_create"); }

/* The same with p equality constraints per problem (A_b: p x n, set with mi355kkt_batch_set_A). */
int mi355kkt_batch_create_eq(mi355kkt_batch** out, int device, int nbatch, int n, int ml, int p) try {
    if (!out || nbatch < 1 || n < 1 || ml < 0 || p < 0 || p > n) { set_last_error("batch_create: invalid argument"); return MI355KKT_EINVAL; }
    if (mi355kkt_device_count() <= device) { set_last_error("batch_create: HIP device %d not available", device); return MI355KKT_EHIP; }
    mi355kkt_batch* b = new mi355kkt_batch();
    b->device = device; b->nbatch = nbatch; b->n = n; b->ml = ml; b->nl = ml; b->p = p;
    auto fail = [&](int code) { mi355kkt_batch_destroy(b); return code; };
    if (hipSetDevice(device) != hipSuccess) return fail(MI355KKT_EHIP);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(MI355KKT_EHIP);
    b->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&b->st, hipStreamNonBlocking) != hipSuccess) return fail(MI355KKT_EHIP);
    for (auto& e : b->ev) if (hipEventCreate(&e) != hipSuccess) return fail(MI355KKT_EHIP);
    const size_t B = nbatch, N = n, M = ml;
    auto alloc = [&](double** p, size_t d) { return DEV_ALLOC(p, sizeof(double) * (d ? d : 1)) == hipSuccess ? 0 : MI355KKT_ENOMEM; };
    int rc;
    if ((rc = alloc(&b->dG, B * M * N))) return fail(rc);
    if ((rc = alloc(&b->dH, B * N * N))) return fail(rc);
    if ((rc = alloc(&b->dS, B * N * N))) return fail(rc);
    if ((rc = alloc(&b->dW, B * M))) return fail(rc);
    if ((rc = alloc(&b->dx, B * N))) return fail(rc);
    if ((rc = alloc(&b->dz, B * M))) return fail(rc);
    if ((rc = alloc(&b->dzs, B * M))) return fail(rc);
    if ((rc = alloc(&b->dwork, B * dmax(dmax(gemv_work_doubles(ml, n), gemv_work_doubles(n, n)), gemv_nt_work_doubles(ml, n))))) return fail(rc);
    if ((rc = alloc(&b->dt1, B * dmax(N, M)))) return fail(rc);
    if ((rc = alloc(&b->dt2, B * dmax(N, M)))) return fail(rc);
    if ((rc = potrf_work_init_batched(b->pw, nbatch))) return fail(rc);
    if ((rc = build_syrk_plan(b->plan, n, ml, b->num_cus, /*allow_split=*/false))) return fail(rc);
    if (p > 0) {
        const size_t Pq = p;
        if ((rc = alloc(&b->dA, B * Pq * N))) return fail(rc);
        if ((rc = alloc(&b->dAsct, B * N * Pq))) return fail(rc);
        if ((rc = alloc(&b->dK, B * Pq * Pq))) return fail(rc);
        if ((rc = alloc(&b->dy, B * Pq))) return fail(rc);
        if ((rc = alloc(&b->dtp, B * dmax(Pq, N)))) return fail(rc);
        if ((rc = potrf_work_init_batched(b->pwK, nbatch))) return fail(rc);
        if ((rc = build_syrk_plan(b->planK, p, n, b->num_cus, false))) return fail(rc);
        if ((rc = build_syrk_plan(b->planAtA, n, p, b->num_cus, false))) return fail(rc);
    }
    *out = b;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_batch_create_eq"); }

/* The same for problems with dims = {'l': nl, 'q': q[0..nq)}: G_b is cdim x n, cdim = nl + sum(q) (see include/mi355kkt.h). */
int mi355kkt_batch_create_cones(mi355kkt_batch** out, int device, int nbatch, int n, int nl, int nq, const int* q, int p) try {
    if (!out || nl < 0 || nq < 0 || (nq > 0 && !q)) { set_last_error("batch_create_cones: invalid argument"); return MI355KKT_EINVAL; }
    int64_t cdim = nl;
    for (int k = 0; k < nq; ++k) {
        if (q[k] < 1) { set_last_error("batch_create_cones: cone dimensions must be positive"); return MI355KKT_EINVAL; }
        cdim += q[k];
    }
    if (cdim > INT_MAX) { set_last_error("batch_create_cones: too many cone rows"); return MI355KKT_EINVAL; }
    if (int e = mi355kkt_batch_create_eq(out, device, nbatch, n, (int)cdim, p)) return e;
    mi355kkt_batch* b = *out;
    b->nl = nl;
    if (nq == 0) return 0;
    auto fail = [&](int code) { mi355kkt_batch_destroy(b); *out = nullptr; return code; };
    b->q.assign(q, q + nq);
    b->sumq = (int)(cdim - nl);
    std::vector<int> hq(2 * (size_t)nq);
    int off = nl;
    for (int k = 0; k < nq; ++k) { hq[k] = off; hq[nq + k] = q[k]; off += q[k]; }
    for (int k = 0; k < nq; ++k)
        if (q[k] > 32) hq.push_back(k);
    b->nlarge = (int)hq.size() - 2 * nq;
    if (DEV_ALLOC(&b->d_qoff, sizeof(int) * hq.size()) != hipSuccess) return fail(MI355KKT_ENOMEM);
    b->d_qdim = b->d_qoff + nq;
    b->d_large = b->d_qoff + 2 * nq;
    if (memcpy_sync(b->d_qoff, hq.data(), sizeof(int) * hq.size(), hipMemcpyHostToDevice) != hipSuccess) return fail(MI355KKT_EHIP);
    const size_t B = nbatch;
    if (DEV_ALLOC(&b->dGs, sizeof(double) * B * (size_t)cdim * n) != hipSuccess) return fail(MI355KKT_ENOMEM);
    if (DEV_ALLOC(&b->dV, sizeof(double) * B * b->sumq) != hipSuccess) return fail(MI355KKT_ENOMEM);
    if (DEV_ALLOC(&b->dBeta, sizeof(double) * B * nq) != hipSuccess) return fail(MI355KKT_ENOMEM);
    return 0;
} catch (...) { return kkt_catch("mi355kkt_batch_create_cones"); }

void mi355kkt_batch_destroy(mi355kkt_batch* b) try {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->st) (void)hipStreamSynchronize(b->st);
    if (b->d_qoff) (void)dev_free(b->d_qoff);
    { double* cb[] = {b->dGs, b->dV, b->dBeta}; for (double* p : cb) if (p) (void)dev_free(p); }
    qp_free(b->qp);
    double* bufs[] = {b->dG, b->dH, b->dS, b->dW, b->dx, b->dz, b->dzs, b->dwork, b->dt1, b->dt2, b->dA, b->dAsct, b->dK, b->dy, b->dtp};
    for (double* p : bufs) if (p) (void)dev_free(p);
    ipm_free(b->ipm);
    potrf_work_free(b->pw);
    potrf_work_free(b->pwK);
    free_syrk_plan(b->plan);
    free_syrk_plan(b->planK);
    free_syrk_plan(b->planAtA);
    for (auto& e : b->ev) if (e) (void)hipEventDestroy(e);
    if (b->st) (void)hipStreamDestroy(b->st);
    delete b;
} catch (...) { (void)kkt_catch("mi355kkt_batch_destroy"); }

/* G: nbatch blocks of ml x n (column-major, contiguous); H: nbatch blocks of n x n (lower triangle used) or NULL.
 * is_device != 0: the pointers are device pointers (copied device-to-device). */
int mi355kkt_batch_set_problem(mi355kkt_batch* b, const double* G, const double* H, int is_device) try {
    if (!b || !G) { set_last_error("batch_set_problem: null argument"); return MI355KKT_EINVAL; }
    KKT_HIP_CHECK(hipSetDevice(b->device));
    const hipMemcpyKind kind = is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const size_t B = b->nbatch, N = b->n, M = b->ml;
    if (M) KKT_HIP_CHECK(memcpy_sync(b->dG, G, sizeof(double) * B * M * N, kind));
    b->hasH = (H != nullptr);
    if (H) {
        KKT_HIP_CHECK(memcpy_sync(b->dH, H, sizeof(double) * B * N * N, kind));
        // only tril(H) is meaningful on input; mirror it so that H x is a plain product (SYRK reads tril only)
        hipLaunchKernelGGL(symmetrize_kernel, dim3((b->n + 15) / 16, (b->n + 15) / 16, b->nbatch), dim3(16, 16), 0, b->st,
                           b->dH, b->n, (int64_t)(N * N));
        KKT_HIP_CHECK(hipStreamSynchronize(b->st));
    }
    return 0;
} catch (...) { return kkt_catch("mi355kkt_batch_set_problem"); }

/* A: nbatch blocks of p x n (column-major, contiguous). */
int mi355kkt_batch_set_A(mi355kkt_batch* b, const double* A, int is_device) try {
    if (!b || (b->p > 0 && !A)) { set_last_error("batch_set_A: null argument"); return MI355KKT_EINVAL; }
    if (b->p == 0) return 0;
    KKT_HIP_CHECK(hipSetDevice(b->device));
    KKT_HIP_CHECK(memcpy_sync(b->dA, A, sizeof(double) * (size_t)b->nbatch * b->p * b->n, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    b->singular = false;
    b->firstcall = true;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_batch_set_A"); }

// Ax[b] = A_b x_b, ATy[b] = A_b' y_b (device arrays; the fA of coneprog.py:2170-2186)
static int batch_a_products(mi355kkt_batch* b, const double* dx, const double* dy, double* Ax, double* ATy) {
    if (b->p <= 0) return 0;
    const size_t B = b->nbatch, N = b->n, Pq = b->p;
    const int64_t sA = (int64_t)(Pq * N);
    if (Ax)
        if (int e = launch_gemv_n_scaled(b->dA, (int64_t)Pq, b->p, b->n, nullptr, dx, Ax, Ax, 1.0, 0.0, b->dwork, b->st, b->nbatch, sA)) return e;
    if (ATy) {
        KKT_HIP_CHECK(hipMemsetAsync(ATy, 0, sizeof(double) * B * N, b->st));
        if (int e = launch_gemv_t_scaled(b->dA, (int64_t)Pq, b->p, b->n, nullptr, dy, b->dtp, ATy, nullptr, b->st, b->nbatch, sA)) return e;
    }
    return 0;
}

/* The residual products of the IPM loop for every problem (reference coneprog.py:2170-2186, fP / fG):
 * Gx[b] = G_b x_b,  GTz[b] = G_b' z_b,  Hx[b] = H_b x_b.  Any output may be NULL.  Host or device arrays. */
int mi355kkt_batch_products(mi355kkt_batch* b, const double* x, const double* z, double* Gx, double* GTz, double* Hx,
                            int is_device) try {
    if (!b) return MI355KKT_EINVAL;
    KKT_HIP_CHECK(hipSetDevice(b->device));
    const size_t B = b->nbatch, N = b->n, M = b->ml;
    const int64_t sG = (int64_t)(M * N), sH = (int64_t)(N * N);
    const double *dx = x, *dz = z;
    if (!is_device) {
        if (x) { KKT_HIP_CHECK(hipMemcpyAsync(b->dx, x, sizeof(double) * B * N, hipMemcpyHostToDevice, b->st)); dx = b->dx; }
        if (z && M) { KKT_HIP_CHECK(hipMemcpyAsync(b->dz, z, sizeof(double) * B * M, hipMemcpyHostToDevice, b->st)); dz = b->dz; }
    }
    auto out_dev = [&](double* user, double* scratch) { return is_device ? user : scratch; };
    // both residual products of an interior-point iteration in one pass over G (device-resident callers)
    if (is_device && Gx && GTz && M && N && dx && dz) {
        if (int e = launch_gemv_nt_fused(b->dG, (int64_t)M, b->ml, b->n, dx, dz, Gx, GTz, b->dwork, b->st, b->nbatch, sG)) return e;
        Gx = nullptr;
        GTz = nullptr;
    }
    if (Gx && M) {
        double* o = out_dev(Gx, b->dzs);
        if (int e = launch_gemv_n_scaled(b->dG, (int64_t)M, b->ml, b->n, nullptr, dx, o, o, 1.0, 0.0, b->dwork, b->st, b->nbatch, sG)) return e;
        if (!is_device) KKT_HIP_CHECK(hipMemcpyAsync(Gx, o, sizeof(double) * B * M, hipMemcpyDeviceToHost, b->st));
    }
    if (GTz) {
        double* o = out_dev(GTz, b->dt1);
        KKT_HIP_CHECK(hipMemsetAsync(o, 0, sizeof(double) * B * N, b->st));
        if (M)
            if (int e = launch_gemv_t_scaled(b->dG, (int64_t)M, b->ml, b->n, nullptr, dz, b->dt2, o, nullptr, b->st, b->nbatch, sG)) return e;
        if (!is_device) KKT_HIP_CHECK(hipMemcpyAsync(GTz, o, sizeof(double) * B * N, hipMemcpyDeviceToHost, b->st));
    }
    if (Hx) {
        double* o = out_dev(Hx, b->dt2);
        if (b->hasH) {
            // dt2 may be in use as gemv_t scratch above: order on the stream makes that safe
            if (int e = launch_gemv_n_scaled(b->dH, (int64_t)N, b->n, b->n, nullptr, dx, o, o, 1.0, 0.0, b->dwork, b->st, b->nbatch, sH)) return e;
        } else {
            KKT_HIP_CHECK(hipMemsetAsync(o, 0, sizeof(double) * B * N, b->st));
        }
        if (!is_device) KKT_HIP_CHECK(hipMemcpyAsync(Hx, o, sizeof(double) * B * N, hipMemcpyDeviceToHost, b->st));
    }
    if (!b->defer_sync) KKT_HIP_CHECK(hipStreamSynchronize(b->st));
    return 0;
} catch (...) { return kkt_catch("mi355kkt_batch_products"); }

/* di: [nbatch][ml]; info: [nbatch] (host), 0 or the failing pivot of that problem.  Returns 0 or <0. */
int mi355kkt_batch_factor(mi355kkt_batch* b, const double* di, int is_device, int* info) try {
    if (!b || (!di && b->ml)) { set_last_error("batch_factor: null argument"); return MI355KKT_EINVAL; }
    KKT_HIP_CHECK(hipSetDevice(b->device));
    const size_t B = b->nbatch, N = b->n, M = b->ml;
    KKT_HIP_CHECK(hipEventRecord(b->ev[0], b->st));
    if (M && di != b->dW)
        KKT_HIP_CHECK(hipMemcpyAsync(b->dW, di, sizeof(double) * B * M, is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, b->st));
    BatchStrides bs;
    bs.a = (int64_t)(M * N); bs.b = (int64_t)M; bs.c = (int64_t)(N * N); bs.d = (int64_t)(N * N);
    const double* Gk = b->dG;
    const double* wk = M ? b->dW : nullptr;
    if (!b->q.empty()) {              // second-order cones: Gs_b = W_b^-T G_b (misc.py:1271), then the plain SYRK
        if (!b->w_set) { set_last_error("batch_factor: the batch has second-order cones, use mi355kkt_batch_factor_cones"); return MI355KKT_EINVAL; }
        if (int e = launch_batch_cone_scale(b->dG, (int64_t)M, (int64_t)(M * N), b->dGs, (int64_t)M, (int64_t)(M * N), b->n, b->nbatch,
                                            b->ml, b->nl, (int)b->q.size(), b->sumq, b->d_qoff, b->d_qdim, b->d_large, b->nlarge, b->dW, b->dV, b->dBeta, b->st))
            return e;
        Gk = b->dGs;
        wk = nullptr;
    }
    if (int e = launch_syrk_scaled(b->plan, Gk, M ? (int64_t)M : 1, wk, b->dS, (int64_t)N,
                                   b->hasH ? b->dH : nullptr, (int64_t)N, b->st, nullptr, b->nbatch, bs))
        return e;
    if (b->singular && b->p > 0) {      // S + A'A mode (misc.py:1433-1447), decided at the first factorisation
        BatchStrides ba;
        ba.a = (int64_t)((size_t)b->p * N); ba.b = 0; ba.c = (int64_t)(N * N); ba.d = (int64_t)(N * N);
        if (int e = launch_syrk_scaled(b->planAtA, b->dA, (int64_t)b->p, nullptr, b->dS, (int64_t)N, b->dS, (int64_t)N, b->st, nullptr,
                                       b->nbatch, ba))
            return e;
    }
    if (int e = launch_potrf_batched(b->dS, (int64_t)N, b->n, b->nbatch, (int64_t)(N * N), b->pw, b->st)) return e;
    if (b->p > 0) {
        const size_t Pq = b->p;
        if (b->firstcall && !b->singular) {       // a singular S in any problem of the batch at the first call: S + A'A for all
            KKT_HIP_CHECK(hipMemcpyAsync(b->pw.h_info, b->pw.d_info, sizeof(int) * B, hipMemcpyDeviceToHost, b->st));
            KKT_HIP_CHECK(hipStreamSynchronize(b->st));
            bool bad = false;
            for (size_t i = 0; i < B; ++i) bad = bad || b->pw.h_info[i] > 0;
            b->firstcall = false;
            if (bad) {
                b->singular = true;
                return mi355kkt_batch_factor(b, b->dW, 1, info);
            }
        }
        b->firstcall = false;
        // Asct_b = L_b^-1 A_b';  K_b = Asct_b' Asct_b;  K_b = L_K L_K'
        hipLaunchKernelGGL(transpose_kernel, dim3((b->n + 31) / 32, (b->p + 31) / 32, b->nbatch), dim3(32, 8), 0, b->st, b->dA,
                           (int64_t)Pq, b->p, b->n, b->dAsct, (int64_t)(Pq * N), (int64_t)(N * Pq));
        if (int e = launch_trsm_lower(b->dS, (int64_t)N, b->n, b->dAsct, (int64_t)N, b->p, 0, b->st, b->nbatch, (int64_t)(N * N),
                                      (int64_t)(N * Pq)))
            return e;
        BatchStrides bk;
        bk.a = (int64_t)(N * Pq); bk.b = 0; bk.c = (int64_t)(Pq * Pq); bk.d = 0;
        if (int e = launch_syrk_scaled(b->planK, b->dAsct, (int64_t)N, nullptr, b->dK, (int64_t)Pq, nullptr, 0, b->st, nullptr, b->nbatch, bk))
            return e;
        if (int e = launch_potrf_batched(b->dK, (int64_t)Pq, b->p, b->nbatch, (int64_t)(Pq * Pq), b->pwK, b->st)) return e;
        // a failing K (rank(A_b) < p) is reported through the same per-problem info words
        hipLaunchKernelGGL(batch_merge_info_kernel, g1(b->nbatch), dim3(256), 0, b->st, b->pw.d_info, b->pwK.d_info, b->n, b->nbatch);
    }
    KKT_HIP_CHECK(hipEventRecord(b->ev[1], b->st));
    if (b->defer_sync) return 0;
    KKT_HIP_CHECK(hipMemcpyAsync(b->pw.h_info, b->pw.d_info, sizeof(int) * B, hipMemcpyDeviceToHost, b->st));
    KKT_HIP_CHECK(hipStreamSynchronize(b->st));
    (void)hipEventElapsedTime(&b->t_factor, b->ev[0], b->ev[1]);
    if (info) memcpy(info, b->pw.h_info, sizeof(int) * B);
    return 0;
} catch (...) { return kkt_catch("mi355kkt_batch_factor"); }

/* Factorisation for a batch with second-order cones: di [nbatch][cdim] (the first nl entries of every slice: 1 / d of the 'l'
 * block), v [nbatch][sum(q)] (the cones' v_k back to back), beta [nbatch][nq] — the W of misc.py:307-354 per problem. */
int mi355kkt_batch_factor_cones(mi355kkt_batch* b, const double* di, const double* v, const double* beta, int is_device, int* info) try {
    if (!b || (!di && b->nl) || (!b->q.empty() && (!v || !beta))) { set_last_error("batch_factor_cones: null argument"); return MI355KKT_EINVAL; }
    KKT_HIP_CHECK(hipSetDevice(b->device));
    if (!b->q.empty()) {
        const hipMemcpyKind kind = is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        const size_t B = b->nbatch;
        if (v != b->dV) KKT_HIP_CHECK(hipMemcpyAsync(b->dV, v, sizeof(double) * B * b->sumq, kind, b->st));
        if (beta != b->dBeta) KKT_HIP_CHECK(hipMemcpyAsync(b->dBeta, beta, sizeof(double) * B * b->q.size(), kind, b->st));
        b->w_set = true;
        if (!di) {                     // no 'l' rows: the scaling kernel never reads di, the LP path wants a valid pointer
            KKT_HIP_CHECK(hipMemsetAsync(b->dW, 0, sizeof(double) * B * b->ml, b->st));
            return mi355kkt_batch_factor(b, b->dW, 1, info);
        }
    }
    return mi355kkt_batch_factor(b, di, is_device, info);
} catch (...) { return kkt_catch("mi355kkt_batch_factor_cones"); }

/* x: [nbatch][n], z: [nbatch][ml], in place: (bx, bz) -> (ux, W uz) per problem (p = 0). */
int mi355kkt_batch_solve(mi355kkt_batch* b, double* x, double* z, int is_device) try {
    if (b && b->p > 0) { set_last_error("batch_solve: the batch has equality constraints, use mi355kkt_batch_solve_eq"); return MI355KKT_EINVAL; }
    return mi355kkt_batch_solve_eq(b, x, nullptr, z, is_device);
} catch (...) { return kkt_catch("mi355kkt_batch_solve"); }

/* x: [nbatch][n], y: [nbatch][p], z: [nbatch][ml], in place: (bx, by, bz) -> (ux, uy, W uz) per problem (misc.py:1513-1563). */
int mi355kkt_batch_solve_eq(mi355kkt_batch* b, double* x, double* y, double* z, int is_device) try {
    if (!b || !x || (!z && b->ml) || (!y && b->p)) { set_last_error("batch_solve: null argument"); return MI355KKT_EINVAL; }
    KKT_HIP_CHECK(hipSetDevice(b->device));
    const size_t B = b->nbatch, N = b->n, M = b->ml, Pq = b->p;
    double *dx = x, *dz = z, *dy = y;
    if (!is_device) {
        KKT_HIP_CHECK(hipMemcpyAsync(b->dx, x, sizeof(double) * B * N, hipMemcpyHostToDevice, b->st));
        if (M) KKT_HIP_CHECK(hipMemcpyAsync(b->dz, z, sizeof(double) * B * M, hipMemcpyHostToDevice, b->st));
        if (Pq) KKT_HIP_CHECK(hipMemcpyAsync(b->dy, y, sizeof(double) * B * Pq, hipMemcpyHostToDevice, b->st));
        dx = b->dx; dz = b->dz; dy = b->dy;
    }
    const int64_t sG = (int64_t)(M * N), sL = (int64_t)(N * N), sA = (int64_t)(Pq * N), sK = (int64_t)(Pq * Pq);
    const bool cones = !b->q.empty();
    if (cones) {                      // zs = W^-T bz;  x += Gs' zs     (misc.py:1306-1311)
        if (int e = launch_batch_cone_scale(dz, (int64_t)M, (int64_t)M, b->dzs, (int64_t)M, (int64_t)M, 1, b->nbatch, b->ml, b->nl,
                                            (int)b->q.size(), b->sumq, b->d_qoff, b->d_qdim, b->d_large, b->nlarge, b->dW, b->dV, b->dBeta, b->st))
            return e;
        if (int e = launch_gemv_t_scaled(b->dGs, (int64_t)M, b->ml, b->n, nullptr, b->dzs, b->dzs, dx, b->dwork, b->st, b->nbatch, sG)) return e;
    } else if (int e = launch_gemv_t_scaled(b->dG, M ? (int64_t)M : 1, b->ml, b->n, b->dW, dz, b->dzs, dx, b->dwork, b->st, b->nbatch, sG)) {
        return e;
    }
    if (b->singular && Pq)                                             // x += A' by  (misc.py:1527)
        if (int e = launch_gemv_t_scaled(b->dA, (int64_t)Pq, b->p, b->n, nullptr, dy, b->dtp, dx, nullptr, b->st, b->nbatch, sA)) return e;
    if (int e = launch_trsm_lower(b->dS, (int64_t)N, b->n, dx, (int64_t)N, 1, 0, b->st, b->nbatch, sL, (int64_t)N)) return e;
    if (Pq) {
        // y := K^-1 (Asct' x - y);  x := x - Asct y                 (misc.py:1541-1553)
        hipLaunchKernelGGL(scal_kernel, g1((int)(B * Pq)), dim3(256), 0, b->st, dy, (int)(B * Pq), -1.0);
        if (int e = launch_gemv_t_scaled(b->dAsct, (int64_t)N, b->n, b->p, nullptr, dx, b->dt1, dy, nullptr, b->st, b->nbatch, sA)) return e;
        if (int e = launch_trsm_lower(b->dK, (int64_t)Pq, b->p, dy, (int64_t)Pq, 1, 0, b->st, b->nbatch, sK, (int64_t)Pq)) return e;
        if (int e = launch_trsm_lower(b->dK, (int64_t)Pq, b->p, dy, (int64_t)Pq, 1, 1, b->st, b->nbatch, sK, (int64_t)Pq)) return e;
        if (int e = launch_gemv_n_scaled(b->dAsct, (int64_t)N, b->n, b->p, nullptr, dy, dx, dx, -1.0, 1.0, b->dwork, b->st, b->nbatch, sA)) return e;
    }
    if (int e = launch_trsm_lower(b->dS, (int64_t)N, b->n, dx, (int64_t)N, 1, 1, b->st, b->nbatch, sL, (int64_t)N)) return e;
    if (int e = launch_gemv_n_scaled(cones ? b->dGs : b->dG, M ? (int64_t)M : 1, b->ml, b->n, cones ? nullptr : b->dW, dx, b->dzs, dz, 1.0, -1.0,
                                     b->dwork, b->st, b->nbatch, sG))
        return e;
    if (!is_device) {
        KKT_HIP_CHECK(hipMemcpyAsync(x, b->dx, sizeof(double) * B * N, hipMemcpyDeviceToHost, b->st));
        if (M) KKT_HIP_CHECK(hipMemcpyAsync(z, b->dz, sizeof(double) * B * M, hipMemcpyDeviceToHost, b->st));
        if (Pq) KKT_HIP_CHECK(hipMemcpyAsync(y, b->dy, sizeof(double) * B * Pq, hipMemcpyDeviceToHost, b->st));
    }
    if (!b->defer_sync) KKT_HIP_CHECK(hipStreamSynchronize(b->st));
    return 0;
} catch (...) { return kkt_catch("mi355kkt_batch_solve_eq"); }

}  // extern "C"

// The interior-point loop itself, shared by the batched and the single-problem entry points.  `ops` supplies the KKT
// work on the caller's stream: products (fills S.Gx, S.GTz, S.Px from S.x, S.z), factor (from S.di; must leave the
// per-problem info words at d_info, on the device) and solve (in place on S.dx, S.dz).
struct IpmOps {
    std::function<int()> products;
    std::function<int(const double* di, int* d_info, int* h_info_first)> factor;
    std::function<int(double* dx, double* dy, double* dz)> solve;
    std::function<int()> check;      // optional, called once per iteration on an idle stream (hand-off timeouts of the solves)
};
struct IpmHostOut {
    double *x, *s, *z; int *status, *iters; double *pcost, *dcost, *gap; int* iterations_run; double* y;
};
static int run_ipm(IpmWork& w, hipStream_t st, IpmOps& ops, const double* q, const double* h, const double* bvec,
                   int maxiters, double abstol, double reltol, double feastol, const IpmHostOut& o) {
    const IpmState& S = w.S;
    const size_t B = w.B, N = S.n, M = S.m, Pq = S.p;
    if (Pq > 0) KKT_HIP_CHECK(hipMemcpyAsync(S.b, bvec, sizeof(double) * B * Pq, hipMemcpyDefault, st));
    int* d_info = ipm_info_words(w);
    KKT_HIP_CHECK(hipMemcpyAsync(S.q, q, sizeof(double) * B * N, hipMemcpyDefault, st));
    KKT_HIP_CHECK(hipMemcpyAsync(S.h, h, sizeof(double) * B * M, hipMemcpyDefault, st));
    KKT_HIP_CHECK(hipMemsetAsync(w.i32, 0, sizeof(int) * (5 * B + 1), st));
    // ---- starting point: W = I  (coneprog.py:2055-2106)
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((B * M + 255) / 256)), dim3(256), 0, st, S.di, 1.0, (int64_t)(B * M));
    int first_bad = -1;
    if (int e = ops.factor(S.di, d_info, &first_bad)) return e;
    if (first_bad >= 0) { set_last_error("coneqp: Rank([P; G]) < n (problem %d)", first_bad); return 1; }
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((B * N + 255) / 256)), dim3(256), 0, st, S.x, S.q, -1.0, (int64_t)(B * N));
    KKT_HIP_CHECK(hipMemcpyAsync(S.z, S.h, sizeof(double) * B * M, hipMemcpyDeviceToDevice, st));
    if (Pq > 0) KKT_HIP_CHECK(hipMemcpyAsync(S.y, S.b, sizeof(double) * B * Pq, hipMemcpyDeviceToDevice, st));
    if (int e = ops.solve(S.x, S.y, S.z)) return e;
    ipm_launch_start(S, (int)B, st);
    int it = 0;
    for (; it <= maxiters; ++it) {
        if (int e = ops.products()) return e;
        KKT_HIP_CHECK(hipMemsetAsync(S.nactive, 0, sizeof(int), st));
        ipm_launch_residual(S, (int)B, it, maxiters, abstol, reltol, feastol, st);
        KKT_HIP_CHECK(hipMemcpyAsync(w.pinned, S.nactive, sizeof(int), hipMemcpyDeviceToHost, st));
        KKT_HIP_CHECK(hipStreamSynchronize(st));
        if (ops.check) if (int e = ops.check()) return e;
        if (w.pinned[0] == 0) break;
        if (int e = ops.factor(S.di, d_info, nullptr)) return e;
        ipm_launch_info(S, d_info, it, (int)B, st);
        for (int i01 = 0; i01 < 2; ++i01) {
            ipm_launch_rhs(S, (int)B, i01, st);
            if (int e = ops.solve(S.dx, S.dy, S.dz)) return e;
            ipm_launch_post(S, (int)B, i01, st);
        }
        ipm_launch_update(S, (int)B, st);
    }
    KKT_HIP_CHECK(hipMemcpyAsync(o.x, S.x_out, sizeof(double) * B * N, hipMemcpyDefault, st));
    if (o.s) KKT_HIP_CHECK(hipMemcpyAsync(o.s, S.s_out, sizeof(double) * B * M, hipMemcpyDefault, st));
    if (o.z) KKT_HIP_CHECK(hipMemcpyAsync(o.z, S.z_out, sizeof(double) * B * M, hipMemcpyDefault, st));
    if (o.y && Pq > 0) KKT_HIP_CHECK(hipMemcpyAsync(o.y, S.y_out, sizeof(double) * B * Pq, hipMemcpyDefault, st));
    KKT_HIP_CHECK(hipMemcpyAsync(o.status, S.status, sizeof(int) * B, hipMemcpyDefault, st));
    KKT_HIP_CHECK(hipMemcpyAsync(o.iters, S.iters, sizeof(int) * B, hipMemcpyDefault, st));
    if (o.pcost) KKT_HIP_CHECK(hipMemcpyAsync(o.pcost, S.pcost, sizeof(double) * B, hipMemcpyDefault, st));
    if (o.dcost) KKT_HIP_CHECK(hipMemcpyAsync(o.dcost, S.dcost, sizeof(double) * B, hipMemcpyDefault, st));
    if (o.gap) KKT_HIP_CHECK(hipMemcpyAsync(o.gap, S.gap_out, sizeof(double) * B, hipMemcpyDefault, st));
    KKT_HIP_CHECK(hipStreamSynchronize(st));
    if (o.iterations_run) *o.iterations_run = it;
    return 0;
}

// The coneqp loop for a batch with second-order cones: the single-problem kernels of coneqp_ipm.hip, one workgroup per
// problem (QpState::nbatch), around the batched factor / solve / products.  Finished problems keep their last scaling (their
// factorisation is repeated with it, harmlessly), their iterates and results are frozen (qp_residual_kernel / qp_update_kernel
// return at once for them).  Same control flow as mi355kkt_coneqp below; status codes as the LP-cone batch.
static int batch_coneqp_cones(mi355kkt_batch* b, const double* q, const double* hv, const double* bvec, int maxiters, double abstol,
                              double reltol, double feastol, double* x, double* y, double* s, double* z, int* status, int* iters,
                              double* pcost, double* dcost, double* gap, int* iterations_run) {
    const int n = b->n, m = b->ml, np = b->p, nb = b->nbatch;
    const size_t B = nb, N = n, M = m, Pq = np;
    if (int e = qp_alloc(b->qp, n, b->nl, b->q, std::vector<int>(), np, nb)) return e;
    QpWork& w = b->qp;
    QpState& S0 = w.S;
    S0.nbatch = nb;                    // (a batch of one problem still runs the batched control flow below)
    S0.correction = b->use_correction;
    const QpState& S = S0;
    hipStream_t st = b->st;
    struct Guard { mi355kkt_batch* b; ~Guard() { b->defer_sync = false; } } guard{b};
    b->defer_sync = true;
    int* d_info = (nb > 1) ? S.active + 3 * B : w.i32 + 4;
    const int refinement = 1;          // coneprog.py:1862-1865: the default with second-order cones
    auto products = [&](const double* xin, const double* yin, const double* zin) -> int {
        if (int e = mi355kkt_batch_products(b, xin, zin, S.Gx, S.GTz, S.Px, 1)) return e;
        return batch_a_products(b, xin, yin, S.Ax, S.ATy);
    };
    auto factor = [&](int* first_bad) -> int {
        if (int e = mi355kkt_batch_factor_cones(b, S.di, S.v, S.beta, 1, nullptr)) return e;
        KKT_HIP_CHECK(hipMemcpyAsync(d_info, b->pw.d_info, sizeof(int) * B, hipMemcpyDeviceToDevice, st));
        if (first_bad) {
            KKT_HIP_CHECK(hipMemcpyAsync(b->pw.h_info, b->pw.d_info, sizeof(int) * B, hipMemcpyDeviceToHost, st));
            KKT_HIP_CHECK(hipStreamSynchronize(st));
            for (int i = 0; i < nb; ++i)
                if (b->pw.h_info[i] > 0) { *first_bad = i; break; }
        }
        return 0;
    };
    auto solve = [&](double* dx, double* dy, double* dz) { return mi355kkt_batch_solve_eq(b, dx, dy, dz, 1); };
    const QpBuf D{S.dx, S.dy, S.dz, S.ds};
    const QpBuf Wsave{S.wx, S.wy, S.wz, S.ws};
    const QpBuf W2{S.wx2, S.wy2, S.wz2, S.ws2};
    KKT_HIP_CHECK(hipMemcpyAsync(S.q, q, sizeof(double) * B * N, hipMemcpyDefault, st));
    KKT_HIP_CHECK(hipMemcpyAsync(S.h, hv, sizeof(double) * B * M, hipMemcpyDefault, st));
    if (np > 0) KKT_HIP_CHECK(hipMemcpyAsync(S.b, bvec, sizeof(double) * B * Pq, hipMemcpyDefault, st));
    KKT_HIP_CHECK(hipMemsetAsync(w.i32, 0, sizeof(int) * 8, st));
    if (nb > 1) KKT_HIP_CHECK(hipMemsetAsync(S.active, 0, sizeof(int) * 4 * B, st));
    KKT_HIP_CHECK(hipMemsetAsync(S.sc, 0, sizeof(double) * B * QP_NSC, st));
    // ---- starting point with W = I (coneprog.py:2054-2106)
    qp_launch_unit_scaling(S, st);
    int first_bad = -1;
    if (int e = factor(&first_bad)) return e;
    if (first_bad >= 0) { set_last_error("coneqp: Rank(A) < p or Rank([P; A; G]) < n (problem %d)", first_bad); return 1; }
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((B * N + 255) / 256)), dim3(256), 0, st, S.x, S.q, -1.0, (int64_t)(B * N));
    if (np > 0) KKT_HIP_CHECK(hipMemcpyAsync(S.y, S.b, sizeof(double) * B * Pq, hipMemcpyDeviceToDevice, st));
    KKT_HIP_CHECK(hipMemcpyAsync(S.z, S.h, sizeof(double) * B * M, hipMemcpyDeviceToDevice, st));
    if (int e = solve(S.x, S.y, S.z)) return e;
    qp_launch_start(S, st);
    int it = 0;
    for (; it <= maxiters; ++it) {
        if (int e = products(S.x, S.y, S.z)) return e;
        KKT_HIP_CHECK(hipMemsetAsync(S.nactive, 0, sizeof(int), st));
        qp_launch_residual(S, it, maxiters, abstol, reltol, feastol, st);
        KKT_HIP_CHECK(hipMemcpyAsync(w.pinned, S.nactive, sizeof(int), hipMemcpyDeviceToHost, st));
        KKT_HIP_CHECK(hipStreamSynchronize(st));
        if (w.pinned[0] == 0) break;
        if (int e = factor(nullptr)) return e;
        qp_launch_singular(S, d_info, it, st);
        for (int i01 = 0; i01 < 2; ++i01) {
            qp_launch_build(S, D, Wsave, i01, refinement > 0 ? 1 : 0, st);
            qp_launch_f4pre(S, D, st);
            if (int e = solve(D.x, D.y, D.z)) return e;
            qp_launch_f4post(S, D, st);
            for (int r = 0; r < refinement; ++r) {                       // coneprog.py:2330-2345
                qp_launch_copy(S, W2, Wsave, st);
                qp_launch_res_a(S, D, st);
                if (int e = products(D.x, D.y, S.wz3)) return e;
                qp_launch_res_b(S, D, W2, st);
                qp_launch_f4pre(S, W2, st);
                if (int e = solve(W2.x, W2.y, W2.z)) return e;
                qp_launch_f4post(S, W2, st);
                qp_launch_add(S, D, W2, st);
            }
            qp_launch_step(S, D, i01, st);
        }
        qp_launch_update(S, D, st);
    }
    KKT_HIP_CHECK(hipMemcpyAsync(x, S.x_out, sizeof(double) * B * N, hipMemcpyDefault, st));
    if (y && np > 0) KKT_HIP_CHECK(hipMemcpyAsync(y, S.y_out, sizeof(double) * B * Pq, hipMemcpyDefault, st));
    if (s) KKT_HIP_CHECK(hipMemcpyAsync(s, S.s_out, sizeof(double) * B * M, hipMemcpyDefault, st));
    if (z) KKT_HIP_CHECK(hipMemcpyAsync(z, S.z_out, sizeof(double) * B * M, hipMemcpyDefault, st));
    KKT_HIP_CHECK(hipMemcpyAsync(status, S.status, sizeof(int) * B, hipMemcpyDefault, st));
    KKT_HIP_CHECK(hipMemcpyAsync(iters, S.iters, sizeof(int) * B, hipMemcpyDefault, st));
    // pcost / dcost / gap: column QP_PCOST / QP_DCOST / QP_GAP_OUT of the [B][QP_NSC] scalar table
    if (pcost) KKT_HIP_CHECK(hipMemcpy2DAsync(pcost, sizeof(double), S.sc + QP_PCOST, sizeof(double) * QP_NSC, sizeof(double), B, hipMemcpyDefault, st));
    if (dcost) KKT_HIP_CHECK(hipMemcpy2DAsync(dcost, sizeof(double), S.sc + QP_DCOST, sizeof(double) * QP_NSC, sizeof(double), B, hipMemcpyDefault, st));
    if (gap) KKT_HIP_CHECK(hipMemcpy2DAsync(gap, sizeof(double), S.sc + QP_GAP_OUT, sizeof(double) * QP_NSC, sizeof(double), B, hipMemcpyDefault, st));
    KKT_HIP_CHECK(hipStreamSynchronize(st));
    if (iterations_run) *iterations_run = it;
    return 0;
}

extern "C" {

/* The whole LP-cone coneqp loop (reference coneprog.py:2044-2547 with dims = {'l': ml}, no equalities) for every problem
 * of the batch, iterates and bookkeeping resident in HBM; per iteration only the count of still-active problems returns to
 * the host.  See include/mi355kkt.h. */
int mi355kkt_batch_coneqp(mi355kkt_batch* b, const double* q, const double* h, int maxiters, double abstol, double reltol,
                          double feastol, double* x, double* s, double* z, int* status, int* iters, double* pcost,
                          double* dcost, double* gap, int* iterations_run) try {
    if (b && b->p > 0) { set_last_error("batch_coneqp: the batch has equality constraints, use mi355kkt_batch_coneqp_eq"); return MI355KKT_EINVAL; }
    return mi355kkt_batch_coneqp_eq(b, q, h, nullptr, maxiters, abstol, reltol, feastol, x, nullptr, s, z, status, iters, pcost, dcost,
                                    gap, iterations_run);
} catch (...) { return kkt_catch("mi355kkt_batch_coneqp"); }

/* The same with equality constraints A_b x = bvec_b (bvec, y: [nbatch][p]). */
int mi355kkt_batch_coneqp_eq(mi355kkt_batch* b, const double* q, const double* h, const double* bvec, int maxiters, double abstol,
                             double reltol, double feastol, double* x, double* y, double* s, double* z, int* status, int* iters,
                             double* pcost, double* dcost, double* gap, int* iterations_run) try {
    if (!b || !q || (!h && b->ml) || !x || !status || !iters || (b->p > 0 && (!bvec || !y))) {
        set_last_error("batch_coneqp: null argument");
        return MI355KKT_EINVAL;
    }
    if (b->ml < 1) { set_last_error("batch_coneqp: needs at least one inequality"); return MI355KKT_EINVAL; }
    KKT_HIP_CHECK(hipSetDevice(b->device));
    if (!b->q.empty())
        return batch_coneqp_cones(b, q, h, bvec, maxiters, abstol, reltol, feastol, x, y, s, z, status, iters, pcost, dcost, gap,
                                  iterations_run);
    if (int e = ipm_alloc(b->ipm, b->nbatch, b->n, b->ml, b->p)) return e;
    b->ipm.S.correction = b->use_correction;
    const IpmState& S = b->ipm.S;
    struct Guard { mi355kkt_batch* b; ~Guard() { b->defer_sync = false; } } guard{b};
    b->defer_sync = true;
    IpmOps ops;
    ops.products = [&]() -> int {
        if (int e = mi355kkt_batch_products(b, S.x, S.z, S.Gx, S.GTz, S.Px, 1)) return e;
        return batch_a_products(b, S.x, S.y, S.Ax, S.ATy);
    };
    ops.factor = [&](const double* di, int* d_info, int* first_bad) -> int {
        if (int e = mi355kkt_batch_factor(b, di, 1, nullptr)) return e;
        KKT_HIP_CHECK(hipMemcpyAsync(d_info, b->pw.d_info, sizeof(int) * b->nbatch, hipMemcpyDeviceToDevice, b->st));
        if (first_bad) {
            KKT_HIP_CHECK(hipMemcpyAsync(b->pw.h_info, b->pw.d_info, sizeof(int) * b->nbatch, hipMemcpyDeviceToHost, b->st));
            KKT_HIP_CHECK(hipStreamSynchronize(b->st));
            for (int i = 0; i < b->nbatch; ++i)
                if (b->pw.h_info[i] > 0) { *first_bad = i; break; }
        }
        return 0;
    };
    ops.solve = [&](double* dx, double* dy, double* dz) { return mi355kkt_batch_solve_eq(b, dx, dy, dz, 1); };
    IpmHostOut o{x, s, z, status, iters, pcost, dcost, gap, iterations_run, y};
    return run_ipm(b->ipm, b->st, ops, q, h, bvec, maxiters, abstol, reltol, feastol, o);
} catch (...) { return kkt_catch("mi355kkt_batch_coneqp_eq"); }

// the mirrored copy of H (only tril(H) is meaningful on input, coneprog.py:1475-1477) for plain products H x
static int ensure_hsym(mi355kkt_solver* hs) {
    if (!hs->dH || hs->sparse || hs->hsym_valid) return 0;
    const int n = hs->n;
    if (hs->h_pending) KKT_HIP_CHECK(hipStreamWaitEvent(hs->st, hs->ev_h, 0));   // asynchronous upload of H in flight
    if (!hs->dHsym) KKT_HIP_CHECK(DEV_ALLOC(&hs->dHsym, sizeof(double) * dmax((size_t)n * n, 1)));
    if (n > 0) {
        KKT_HIP_CHECK(hipMemcpy2DAsync(hs->dHsym, sizeof(double) * n, hs->dH, sizeof(double) * hs->ldH, sizeof(double) * n, n,
                                       hipMemcpyDeviceToDevice, hs->st));
        hipLaunchKernelGGL(symmetrize_kernel, dim3((n + 15) / 16, (n + 15) / 16, 1), dim3(16, 16), 0, hs->st, hs->dHsym, n, (int64_t)0);
    }
    hs->hsym_valid = true;
    return 0;
}
// G x, G' z, P x of the sparse engine for the loops' cone-space vectors of length cdim.  In the S + A'A mode the engine's G
// carries p more rows (the rows of A): z goes in with a zero tail and G x comes back through a (cdim + p)-vector.
static int sparse_products_cdim(mi355kkt_solver* hs, const double* xin, const double* zin, double* Gx, double* GTz, double* Px,
                                hipStream_t st) {
    if (hs->sp_extra <= 0) return sparse_engine_products(hs->sp, xin, zin, Gx, GTz, Px, st);
    const size_t m = (size_t)hs->cdim;
    KKT_HIP_CHECK(hipMemcpyAsync(hs->dz, zin, sizeof(double) * m, hipMemcpyDeviceToDevice, st));
    KKT_HIP_CHECK(hipMemsetAsync(hs->dz + m, 0, sizeof(double) * hs->sp_extra, st));
    if (int e = sparse_engine_products(hs->sp, xin, hs->dz, hs->dzs, GTz, Px, st)) return e;
    KKT_HIP_CHECK(hipMemcpyAsync(Gx, hs->dzs, sizeof(double) * m, hipMemcpyDeviceToDevice, st));
    return 0;
}
// A xin -> Ax, A' yin -> ATy with the A of the handle (dense, or CSR/CSC in sparse mode)
static int a_products(mi355kkt_solver* hs, const double* xin, const double* yin, double* Ax, double* ATy, double* gwork, hipStream_t st) {
    const int np = hs->p;
    if (np <= 0) return 0;
    if (int e = A_mul(hs, xin, Ax, gwork, st)) return e;
    return A_mulT(hs, yin, ATy, 0.0, gwork, st);
}
static int ensure_gemv_work(mi355kkt_solver* hs) {
    if (hs->dIpmWork) return 0;
    const int n = hs->n, m = hs->cdim, np = hs->p;
    if (hs->sparse) {          // only the (dense) A products need it
        if (np > 0) KKT_HIP_CHECK(DEV_ALLOC(&hs->dIpmWork, sizeof(double) * gemv_work_doubles(np, n)));
        return 0;
    }
    KKT_HIP_CHECK(DEV_ALLOC(&hs->dIpmWork, sizeof(double) * dmax(dmax(gemv_work_doubles(m, n), gemv_work_doubles(n, n)),
                                                                gemv_work_doubles(np, n))));
    return 0;
}

/* out = op(M) x on the device, host vectors in and out: which = 0: G (cdim x n), 1: A (p x n), 2: H (n x n, symmetric,
 * tril(H) as set); trans != 0: op = transpose.  The operator form of the reference's fG / fA / fP closures
 * (coneprog.py:531-550, :1843-1844, :1896-1916) for callers that hand conelp / coneqp callables instead of matrices. */
int mi355kkt_product(mi355kkt_solver* hs, int which, int trans, const double* x, double* out) try {
    if (!hs || !x || !out || which < 0 || which > 2) { set_last_error("product: invalid argument"); return MI355KKT_EINVAL; }
    if (int e = bind(hs)) return e;
    const int n = hs->n, m = hs->cdim, np = hs->p;
    hipStream_t st = hs->st;
    const int rows = which == 0 ? m : (which == 1 ? np : n);
    const int nin = (which == 2) ? n : (trans ? rows : n), nout = (which == 2) ? n : (trans ? n : rows);
    if (nout == 0) return 0;
    double* din = (nin == n) ? hs->dx : (which == 0 ? hs->dz : hs->dy);
    double* dout = (nout == n) ? hs->dtn : (which == 0 ? hs->dzs : hs->dtp);
    if (which == 2 && !hs->sparse && !hs->dH) { memset(out, 0, sizeof(double) * n); return 0; }
    if (nin > 0) {
        memcpy(hs->hbuf, x, sizeof(double) * nin);
        KKT_HIP_CHECK(hipMemcpyAsync(din, hs->hbuf, sizeof(double) * nin, hipMemcpyHostToDevice, st));
    }
    if (hs->sparse && which != 1) {
        if (which == 0 && trans && hs->sp_extra > 0)       // S + A'A mode: the engine's G has p more rows; they get zeros
            KKT_HIP_CHECK(hipMemsetAsync(din + hs->cdim, 0, sizeof(double) * hs->sp_extra, st));
        if (int e = sparse_engine_product(hs->sp, which, trans, din, dout, st)) return e;
    } else if (which == 1 && hs->A_sparse) {       // sparse A (CSR / CSC on the device)
        if (nin == 0) KKT_HIP_CHECK(hipMemsetAsync(dout, 0, sizeof(double) * nout, st));
        else if (!trans) { if (int e = A_mul(hs, din, dout, nullptr, st)) return e; }
        else if (int e = A_mulT(hs, din, dout, 0.0, nullptr, st)) return e;
    } else {
        if ((which == 0 && !hs->dG) || (which == 1 && np > 0 && !hs->dA)) { set_last_error("product: matrix not set"); return MI355KKT_EINVAL; }
        if (int e = ensure_gemv_work(hs)) return e;
        const double* M = which == 0 ? hs->dG : (which == 1 ? hs->dA : nullptr);
        const int64_t ld = which == 0 ? hs->ldG : hs->ldA;
        if (which == 2) {
            if (int e = ensure_hsym(hs)) return e;
            if (int e = launch_gemv_n_scaled(hs->dHsym, n, n, n, nullptr, din, dout, dout, 1.0, 0.0, hs->dIpmWork, st)) return e;
        } else if (nin == 0) {
            KKT_HIP_CHECK(hipMemsetAsync(dout, 0, sizeof(double) * nout, st));
        } else if (!trans) {
            if (int e = launch_gemv_n_scaled(M, ld, rows, n, nullptr, din, dout, dout, 1.0, 0.0, hs->dIpmWork, st)) return e;
        } else {
            KKT_HIP_CHECK(hipMemsetAsync(dout, 0, sizeof(double) * n, st));
            double* scratch = which == 0 ? hs->dzs : hs->dtp;
            if (int e = launch_gemv_t_scaled(M, ld, rows, n, nullptr, din, scratch, dout, hs->dIpmWork, st)) return e;
        }
    }
    KKT_HIP_CHECK(hipMemcpyAsync(hs->hbuf, dout, sizeof(double) * nout, hipMemcpyDeviceToHost, st));
    KKT_HIP_CHECK(hipStreamSynchronize(st));
    memcpy(out, hs->hbuf, sizeof(double) * nout);
    return 0;
} catch (...) { return kkt_catch("mi355kkt_product"); }

/* Single problem, LP cone, no equality constraints: the coneqp loop of coneprog.py:2044-2547 resident on the device around
 * this handle's own factor/solve (dense or sparse mode).  G (and H, if any) must have been set.  See include/mi355kkt.h. */
int mi355kkt_coneqp_lp(mi355kkt_solver* hs, const double* q, const double* hv, const double* bv, int maxiters,
                       double abstol, double reltol, double feastol, double* x, double* y, double* s, double* z, int* status,
                       int* iters, double* pcost, double* dcost, double* gap) try {
    if (!hs || !q || !hv || !x || !status || !iters || (hs->p > 0 && (!bv || !y))) {
        set_last_error("coneqp_lp: null argument");
        return MI355KKT_EINVAL;
    }
    if (!hs->q.empty() || !hs->s.empty() || hs->ml < 1) {
        set_last_error("coneqp_lp: needs dims = {'l': m > 0}");
        return MI355KKT_ENOTIMPL;
    }
    if (hs->p > 0 && !hs->dA && !hs->A_sparse) { set_last_error("coneqp_lp: A not set"); return MI355KKT_EINVAL; }
    if (int e = bind(hs)) return e;
    const int n = hs->n, m = hs->ml;
    const int np = hs->p;
    if (int e = ipm_alloc(hs->ipm, 1, n, m, np)) return e;
    hs->ipm.S.correction = hs->use_correction;
    const IpmState& S = hs->ipm.S;
    hipStream_t st = hs->st;
    if (int e = ensure_hsym(hs)) return e;
    double* scratch = hs->dzs;       // >= cdim doubles; free between solves
    if (int e = ensure_gemv_work(hs)) return e;
    double* gwork = hs->dIpmWork;
    IpmOps ops;
    ops.check = [&]() { return loop_check_handoff(hs); };
    ops.products = [&]() -> int {
        if (hs->sparse) {
            if (int e = sparse_products_cdim(hs, S.x, S.z, S.Gx, S.GTz, S.Px, st)) return e;
            return a_products(hs, S.x, S.y, S.Ax, S.ATy, gwork, st);
        }
        if (int e = launch_gemv_n_scaled(hs->dG, hs->ldG, m, n, nullptr, S.x, S.Gx, S.Gx, 1.0, 0.0, gwork, st)) return e;
        KKT_HIP_CHECK(hipMemsetAsync(S.GTz, 0, sizeof(double) * n, st));
        if (int e = launch_gemv_t_scaled(hs->dG, hs->ldG, m, n, nullptr, S.z, scratch, S.GTz, gwork, st)) return e;
        if (hs->dH) {
            if (int e = launch_gemv_n_scaled(hs->dHsym, n, n, n, nullptr, S.x, S.Px, S.Px, 1.0, 0.0, gwork, st)) return e;
        } else {
            KKT_HIP_CHECK(hipMemsetAsync(S.Px, 0, sizeof(double) * n, st));
        }
        if (np > 0) {                                  // A x and A' y (coneprog.py:2176, :2181)
            if (int e = launch_gemv_n_scaled(hs->dA, hs->ldA, np, n, nullptr, S.x, S.Ax, S.Ax, 1.0, 0.0, gwork, st)) return e;
            KKT_HIP_CHECK(hipMemsetAsync(S.ATy, 0, sizeof(double) * n, st));
            if (int e = launch_gemv_t_scaled(hs->dA, hs->ldA, np, n, nullptr, S.y, hs->dtp, S.ATy, gwork, st)) return e;
        }
        return 0;
    };
    ops.factor = [&](const double* di, int* d_info, int* first_bad) -> int {
        mi355kkt_scaling W = {};
        W.di = di;
        const int info = mi355kkt_factor_device(hs, &W);
        if (info < 0) return info;
        if (first_bad && info > 0) *first_bad = 0;
        hs->ipm.pinned[1] = info;
        KKT_HIP_CHECK(hipMemcpyAsync(d_info, hs->ipm.pinned + 1, sizeof(int), hipMemcpyHostToDevice, st));
        if (info > 0) hs->factored = true;   // the loop drops the problem before any solve result is used
        return 0;
    };
    ops.solve = [&](double* dx, double* dy, double* dz) { return mi355kkt_solve_device(hs, dx, dy, dz); };
    IpmHostOut o{x, s, z, status, iters, pcost, dcost, gap, nullptr, y};
    return run_ipm(hs->ipm, st, ops, q, hv, bv, maxiters, abstol, reltol, feastol, o);
} catch (...) { return kkt_catch("mi355kkt_coneqp_lp"); }

// The device loops keep the 's' blocks of every cone vector as full symmetric matrices (cone_ops_s.h), so that G x is symmetric
// and G'z is the reference's sgemv (misc.py:801-832: only the lower triangles of the 's' blocks of the columns of G count).
// One pass over the handle's own copy of G: upper triangles := lower triangles.  Harmless for factor / solve, which read the
// lower triangles only.
__global__ __launch_bounds__(256) void g_symm_sblocks_kernel(double* G, int64_t ldG, const int* sdim, const int* soff) {
    double* col = G + (size_t)blockIdx.x * ldG + soff[blockIdx.y];
    const int m = sdim[blockIdx.y];
    for (int e = threadIdx.x; e < m * m; e += 256)
        if (e % m < e / m) col[e] = col[(e / m) + (size_t)(e % m) * m];
}
static int symmetrize_G_sblocks(mi355kkt_solver* hs) {
    if (hs->s.empty() || hs->n == 0) return 0;
    if (!hs->G_owned || hs->dG != hs->G_owned) {
        set_last_error("device loops with 's' cones need G set with mi355kkt_set_G_dense / set_G_csc (the handle's own copy)");
        return MI355KKT_ENOTIMPL;
    }
    hipLaunchKernelGGL(g_symm_sblocks_kernel, dim3(hs->n, (unsigned)hs->s.size()), dim3(256), 0, hs->st, hs->G_owned, hs->ldG,
                       hs->cl.d_sdim, hs->cl.d_soff);
    KKT_HIP_CHECK(hipGetLastError());
    return 0;
}

/* Single problem, 'l' + 'q' cones: the conelp loop of coneprog.py:586-1436 (self-dual embedding, default starting point,
 * refinement 0 for the LP cone and 1 with second-order cones, :502-507) resident on the device around this handle's
 * factor/solve (H must be absent).  See include/mi355kkt.h. */
int mi355kkt_conelp(mi355kkt_solver* hs, const double* c, const double* hv, const double* bv, int maxiters, double abstol,
                    double reltol, double feastol, int refinement, double* x, double* y, double* s, double* z, int* status,
                    int* iters, double* stats) try {
    return mi355kkt_conelp_init(hs, c, hv, bv, maxiters, abstol, reltol, feastol, refinement, 0, 0, x, y, s, z, status, iters, stats);
} catch (...) { return kkt_catch("mi355kkt_conelp"); }

/* have_primal: x, s hold primalstart on entry; have_dual: y, z hold dualstart (coneprog.py:696-739; the caller has checked that the
 * given s / z are in the interior of the cone).  The part that is not given is constructed as in the default start. */
int mi355kkt_conelp_init(mi355kkt_solver* hs, const double* c, const double* hv, const double* bv, int maxiters, double abstol,
                         double reltol, double feastol, int refinement, int have_primal, int have_dual, double* x, double* y,
                         double* s, double* z, int* status, int* iters, double* stats) try {
    if (!hs || !c || !hv || !x || !s || !z || !status || !iters || (hs->p > 0 && (!bv || !y))) {
        set_last_error("conelp: null argument");
        return MI355KKT_EINVAL;
    }
    if (hs->cdim < 1) { set_last_error("conelp: needs at least one cone row"); return MI355KKT_ENOTIMPL; }
    if (hs->dH) { set_last_error("conelp: the handle carries a quadratic term (H)"); return MI355KKT_EINVAL; }
    if (hs->p > 0 && !hs->dA && !hs->A_sparse) { set_last_error("conelp: A not set"); return MI355KKT_EINVAL; }
    if (int e = bind(hs)) return e;
    const int n = hs->n, m = hs->cdim, np = hs->p;
    if (refinement < 0) refinement = (hs->q.empty() && hs->s.empty()) ? 0 : 1;
    if (int e = lp_alloc(hs->lp, n, hs->ml, hs->q, hs->s, np)) return e;
    if (int e = symmetrize_G_sblocks(hs)) return e;
    LpWork& w = hs->lp;
    const LpState& S = w.S;
    hipStream_t st = hs->st;
    if (int e = ensure_gemv_work(hs)) return e;
    double* gwork = hs->dIpmWork;
    int* d_info = w.i32 + 5;
    // G xin -> Gx, A xin -> Ax, G' zin -> GTz, A' yin -> ATy
    auto products = [&](const double* xin, const double* yin, const double* zin) -> int {
        if (hs->sparse) {
            if (int e = sparse_products_cdim(hs, xin, zin, S.Gx, S.GTz, hs->dtn /* P x = 0 lands in the handle's scratch n-vector (only used inside solve()); NOT x_out: when the constructed starting point is already optimal it holds the answer (coneprog.py:752-790) */, st)) return e;
            return a_products(hs, xin, yin, S.Ax, S.ATy, gwork, st);
        }
        if (int e = launch_gemv_n_scaled(hs->dG, hs->ldG, m, n, nullptr, xin, S.Gx, S.Gx, 1.0, 0.0, gwork, st)) return e;
        KKT_HIP_CHECK(hipMemsetAsync(S.GTz, 0, sizeof(double) * n, st));
        if (int e = launch_gemv_t_scaled(hs->dG, hs->ldG, m, n, nullptr, zin, hs->dzs, S.GTz, gwork, st)) return e;
        if (np > 0) {
            if (int e = launch_gemv_n_scaled(hs->dA, hs->ldA, np, n, nullptr, xin, S.Ax, S.Ax, 1.0, 0.0, gwork, st)) return e;
            KKT_HIP_CHECK(hipMemsetAsync(S.ATy, 0, sizeof(double) * n, st));
            if (int e = launch_gemv_t_scaled(hs->dA, hs->ldA, np, n, nullptr, yin, hs->dtp, S.ATy, gwork, st)) return e;
        }
        return 0;
    };
    auto factor = [&](int* info_out) -> int {
        mi355kkt_scaling W = {};
        W.di = S.di; W.d = S.d; W.v = S.v; W.beta = S.beta; W.r = S.r; W.rti = S.rti;
        const int info = mi355kkt_factor_device(hs, &W);
        if (info < 0) return info;
        *info_out = info;
        w.pinned[1] = info;
        KKT_HIP_CHECK(hipMemcpyAsync(d_info, w.pinned + 1, sizeof(int), hipMemcpyHostToDevice, st));
        if (info > 0) hs->factored = true;   // the loop leaves before any solve result is used
        return 0;
    };
    // (the solve returns the lower triangles of the 's' blocks of z, misc_solvers.c:552-601; the loop keeps both)
    auto solve = [&](double* dx, double* dy, double* dz) -> int {
        if (int e = mi355kkt_solve_device(hs, dx, dy, dz)) return e;
        lp_launch_symm(S, dz, st);
        return 0;
    };
    auto dcopy = [&](double* dst, const double* src, size_t k) -> int {
        if (k) KKT_HIP_CHECK(hipMemcpyAsync(dst, src, sizeof(double) * k, hipMemcpyDeviceToDevice, st));
        return 0;
    };
    const LpBuf D{S.dx, S.dy, S.dz, S.ds, LP_DTAU, LP_DKAPPA};
    const LpBuf Wsave{S.wx, S.wy, S.wz, S.ws, LP_WTAU, LP_WKAPPA};
    const LpBuf W2{S.wx2, S.wy2, S.wz2, S.ws2, LP_WTAU2, LP_WKAPPA2};
    KKT_HIP_CHECK(hipMemcpyAsync(S.c, c, sizeof(double) * n, hipMemcpyHostToDevice, st));
    KKT_HIP_CHECK(hipMemcpyAsync(S.h, hv, sizeof(double) * m, hipMemcpyHostToDevice, st));
    lp_launch_symm(S, S.h, st);                       // only the lower triangles of the 's' blocks of h are significant
    if (np > 0) KKT_HIP_CHECK(hipMemcpyAsync(S.b, bv, sizeof(double) * np, hipMemcpyHostToDevice, st));
    KKT_HIP_CHECK(hipMemsetAsync(w.i32, 0, sizeof(int) * 8, st));
    KKT_HIP_CHECK(hipMemsetAsync(S.sc, 0, sizeof(double) * LP_NSC, st));
    // ---- starting points with W = I (coneprog.py:664-748)
    lp_launch_unit_scaling(S, st);
    int info = 0;
    if (!(have_primal && have_dual)) {                    // (the reference factors W = I only if a start has to be constructed)
        if (int e = factor(&info)) return e;
        if (info > 0) { set_last_error("conelp: Rank(A) < p or Rank([G; A]) < n"); return 1; }
    }
    if (have_primal) {                                   // coneprog.py:703-705
        KKT_HIP_CHECK(hipMemcpyAsync(S.x, x, sizeof(double) * n, hipMemcpyHostToDevice, st));
        KKT_HIP_CHECK(hipMemcpyAsync(S.s, s, sizeof(double) * m, hipMemcpyHostToDevice, st));
        lp_launch_symm(S, S.s, st);
    } else {
        KKT_HIP_CHECK(hipMemsetAsync(S.x, 0, sizeof(double) * n, st));
        if (int e = dcopy(S.dy, S.b, np)) return e;
        if (int e = dcopy(S.s, S.h, m)) return e;
        if (int e = solve(S.x, S.dy, S.s)) return e;
    }
    lp_launch_init_primal(S, st, have_primal);
    if (have_dual) {                                     // coneprog.py:735-737
        if (np > 0) KKT_HIP_CHECK(hipMemcpyAsync(S.y, y, sizeof(double) * np, hipMemcpyHostToDevice, st));
        KKT_HIP_CHECK(hipMemcpyAsync(S.z, z, sizeof(double) * m, hipMemcpyHostToDevice, st));
        lp_launch_symm(S, S.z, st);
    } else {
        hipLaunchKernelGGL(axpby_kernel, dim3((n + 255) / 256), dim3(256), 0, st, S.dx, S.c, -1.0, (int64_t)n);
        if (np > 0) KKT_HIP_CHECK(hipMemsetAsync(S.y, 0, sizeof(double) * np, st));
        KKT_HIP_CHECK(hipMemsetAsync(S.z, 0, sizeof(double) * m, st));
        if (int e = solve(S.dx, S.y, S.z)) return e;
    }
    lp_launch_init_dual(S, abstol, reltol, st, have_primal, have_dual);
    int it = 0;
    for (; it <= maxiters; ++it) {
        if (int e = products(S.x, S.y, S.z)) return e;
        KKT_HIP_CHECK(hipMemsetAsync(S.nactive, 0, sizeof(int), st));
        lp_launch_residual(S, it, maxiters, abstol, reltol, feastol, st);
        KKT_HIP_CHECK(hipMemcpyAsync(w.pinned, S.nactive, sizeof(int), hipMemcpyDeviceToHost, st));
        KKT_HIP_CHECK(hipStreamSynchronize(st));
        {   // coneprog.py:984-990: pcost dcost gap pres dres k/t
            static const int idx[5] = {LP_PCOST, LP_DCOST, LP_GAP_OUT, LP_PRES, LP_DRES};
            if (int e = report_progress(hs, it, S.sc, LP_NSC, idx, 5, LP_TAU, LP_KAPPA)) return e;
        }
        if (int e = loop_check_handoff(hs)) return e;
        if (w.pinned[0] == 0) break;
        if (int e = factor(&info)) return e;
        lp_launch_singular(S, d_info, it, st);
        if (info > 0) break;
        if (int e = solve(S.x1, S.y1, S.z1)) return e;
        lp_launch_scale1(S, st);
        for (int i01 = 0; i01 < 2; ++i01) {
            lp_launch_build(S, D, Wsave, i01, refinement > 0 ? 1 : 0, st);
            lp_launch_f6pre(S, D, st);                                   // f6 = f6_no_ir + `refinement` correction steps
            if (int e = solve(D.x, D.y, D.z)) return e;
            lp_launch_f6post(S, D, st);
            for (int r = 0; r < refinement; ++r) {                       // coneprog.py:1220-1235
                lp_launch_copy(S, W2, Wsave, st);
                lp_launch_res_a(S, D, st);
                if (int e = products(D.x, D.y, S.wz3)) return e;
                lp_launch_res_b(S, D, W2, st);
                lp_launch_f6pre(S, W2, st);
                if (int e = solve(W2.x, W2.y, W2.z)) return e;
                lp_launch_f6post(S, W2, st);
                lp_launch_add(S, D, W2, st);
            }
            lp_launch_step(S, D, i01, st);
        }
        lp_launch_update(S, D, st);
    }
    KKT_HIP_CHECK(hipMemcpyAsync(x, S.x_out, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    if (np > 0) KKT_HIP_CHECK(hipMemcpyAsync(y, S.y_out, sizeof(double) * np, hipMemcpyDeviceToHost, st));
    KKT_HIP_CHECK(hipMemcpyAsync(s, S.s_out, sizeof(double) * m, hipMemcpyDeviceToHost, st));
    KKT_HIP_CHECK(hipMemcpyAsync(z, S.z_out, sizeof(double) * m, hipMemcpyDeviceToHost, st));
    KKT_HIP_CHECK(hipMemcpyAsync(status, S.status, sizeof(int), hipMemcpyDeviceToHost, st));
    KKT_HIP_CHECK(hipMemcpyAsync(iters, S.iters, sizeof(int), hipMemcpyDeviceToHost, st));
    double hsc[LP_NSC];
    KKT_HIP_CHECK(hipMemcpyAsync(hsc, S.sc, sizeof(double) * LP_NSC, hipMemcpyDeviceToHost, st));
    KKT_HIP_CHECK(hipStreamSynchronize(st));
    if (stats) {   // gap, relative gap, pcost, dcost, pres, dres, pinfres, dinfres, ts, tz (1e300 = "None")
        stats[0] = hsc[LP_GAP_OUT]; stats[1] = hsc[LP_RELGAP]; stats[2] = hsc[LP_PCOST]; stats[3] = hsc[LP_DCOST];
        stats[4] = hsc[LP_PRES]; stats[5] = hsc[LP_DRES]; stats[6] = hsc[LP_PINFRES]; stats[7] = hsc[LP_DINFRES];
        stats[8] = hsc[LP_TS]; stats[9] = hsc[LP_TZ];
        if (*iters == 0 && *status == 1 && hsc[LP_GAP_OUT] == 0.0) stats[0] = hsc[LP_GAP];   // optimal starting point
    }
    return 0;
} catch (...) { return kkt_catch("mi355kkt_conelp_init"); }

/* Single problem, 'l' + 'q' cones: the coneqp loop of coneprog.py:2044-2547 (refinement 0 for the LP cone, 1 with
 * second-order cones, :1862-1865) resident on the device around this handle's factor/solve.  See include/mi355kkt.h. */
int mi355kkt_coneqp(mi355kkt_solver* hs, const double* q, const double* hv, const double* bv, int maxiters, double abstol,
                    double reltol, double feastol, int refinement, double* x, double* y, double* s, double* z, int* status,
                    int* iters, double* stats) try {
    return mi355kkt_coneqp_init(hs, q, hv, bv, maxiters, abstol, reltol, feastol, refinement, 0, x, y, s, z, status, iters, stats);
} catch (...) { return kkt_catch("mi355kkt_coneqp"); }

/* have_init != 0: x, y, s, z hold the caller's starting point on entry (initvals of solvers.coneqp, coneprog.py:2109-2149: the
 * caller has filled in the reference's defaults -- x = 0, y = 0, s = z = e -- for missing entries and checked s, z > 0); the
 * factorisation / solve with W = I of the default start is skipped. */
int mi355kkt_coneqp_init(mi355kkt_solver* hs, const double* q, const double* hv, const double* bv, int maxiters, double abstol,
                         double reltol, double feastol, int refinement, int have_init, double* x, double* y, double* s, double* z,
                         int* status, int* iters, double* stats) try {
    if (!hs || !q || !hv || !x || !s || !z || !status || !iters || (hs->p > 0 && (!bv || !y))) {
        set_last_error("coneqp: null argument");
        return MI355KKT_EINVAL;
    }
    if (hs->cdim < 1) { set_last_error("coneqp: needs at least one cone row"); return MI355KKT_ENOTIMPL; }
    if (hs->p > 0 && !hs->dA && !hs->A_sparse) { set_last_error("coneqp: A not set"); return MI355KKT_EINVAL; }
    if (int e = bind(hs)) return e;
    const int n = hs->n, m = hs->cdim, np = hs->p;
    if (refinement < 0) refinement = (hs->q.empty() && hs->s.empty()) ? 0 : 1;
    if (int e = qp_alloc(hs->qp, n, hs->ml, hs->q, hs->s, np)) return e;
    hs->qp.S.correction = hs->use_correction;
    if (int e = symmetrize_G_sblocks(hs)) return e;
    QpWork& w = hs->qp;
    const QpState& S = w.S;
    hipStream_t st = hs->st;
    if (int e = ensure_gemv_work(hs)) return e;
    if (int e = ensure_hsym(hs)) return e;
    double* gwork = hs->dIpmWork;
    int* d_info = w.i32 + 4;
    // P xin -> Px, G xin -> Gx, A xin -> Ax, G' zin -> GTz, A' yin -> ATy
    auto products = [&](const double* xin, const double* yin, const double* zin) -> int {
        if (hs->sparse) {
            if (int e = sparse_products_cdim(hs, xin, zin, S.Gx, S.GTz, S.Px, st)) return e;
            return a_products(hs, xin, yin, S.Ax, S.ATy, gwork, st);
        }
        if (int e = launch_gemv_n_scaled(hs->dG, hs->ldG, m, n, nullptr, xin, S.Gx, S.Gx, 1.0, 0.0, gwork, st)) return e;
        KKT_HIP_CHECK(hipMemsetAsync(S.GTz, 0, sizeof(double) * n, st));
        if (int e = launch_gemv_t_scaled(hs->dG, hs->ldG, m, n, nullptr, zin, hs->dzs, S.GTz, gwork, st)) return e;
        if (hs->dH) {
            if (int e = launch_gemv_n_scaled(hs->dHsym, n, n, n, nullptr, xin, S.Px, S.Px, 1.0, 0.0, gwork, st)) return e;
        } else {
            KKT_HIP_CHECK(hipMemsetAsync(S.Px, 0, sizeof(double) * n, st));
        }
        if (np > 0) {
            if (int e = launch_gemv_n_scaled(hs->dA, hs->ldA, np, n, nullptr, xin, S.Ax, S.Ax, 1.0, 0.0, gwork, st)) return e;
            KKT_HIP_CHECK(hipMemsetAsync(S.ATy, 0, sizeof(double) * n, st));
            if (int e = launch_gemv_t_scaled(hs->dA, hs->ldA, np, n, nullptr, yin, hs->dtp, S.ATy, gwork, st)) return e;
        }
        return 0;
    };
    auto factor = [&](int* info_out) -> int {
        mi355kkt_scaling W = {};
        W.di = S.di; W.d = S.d; W.v = S.v; W.beta = S.beta; W.r = S.r; W.rti = S.rti;
        const int info = mi355kkt_factor_device(hs, &W);
        if (info < 0) return info;
        *info_out = info;
        w.pinned[1] = info;
        KKT_HIP_CHECK(hipMemcpyAsync(d_info, w.pinned + 1, sizeof(int), hipMemcpyHostToDevice, st));
        if (info > 0) hs->factored = true;
        return 0;
    };
    auto solve = [&](double* dx, double* dy, double* dz) -> int {
        if (int e = mi355kkt_solve_device(hs, dx, dy, dz)) return e;
        qp_launch_symm(S, dz, st);
        return 0;
    };
    const QpBuf D{S.dx, S.dy, S.dz, S.ds};
    const QpBuf Wsave{S.wx, S.wy, S.wz, S.ws};
    const QpBuf W2{S.wx2, S.wy2, S.wz2, S.ws2};
    KKT_HIP_CHECK(hipMemcpyAsync(S.q, q, sizeof(double) * n, hipMemcpyHostToDevice, st));
    KKT_HIP_CHECK(hipMemcpyAsync(S.h, hv, sizeof(double) * m, hipMemcpyHostToDevice, st));
    qp_launch_symm(S, S.h, st);                       // only the lower triangles of the 's' blocks of h are significant
    if (np > 0) KKT_HIP_CHECK(hipMemcpyAsync(S.b, bv, sizeof(double) * np, hipMemcpyHostToDevice, st));
    KKT_HIP_CHECK(hipMemsetAsync(w.i32, 0, sizeof(int) * 8, st));
    KKT_HIP_CHECK(hipMemsetAsync(S.sc, 0, sizeof(double) * QP_NSC, st));
    int info = 0;
    if (have_init) {
        // ---- the caller's starting point (coneprog.py:2109-2149)
        KKT_HIP_CHECK(hipMemcpyAsync(S.x, x, sizeof(double) * n, hipMemcpyHostToDevice, st));
        if (np > 0) KKT_HIP_CHECK(hipMemcpyAsync(S.y, y, sizeof(double) * np, hipMemcpyHostToDevice, st));
        KKT_HIP_CHECK(hipMemcpyAsync(S.s, s, sizeof(double) * m, hipMemcpyHostToDevice, st));
        KKT_HIP_CHECK(hipMemcpyAsync(S.z, z, sizeof(double) * m, hipMemcpyHostToDevice, st));
        qp_launch_symm(S, S.s, st);
        qp_launch_symm(S, S.z, st);
        qp_launch_start(S, st, 1);
    } else {
        // ---- starting point with W = I (coneprog.py:2054-2106)
        qp_launch_unit_scaling(S, st);
        if (int e = factor(&info)) return e;
        if (info > 0) { set_last_error("coneqp: Rank(A) < p or Rank([P; A; G]) < n"); return 1; }
        hipLaunchKernelGGL(axpby_kernel, dim3((n + 255) / 256), dim3(256), 0, st, S.x, S.q, -1.0, (int64_t)n);
        if (np > 0) KKT_HIP_CHECK(hipMemcpyAsync(S.y, S.b, sizeof(double) * np, hipMemcpyDeviceToDevice, st));
        KKT_HIP_CHECK(hipMemcpyAsync(S.z, S.h, sizeof(double) * m, hipMemcpyDeviceToDevice, st));
        if (int e = solve(S.x, S.y, S.z)) return e;
        qp_launch_start(S, st);
    }
    int it = 0;
    for (; it <= maxiters; ++it) {
        if (int e = products(S.x, S.y, S.z)) return e;
        KKT_HIP_CHECK(hipMemsetAsync(S.nactive, 0, sizeof(int), st));
        qp_launch_residual(S, it, maxiters, abstol, reltol, feastol, st);
        KKT_HIP_CHECK(hipMemcpyAsync(w.pinned, S.nactive, sizeof(int), hipMemcpyDeviceToHost, st));
        KKT_HIP_CHECK(hipStreamSynchronize(st));
        {   // coneprog.py:2206-2208: pcost dcost gap pres dres
            static const int idx[5] = {QP_PCOST, QP_DCOST, QP_GAP_OUT, QP_PRES, QP_DRES};
            if (int e = report_progress(hs, it, S.sc, QP_NSC, idx, 5, -1, -1)) return e;
        }
        if (int e = loop_check_handoff(hs)) return e;
        if (w.pinned[0] == 0) break;
        if (int e = factor(&info)) return e;
        if (info > 0 && it == 0) { set_last_error("coneqp: Rank(A) < p or Rank([P; A; G]) < n"); return 1; }
        qp_launch_singular(S, d_info, it, st);
        if (info > 0) break;
        for (int i01 = 0; i01 < 2; ++i01) {
            qp_launch_build(S, D, Wsave, i01, refinement > 0 ? 1 : 0, st);
            qp_launch_f4pre(S, D, st);
            if (int e = solve(D.x, D.y, D.z)) return e;
            qp_launch_f4post(S, D, st);
            for (int r = 0; r < refinement; ++r) {                       // coneprog.py:2330-2345
                qp_launch_copy(S, W2, Wsave, st);
                qp_launch_res_a(S, D, st);
                if (int e = products(D.x, D.y, S.wz3)) return e;
                qp_launch_res_b(S, D, W2, st);
                qp_launch_f4pre(S, W2, st);
                if (int e = solve(W2.x, W2.y, W2.z)) return e;
                qp_launch_f4post(S, W2, st);
                qp_launch_add(S, D, W2, st);
            }
            qp_launch_step(S, D, i01, st);
        }
        qp_launch_update(S, D, st);
    }
    KKT_HIP_CHECK(hipMemcpyAsync(x, S.x_out, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    if (np > 0) KKT_HIP_CHECK(hipMemcpyAsync(y, S.y_out, sizeof(double) * np, hipMemcpyDeviceToHost, st));
    KKT_HIP_CHECK(hipMemcpyAsync(s, S.s_out, sizeof(double) * m, hipMemcpyDeviceToHost, st));
    KKT_HIP_CHECK(hipMemcpyAsync(z, S.z_out, sizeof(double) * m, hipMemcpyDeviceToHost, st));
    KKT_HIP_CHECK(hipMemcpyAsync(status, S.status, sizeof(int), hipMemcpyDeviceToHost, st));
    KKT_HIP_CHECK(hipMemcpyAsync(iters, S.iters, sizeof(int), hipMemcpyDeviceToHost, st));
    double hsc[QP_NSC];
    KKT_HIP_CHECK(hipMemcpyAsync(hsc, S.sc, sizeof(double) * QP_NSC, hipMemcpyDeviceToHost, st));
    KKT_HIP_CHECK(hipStreamSynchronize(st));
    if (stats) {   // gap, relative gap (1e300 = None), pcost, dcost, pres, dres
        stats[0] = hsc[QP_GAP_OUT]; stats[1] = hsc[QP_RELGAP]; stats[2] = hsc[QP_PCOST]; stats[3] = hsc[QP_DCOST];
        stats[4] = hsc[QP_PRES]; stats[5] = hsc[QP_DRES];
    }
    return 0;
} catch (...) { return kkt_catch("mi355kkt_coneqp_init"); }

float mi355kkt_batch_last_factor_ms(const mi355kkt_batch* b) { return b ? b->t_factor : 0.0f; }

// ---- stand-alone operators -----------------------------------------------------------------------------
struct OpTimer {
    hipEvent_t a = nullptr, b = nullptr;
    float* ms;
    explicit OpTimer(float* m) : ms(m) {
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        (void)hipEventRecord(a, nullptr);
    }
    int finish() {
        (void)hipEventRecord(b, nullptr);
        hipError_t e = hipEventSynchronize(b);
        float t = 0;
        (void)hipEventElapsedTime(&t, a, b);
        if (ms) *ms = t;
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
        if (e != hipSuccess) {
            set_last_error("op failed: %s", hipGetErrorString(e));
            return MI355KKT_EHIP;
        }
        return 0;
    }
};

static int cur_num_cus() {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount;
}

int mi355kkt_op_syrk_scaled(const double* dG, int64_t ldG, int m, int n, const double* ddi, const double* dH,
                            int64_t ldH, double* dS, int64_t ldS, float* ms) try {
    static SyrkPlan plan;   // cached for repeated calls with one shape (profiling loops)
    const int kc = m;
    if (plan.n != n || plan.K != kc || !plan.d_items)
        if (int e = build_syrk_plan(plan, n, kc, cur_num_cus())) return e;
    OpTimer t(ms);
    for (int k0 = 0; k0 < m || k0 == 0; k0 += (kc > 0 ? kc : 1)) {
        if (int e = launch_syrk_scaled(plan, dG + k0, ldG, ddi ? ddi + k0 : nullptr, dS, ldS, k0 == 0 ? dH : dS,
                                       k0 == 0 ? ldH : ldS, nullptr))
            return e;
        if (m == 0) break;
    }
    return t.finish();
} catch (...) { return kkt_catch("mi355kkt_op_syrk_scaled"); }

/* host-only: the symbolic analysis of the sparse engine (ordering + supernodes) for the pattern of H + G'G */
int mi355kkt_op_symbolic(int n, int m, const int64_t* gcolptr, const int64_t* growind, const int64_t* hcolptr,
                         const int64_t* hrowind, int* perm, int64_t* nnzL, int* nsupernodes, int* nlevels) try {
    SparseSymbolic S;
    if (int e = symbolic_analyze(S, n, m, gcolptr, growind, hcolptr, hrowind)) return e;
    if (perm) for (int k = 0; k < n; ++k) perm[k] = S.perm[k];
    if (nnzL) *nnzL = S.nnzL;
    if (nsupernodes) *nsupernodes = S.ns;
    if (nlevels) *nlevels = S.nlevels;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_op_symbolic"); }

/* host-only: the whole plan of the symbolic analysis as one flat int64 array, for the CPU tests that execute the plan in
 * NumPy (tests/test_sparse_plan_cpu.py).  out = header[16] followed by the arrays in the order of the header counts:
 *   header = { n, ns, nlevels, store_doubles, ntargets, ncontrib, order_method, nrows, nchildren, nrelmap, 0... }
 *   perm[n], sn_first[ns+1], sn_rowptr[ns+1], sn_rows[nrows], panel_off[ns+1], upd_off[ns], upd_ld[ns], big[ns],
 *   sn_level[ns], child_ptr[ns+1], child_list[nchildren], relmap_off[ns+1], relmap[nrelmap],
 *   asm_slot[ntargets], asm_ptr[ntargets+1], asm_a[ncontrib], asm_b[ncontrib], asm_r[ncontrib],
 *   level_ptr[nlevels+1], level_sn[ns], level_nsmall[nlevels], vb_ptr[nlevels+1], vb[5 * nvb] (off, h, w, col0, supernode),
 *   heavy_ptr[nlevels+1], heavy[nheavy]    (header[10] = nvb, header[11] = nheavy).
 * Returns the number of int64 entries of the plan (call with cap = 0 to size the buffer), or a negative error code. */
int64_t mi355kkt_test_symbolic_plan(int n, int m, const int64_t* gcolptr, const int64_t* growind, const int64_t* hcolptr,
                                     const int64_t* hrowind, int64_t* out, int64_t cap) try {
    SparseSymbolic S;
    if (int e = symbolic_analyze(S, n, m, gcolptr, growind, hcolptr, hrowind)) return e;
    const int ns = S.ns;
    const int64_t nrows = ns ? S.sn_rowptr[ns] : 0, nch = ns ? S.child_ptr[ns] : 0, nrel = ns ? S.relmap_off[ns] : 0;
    const int64_t nt = (int64_t)S.asm_slot.size(), nc = (int64_t)S.asm_a.size();
    const int64_t nvb = (int64_t)S.vb.size(), nheavy = (int64_t)S.heavy.size(), nl = S.nlevels;
    const int64_t need = 16 + (int64_t)n + 2 * (int64_t)(ns + 1) + nrows + (ns + 1) + 4 * (int64_t)ns + (ns + 1) + nch + (ns + 1) + nrel +
                         nt + (nt + 1) + 3 * nc + (nl + 1) + ns + nl + (nl + 1) + 5 * nvb + (nl + 1) + nheavy;
    if (!out || cap < need) return need;
    int64_t* p = out;
    const int64_t header[16] = {n, ns, S.nlevels, S.store_doubles, nt, nc, S.order_method, nrows, nch, nrel, nvb, nheavy, 0, 0, 0, 0};
    for (int64_t v : header) *p++ = v;
    auto put = [&](const auto& vec, int64_t count) {
        for (int64_t k = 0; k < count; ++k) *p++ = (int64_t)vec[k];
    };
    put(S.perm, n);
    put(S.sn_first, ns + 1);
    put(S.sn_rowptr, ns + 1);
    put(S.sn_rows, nrows);
    put(S.panel_off, ns + 1);
    put(S.upd_off, ns);
    put(S.upd_ld, ns);
    put(S.big, ns);
    put(S.sn_level, ns);
    put(S.child_ptr, ns + 1);
    put(S.child_list, nch);
    put(S.relmap_off, ns + 1);
    put(S.relmap, nrel);
    put(S.asm_slot, nt);
    put(S.asm_ptr, nt + 1);
    put(S.asm_a, nc);
    put(S.asm_b, nc);
    put(S.asm_r, nc);
    put(S.level_ptr, nl + 1);
    put(S.level_sn, ns);
    put(S.level_nsmall, nl);
    put(S.vb_ptr, nl + 1);
    for (const VbDesc& d : S.vb) { *p++ = d.off; *p++ = d.h; *p++ = d.w; *p++ = d.col0; *p++ = d.pad; }
    put(S.heavy_ptr, nl + 1);
    put(S.heavy, nheavy);
    return (p - out == need) ? need : (int64_t)MI355KKT_EINVAL;
} catch (...) { return (int64_t)kkt_catch("mi355kkt_test_symbolic_plan"); }

int mi355kkt_op_cone_scale(int ml, int nq, const int* q, double* dX, int64_t ldX, int ncols, const double* ddi,
                           const double* dv, const double* dbeta, float* ms) try {
    ConeLayout cl;
    std::vector<int> qq(q, q + nq);
    if (int e = cone_layout_build(cl, ml, qq)) return e;
    int rc = 0;
    if (nq > 0) rc = cone_layout_set_beta(cl, dbeta, nullptr);
    OpTimer t(ms);
    if (!rc) rc = launch_cone_scale(cl, dX, ldX, dX, ldX, ncols, ddi, dv, dbeta, 1.0, nullptr);
    if (!rc) rc = t.finish();
    cone_layout_free(cl);
    return rc;
} catch (...) { return kkt_catch("mi355kkt_op_cone_scale"); }

}  // extern "C"
#ifdef MI355KKT_DEBUG
__global__ void hwid_probe_kernel(unsigned* out) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID, all 32 bits
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
    }
    __builtin_amdgcn_s_sleep(127);
}
extern "C" {
/* developer probe: HW_ID / XCC_ID of nblocks one-wave workgroups (out: 2 * nblocks words, host) */
int mi355kkt_debug_hwid(unsigned* out, int nblocks) try {
    unsigned* d = nullptr;
    KKT_HIP_CHECK(DEV_ALLOC(&d, sizeof(unsigned) * 2 * nblocks));
    hipLaunchKernelGGL(hwid_probe_kernel, dim3(nblocks), dim3(64), 0, nullptr, d);
    KKT_HIP_CHECK(memcpy_sync(out, d, sizeof(unsigned) * 2 * nblocks, hipMemcpyDeviceToHost));
    (void)dev_free(d);
    return 0;
} catch (...) { return kkt_catch("mi355kkt_debug_hwid"); }
}  // extern "C"
#endif  // MI355KKT_DEBUG
extern "C" {
/* Host execution of the second-order-cone operations the device-resident loops use (the SAME source, cone_ops.h, compiled
 * for the host): for the CPU parity tests against the reference's misc / misc_solvers functions.  One cone of dimension mk.
 * op: 0 sprod (x := x o y), 1 sinv (x := y o\ x), 2 ssqr (x := y o y), 3 scale2 (x := H(y^{1/2}) x; inverse: arg),
 * 4 scale (x := W x with v = y, beta = w[0]; inverse: arg), 5 jnrm2 -> w[0], 6 compute_scaling (s = x, z = y -> v = w[0:mk],
 * lambda = w[mk:2mk], beta = w[2mk]), 7 update_scaling (s = x, z = y normalised in place; v = w[0:mk], lambda = w[mk:2mk],
 * beta = w[2mk] updated), 8 max_step term ||x1|| - x0 -> w[0]. */
int mi355kkt_test_cone_op_host(int op, int mk, int arg, double* x, double* y, double* w) try {
    if (mk < 1 || !x) return MI355KKT_EINVAL;
    switch (op) {
        case 0: mi355kkt::q_sprod(x, y, mk); break;
        case 1: mi355kkt::q_sinv(x, y, mk); break;
        case 2: mi355kkt::q_ssqr(x, y, mk); break;
        case 3: mi355kkt::q_scale2(y, x, mk, arg != 0); break;
        case 4: mi355kkt::q_scale(x, y, w[0], mk, arg != 0); break;
        case 5: w[0] = mi355kkt::q_jnrm2(x, mk); break;
        case 6: mi355kkt::q_compute_scaling(x, y, w, w + 2 * mk, w + mk, mk); break;
        case 7: mi355kkt::q_update_scaling(x, y, w, w + 2 * mk, w + mk, mk); break;
        case 8: w[0] = mi355kkt::q_nrm1(x, mk) - x[0]; break;
        default: return MI355KKT_EINVAL;
    }
    return 0;
} catch (...) { return kkt_catch("mi355kkt_test_cone_op_host"); }
/* The same for the 's'-block operations (cone_ops_s.h, instantiated with a team of one thread): one block of order m, column-
 * major.  inverse = arg & 1, trans = arg & 2.
 * op: 0 scale (x := W x: r'xr | rxr' (trans) | rti x rti' (inverse) | rti'x rti (both)), 1 sprod (x := (xy + yx)/2),
 * 2 sprod diag = 'D' (x := x o diag(lam); inverse: sinv), 3 scale2 (lam, x; inverse), 4 smallest eigenvalue -> lam[0],
 * 5 eigenvalue decomposition (x := eigenvectors, lam := eigenvalues ascending), 6 compute_scaling (s = x, z = y -> r, rti, lam),
 * 7 update_scaling (Ls = x, Lz = y destroyed; r, rti, lam updated), 8 potrf (x := chol(x), strict upper zeroed). */
int mi355kkt_test_sdp_op_host(int op, int m, int arg, double* x, double* y, double* r, double* rti, double* lam) try {
    if (m < 1 || !x) return MI355KKT_EINVAL;
    const mi355kkt::ParHost par;
    const size_t mm = (size_t)m * m;
    std::vector<double> w(3 * mm + mi355kkt::s_jw_doubles(m, 1) + m);
    double *T1 = w.data(), *T2 = T1 + mm, *T3 = T2 + mm, *jw = T3 + mm, *sg = jw + mi355kkt::s_jw_doubles(m, 1);
    const bool inverse = arg & 1, trans = arg & 2;
    switch (op) {
        case 0: mi355kkt::s_scale_blk(par, x, inverse ? rti : r, m, trans == inverse, T1); break;
        case 1: mi355kkt::s_sprod_blk(par, x, y, m, T1); break;
        case 2: mi355kkt::s_sprod_diag_blk(par, x, lam, m, inverse); break;
        case 3: mi355kkt::s_scale2_blk(par, lam, x, m, inverse); break;
        case 4: lam[0] = mi355kkt::s_min_eig_blk(par, x, m, T1, sg, jw); break;
        case 5: mi355kkt::s_eig_blk(par, x, lam, m, T1, T2, jw); break;
        case 6: return mi355kkt::s_compute_scaling_blk(par, x, y, r, rti, lam, m, T1, T2, T3, jw);
        case 7: mi355kkt::s_update_scaling_blk(par, x, y, r, rti, lam, m, T1, T2, jw); break;
        case 8: return mi355kkt::s_potrf(par, x, m);
        default: return MI355KKT_EINVAL;
    }
    return 0;
} catch (...) { return kkt_catch("mi355kkt_test_sdp_op_host"); }
/* ... and on the DEVICE: one workgroup (team = 0: 1024 threads, the loops' configuration) or one wave (team = 1) runs the
 * operation on copies of the host arrays (any may be NULL where the operation does not use it); results come back in place. */
int mi355kkt_test_sdp_op_device(int op, int m, int arg, int team, double* x, double* y, double* r, double* rti, double* lam) try {
    if (m < 1 || !x || mi355kkt_device_count() < 1) return MI355KKT_EINVAL;
    const size_t mm = (size_t)m * m, nw = 3 * mm + mi355kkt::s_jw_doubles(m, 1024) + 64;
    double* d = nullptr;
    KKT_HIP_CHECK(DEV_ALLOC(&d, sizeof(double) * (4 * mm + (size_t)m + nw + 8)));
    double *dx = d, *dy = dx + mm, *dr = dy + mm, *drti = dr + mm, *dl = drti + mm, *dw = dl + m, *dout = dw + nw;
    KKT_HIP_CHECK(memset_sync(d, 0, sizeof(double) * (4 * mm + (size_t)m + nw + 8)));
    KKT_HIP_CHECK(memcpy_sync(dx, x, sizeof(double) * mm, hipMemcpyHostToDevice));
    if (y) KKT_HIP_CHECK(memcpy_sync(dy, y, sizeof(double) * mm, hipMemcpyHostToDevice));
    if (r) KKT_HIP_CHECK(memcpy_sync(dr, r, sizeof(double) * mm, hipMemcpyHostToDevice));
    if (rti) KKT_HIP_CHECK(memcpy_sync(drti, rti, sizeof(double) * mm, hipMemcpyHostToDevice));
    if (lam) KKT_HIP_CHECK(memcpy_sync(dl, lam, sizeof(double) * m, hipMemcpyHostToDevice));
    if (int e = mi355kkt::sdp_op_debug_launch(op, m, arg, team, dx, dy, dr, drti, dl, dw, dout, nullptr)) { (void)dev_free(d); return e; }
    KKT_HIP_CHECK(hipDeviceSynchronize());
    KKT_HIP_CHECK(memcpy_sync(x, dx, sizeof(double) * mm, hipMemcpyDeviceToHost));
    if (y) KKT_HIP_CHECK(memcpy_sync(y, dy, sizeof(double) * mm, hipMemcpyDeviceToHost));
    if (r) KKT_HIP_CHECK(memcpy_sync(r, dr, sizeof(double) * mm, hipMemcpyDeviceToHost));
    if (rti) KKT_HIP_CHECK(memcpy_sync(rti, drti, sizeof(double) * mm, hipMemcpyDeviceToHost));
    double ret = 0.0;
    KKT_HIP_CHECK(memcpy_sync(&ret, dout, sizeof(double), hipMemcpyDeviceToHost));
    if (lam) {
        KKT_HIP_CHECK(memcpy_sync(lam, dl, sizeof(double) * m, hipMemcpyDeviceToHost));
        if (op == 4) lam[0] = ret;
    }
    (void)dev_free(d);
    return (op == 6 || op == 8) ? (int)ret : 0;
} catch (...) { return kkt_catch("mi355kkt_test_sdp_op_device"); }

/* The same operations run by a TEAM of nt host threads (pthread barrier as the team barrier): the SPMD form of cone_ops_s.h
 * with real concurrency between the threads of a team, as on the device (workgroup teams of 1024, wave teams of 64), for the
 * CPU tests -- a data race or a missing barrier shows up here (and under ThreadSanitizer) without a GPU. */
}  // extern "C"
#include <pthread.h>
#include <thread>
namespace {
struct ParThreads {
    int id, n;
    pthread_barrier_t* bar;
    double* red;
    int tid() const { return id; }
    int nt() const { return n; }
    void sync() const { pthread_barrier_wait(bar); }
    double sum(double v) const {
        red[id] = v;
        sync();
        double a = 0.0;
        for (int i = 0; i < n; ++i) a += red[i];
        sync();
        return a;
    }
    double max(double v) const {
        red[id] = v;
        sync();
        double a = red[0];
        for (int i = 1; i < n; ++i) a = a > red[i] ? a : red[i];
        sync();
        return a;
    }
    void jacobi(double* G, double* V, int m, double* jw) const { mi355kkt::s_jacobi_rotations(*this, G, V, m, jw); }
};
}  // namespace
extern "C" {
int mi355kkt_test_sdp_op_host_team(int op, int m, int arg, int nt, double* x, double* y, double* r, double* rti, double* lam) try {
    if (m < 1 || !x || nt < 1 || nt > 1024) return MI355KKT_EINVAL;
    const size_t mm = (size_t)m * m;
    std::vector<double> w(3 * mm + mi355kkt::s_jw_doubles(m, nt) + m), red(nt);
    double *T1 = w.data(), *T2 = T1 + mm, *T3 = T2 + mm, *jw = T3 + mm, *sg = jw + mi355kkt::s_jw_doubles(m, nt);
    const bool inverse = arg & 1, trans = arg & 2;
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, nullptr, (unsigned)nt);
    std::vector<int> rc(nt, 0);
    std::vector<double> ret(nt, 0.0);
    auto body = [&](int id) {
        const ParThreads par{id, nt, &bar, red.data()};
        switch (op) {
            case 0: mi355kkt::s_scale_blk(par, x, inverse ? rti : r, m, trans == inverse, T1); break;
            case 1: mi355kkt::s_sprod_blk(par, x, y, m, T1); break;
            case 2: mi355kkt::s_sprod_diag_blk(par, x, lam, m, inverse); break;
            case 3: mi355kkt::s_scale2_blk(par, lam, x, m, inverse); break;
            case 4: ret[id] = mi355kkt::s_min_eig_blk(par, x, m, T1, sg, jw); break;
            case 5: mi355kkt::s_eig_blk(par, x, lam, m, T1, T2, jw); break;
            case 6: rc[id] = mi355kkt::s_compute_scaling_blk(par, x, y, r, rti, lam, m, T1, T2, T3, jw); break;
            case 7: mi355kkt::s_update_scaling_blk(par, x, y, r, rti, lam, m, T1, T2, jw); break;
            case 8: rc[id] = mi355kkt::s_potrf(par, x, m); break;
            default: rc[id] = MI355KKT_EINVAL;
        }
    };
    std::vector<std::thread> th;
    for (int id = 1; id < nt; ++id) th.emplace_back(body, id);
    body(0);
    for (auto& t : th) t.join();
    pthread_barrier_destroy(&bar);
    if (op == 4) lam[0] = ret[0];
    return rc[0];
} catch (...) { return kkt_catch("mi355kkt_test_sdp_op_host_team"); }
/* The static SYRK schedule for an n x n result contracted over K on a device with num_cus compute units, as plain integers
 * (host only): out[8 * i + 0..7] = ti, tj, k0, k1, slot, first, nparts, next of segment i; the first `*nlaunch` segments are the
 * launch's workgroups in launch order, the others continuation segments reached through `next` (1 + index).  For the CPU tests
 * of the plan's invariants.  Returns the number of segments (<= max_items are written). */
int mi355kkt_test_syrk_plan(int n, int K, int num_cus, int allow_split, int* out, int max_items, int* nslabs, int* nsplit,
                            int* nlaunch) try {
    std::vector<mi355kkt::SyrkItem> items, split_tiles;
    int ns = 0, nl = 0;
    mi355kkt::make_syrk_items(n, K, num_cus, allow_split != 0, items, nl, split_tiles, ns);
    for (size_t i = 0; i < items.size() && (int)i < max_items; ++i) {
        const mi355kkt::SyrkItem& it = items[i];
        int* o = out + 8 * i;
        o[0] = it.ti; o[1] = it.tj; o[2] = it.k0; o[3] = it.k1; o[4] = it.slot; o[5] = it.first; o[6] = it.nparts; o[7] = it.next;
    }
    if (nslabs) *nslabs = ns;
    if (nsplit) *nsplit = (int)split_tiles.size();
    if (nlaunch) *nlaunch = nl;
    return (int)items.size();
} catch (...) { return kkt_catch("mi355kkt_test_syrk_plan"); }
/* Fill-reducing ordering of a symmetric pattern (host only, for the CPU tests of csrc/ordering.cpp): colptr/rowind = CSC
 * pattern of any part of the matrix that contains each off-diagonal pair at least once; method 0 = choose, 1 = nested
 * dissection, 2 = approximate minimum degree.  perm[new] = old.  stats[0..6] = method chosen, nnz(L) and flops of the
 * dissection candidate, nnz(L) and flops of the minimum-degree candidate, supernodal tree heights of the two; stats[7] = 1
 * when the two column-count algorithms agree on the returned order (and with the tree / counts the analysis keeps). */
int mi355kkt_test_ordering(int n, const int64_t* colptr, const int64_t* rowind, int method, int* perm, double* stats) try {
    if (n < 0 || !colptr || !perm) return MI355KKT_EINVAL;
    mi355kkt::Graph adj(n);
    for (int j = 0; j < n; ++j)
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            const int i = (int)rowind[k];
            if (i < 0 || i >= n) return MI355KKT_EINVAL;
            if (i != j) { adj[i].push_back(j); adj[j].push_back(i); }
        }
    for (auto& a : adj) {
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
    }
    std::vector<int> order;
    mi355kkt::OrderingInfo info;
    mi355kkt::fill_reducing_ordering(adj, order, method, &info);
    if ((int)order.size() != n) return MI355KKT_EINVAL;
    std::copy(order.begin(), order.end(), perm);
    if (stats) {
        stats[0] = info.method; stats[1] = (double)info.nnz_nd; stats[2] = info.flops_nd; stats[3] = (double)info.nnz_amd;
        stats[4] = info.flops_amd; stats[5] = info.levels_nd; stats[6] = info.levels_amd;
        // cross-check of the column counts (skeleton / LCA algorithm) against the row-subtree walk, on the final order
        std::vector<int> par;
        std::vector<int64_t> fast, slow;
        mi355kkt::etree_and_counts(adj, order, par, fast);
        mi355kkt::column_counts_by_row_subtrees(adj, order, par, slow);
        stats[7] = (fast == slow && fast == info.colcount && par == info.parent) ? 1.0 : 0.0;
    }
    return 0;
} catch (...) { return kkt_catch("mi355kkt_test_ordering"); }
#ifdef MI355KKT_DEBUG       // include/mi355kkt_debug.h: process-global developer switches, never in a production build
int mi355kkt_debug_tile_ts(void* dptr) { return mi355kkt::set_tile_ts((long long*)dptr); }
int mi355kkt_debug_wide_ts(void* dptr) { return mi355kkt::set_wide_ts((long long*)dptr); }
int mi355kkt_debug_potf2_ts(void* dptr) { return mi355kkt::set_potf2_ts((long long*)dptr); }
int mi355kkt_debug_syrk_skip(int mask) { return mi355kkt::set_syrk_skip(mask); }
#endif
/* test / developer knobs (csrc/knobs.h): value == NULL unsets one, name == NULL unsets all */
int mi355kkt_test_set_knob(const char* name, const char* value) try {
    return mi355kkt::set_dev_knob(name, value);
} catch (...) { return kkt_catch("mi355kkt_test_set_knob"); }
/* writes the ring of the library's last 65536 device allocations / releases to `path` when the process receives SIGABRT (the
 * HIP runtime abort()s on one of its own threads after reporting a GPU memory fault), then lets the previous handler run */
int mi355kkt_test_install_abort_dump(const char* path) try {
    return mi355kkt::install_abort_dump(path) == 0 ? 0 : MI355KKT_EINVAL;
} catch (...) { return kkt_catch("mi355kkt_test_install_abort_dump"); }
int mi355kkt_test_guard_violations(void) { return mi355kkt::guard_violations(); }
/* allocates ndoubles doubles through the library's allocator and reads the element at index `at` from a kernel (at >= ndoubles:
 * out of bounds on purpose -- under the knob MI355KKT_ALLOC_GUARD that must be a GPU memory fault, which ends the process);
 * *out = the value read */
int mi355kkt_test_guard_probe(int ndoubles, int at, double* out) try {
    if (ndoubles < 1 || at < 0 || !out) return MI355KKT_EINVAL;
    double *blk = nullptr, *res = nullptr;
    KKT_HIP_CHECK(DEV_ALLOC(&blk, sizeof(double) * (size_t)ndoubles));
    if (DEV_ALLOC(&res, sizeof(double)) != hipSuccess) { (void)dev_free(blk); return MI355KKT_ENOMEM; }
    hipLaunchKernelGGL(row_gather_kernel, dim3(1), dim3(64), 0, nullptr, blk + at, (int64_t)0, 1, res);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, res, sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) {          // (a faulted queue: releasing the blocks would only fail again)
        set_last_error("guard probe: %s", hipGetErrorString(e));
        return MI355KKT_EHIP;
    }
    (void)dev_free(blk);
    (void)dev_free(res);
    return 0;
} catch (...) { return kkt_catch("mi355kkt_test_guard_probe"); }
int mi355kkt_test_touches_brk_heap(const void* ptr, size_t bytes) try {
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    return touches_brk_heap(a, a + bytes) ? 1 : 0;
} catch (...) { return kkt_catch("mi355kkt_test_touches_brk_heap"); }
int mi355kkt_test_throw(int kind) try {
    if (kind == 0) throw std::bad_alloc();
    if (kind == 1) throw std::runtime_error("requested by the caller");
    if (kind == 2) throw 42;
    return 0;
} catch (...) { return kkt_catch("mi355kkt_test_throw"); }

int mi355kkt_op_mfma_f64_peak(int iters, float* tflops) { return run_mfma_f64_peak(iters, cur_num_cus(), tflops); }

int mi355kkt_op_potrf(double* dA, int64_t ldA, int n, int* info, float* ms) try {
    PotrfWork w;
    if (int e = potrf_work_init(w)) return e;
    if (int e = potrf_work_reserve(w, n)) { potrf_work_free(w); return e; }
    OpTimer t(ms);
    int rc = launch_potrf(dA, ldA, n, w, nullptr);
    if (!rc) rc = t.finish();
    if (!rc) {
        if (memcpy_sync(w.h_info, w.d_info, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) rc = MI355KKT_EHIP;
        if (info) *info = *w.h_info;
    }
    potrf_work_free(w);
    return rc;
} catch (...) { return kkt_catch("mi355kkt_op_potrf"); }

int mi355kkt_op_trsm_lower(const double* dL, int64_t ldL, int n, double* dX, int64_t ldX, int nrhs, int trans,
                           float* ms) try {
    OpTimer t(ms);
    if (int e = launch_trsm_lower(dL, ldL, n, dX, ldX, nrhs, trans, nullptr)) return e;
    return t.finish();
} catch (...) { return kkt_catch("mi355kkt_op_trsm_lower"); }

int mi355kkt_op_gemv_t_scaled(const double* dG, int64_t ldG, int m, int n, const double* dw, const double* dz,
                              double* dzs, double* dy, float* ms) try {
    double* work = nullptr;
    KKT_HIP_CHECK(DEV_ALLOC(&work, sizeof(double) * (size_t)(m > 0 ? m : 1)));
    OpTimer t(ms);
    int rc = launch_gemv_t_scaled(dG, ldG, m, n, dw, dz, dzs, dy, work, nullptr);
    if (!rc) rc = t.finish();
    (void)dev_free(work);
    return rc;
} catch (...) { return kkt_catch("mi355kkt_op_gemv_t_scaled"); }

int mi355kkt_op_gemv_n_scaled(const double* dG, int64_t ldG, int m, int n, const double* dw, const double* dx,
                              const double* dzs, double* dz, float* ms) try {
    double* work = nullptr;
    KKT_HIP_CHECK(DEV_ALLOC(&work, sizeof(double) * gemv_work_doubles(m, n)));
    OpTimer t(ms);
    int rc = launch_gemv_n_scaled(dG, ldG, m, n, dw, dx, dzs, dz, 1.0, -1.0, work, nullptr);
    if (!rc) rc = t.finish();
    (void)dev_free(work);
    return rc;
} catch (...) { return kkt_catch("mi355kkt_op_gemv_n_scaled"); }

}  // extern "C"

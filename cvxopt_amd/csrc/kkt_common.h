// Shared declarations for the MI355X (gfx950) KKT backend.  Internal header (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include "knobs.h"
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace mi355kkt {

// ---- error plumbing ---------------------------------------------------------------------------
void set_last_error(const char* fmt, ...);

#define KKT_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::mi355kkt::set_last_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,     \
                                       hipGetErrorString(_e));                           \
            return -2; /* MI355KKT_EHIP */                                               \
        }                                                                                \
    } while (0)

// ---- host-synchronous copies / fills ---------------------------------------------------------------
// hipMemcpy(device-to-device) and hipMemset on device memory return BEFORE the work is done: they are only ordered on the legacy
// stream, and every handle computes on its own hipStreamNonBlocking stream, which the legacy stream does not order.  (Found by
// tests/test_gpu_stress.py in round 4: a d2d copy of a right-hand side raced with the solve launched right after it once the
// legacy stream was busy.)  Everything in this library that means "copy / fill now, then launch on another stream" goes through
// these: the plain call followed by a wait on the legacy stream.  Setup-time cost only (never inside factor / solve).
inline hipError_t memcpy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    hipError_t e = hipMemcpy(dst, src, bytes, kind);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}
inline hipError_t memcpy2d_sync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                                hipMemcpyKind kind) {
    hipError_t e = hipMemcpy2D(dst, dpitch, src, spitch, width, height, kind);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}
inline hipError_t memset_sync(void* p, int value, size_t bytes) {
    hipError_t e = hipMemset(p, value, bytes);
    return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}

// ---- device allocations (devmem.cpp) ---------------------------------------------------------------
// Every device allocation of this library goes through DEV_ALLOC / dev_free: allocate, clear to zero on a private stream, wait.
// A fresh process gets zero pages from the driver anyway; a long-lived one gets whatever an earlier handle left in the recycled
// block -- the state of a handle must not depend on which of the two it is.  Test knobs (mi355kkt_test.h) turn the
// clear into 0xff poison (MI355KKT_ALLOC_POISON), drop it (MI355KKT_ALLOC_RAW) or put every block at the end of its own mapping
// so that an out-of-bounds access faults at once (MI355KKT_ALLOC_GUARD).  The call site is recorded for the abort dump.
hipError_t dev_alloc(void** p, size_t bytes, const char* file, int line);
hipError_t dev_free(void* p);
int install_abort_dump(const char* path);
int guard_violations();          // MI355KKT_ALLOC_GUARD: overwritten poison found so far (an out-of-bounds write)
template <class T>
inline hipError_t dev_alloc_typed(T** p, size_t bytes, const char* file, int line) {
    return dev_alloc(reinterpret_cast<void**>(p), bytes, file, line);
}
#define DEV_ALLOC(ptr_, bytes_) ::mi355kkt::dev_alloc_typed((ptr_), (bytes_), __FILE__, __LINE__)

// ---- tile geometry of the FP64 MFMA kernels -----------------------------------------------------
constexpr int TILE = 128;      // C tile (both dims) owned by one 256-thread workgroup
constexpr int BK = 16;         // k-depth staged per LDS buffer
constexpr int LDT_K = BK + 2;  // K-major LDS layout: [idx][k], stride 18 doubles  (ds_read_b64 conflict-free)
constexpr int LDT_M = TILE + 16;  // M-major LDS layout: [k][idx], stride 144 doubles (conflict-free)
constexpr int STAGE_DOUBLES = TILE * LDT_K;   // == BK * LDT_M == 2304 doubles per operand per stage
static_assert(TILE * LDT_K == BK * LDT_M, "both LDS layouts must use the same footprint");

struct BatchStrides { int64_t a = 0, b = 0, c = 0, d = 0; };   // element strides between batched problems

struct SyrkItem {  // one contraction segment of one tile in the scaled SYRK
    int ti, tj;    // tile row (C rows, i) / tile column (C cols, j); ti >= tj
    int k0, k1;    // contraction range [k0, k1)
    int slot;      // <0: final (write C = P + acc); >=0: partial slab index
    int first;     // for partial items: index of first slab of this tile; count in `nparts`
    int nparts;
    int next;      // 0: the workgroup is done after this segment; else 1 + index of the segment it continues with (stream-K)
};

struct SyrkPlan {
    int n = 0, K = 0;
    int nitems = 0;                    // workgroups of the launch (= first segments); items beyond are continuation segments
    int nitems_total = 0, nslabs = 0, nsplit_tiles = 0;
    SyrkItem* d_items = nullptr;       // device copy, XCD-friendly order
    SyrkItem* d_split_tiles = nullptr; // one entry per split tile (first/nparts used by the reducer)
    double* d_slabs = nullptr;         // nslabs * TILE*TILE doubles
};

int build_syrk_plan(SyrkPlan& plan, int n, int K, int num_cus, bool allow_split = true);
// items: the launch's workgroups first (nlaunch of them), continuation segments after
void make_syrk_items(int n, int K, int num_cus, bool allow_split, std::vector<SyrkItem>& items, int& nlaunch,
                     std::vector<SyrkItem>& split_tiles, int& nslabs);
void free_syrk_plan(SyrkPlan& plan);

// C(lower) = P(lower) + Gs' Gs with Gs = diag(di) G   (di == nullptr: no scaling; P == nullptr: 0)
// kernel_events (optional): two events recorded immediately around the syrk_tn_kernel launch.
int launch_syrk_scaled(const SyrkPlan& plan, const double* G, int64_t ldg, const double* di,
                       double* C, int64_t ldc, const double* P, int64_t ldp, hipStream_t st,
                       hipEvent_t* kernel_events = nullptr, int nbatch = 1, BatchStrides bs = BatchStrides());
// (batched: strides a = G, b = di, c = C, d = P between consecutive problems, along blockIdx.z)

// C(lower tiles of an nrows x nrows block) -= A A' where A is nrows x K (column-major, lda)
int launch_syrk_nt_update(double* C, int64_t ldc, const double* A, int64_t lda, int nrows, int K,
                          hipStream_t st, int nbatch = 1, int64_t bstride = 0, bool one_wg_per_cu = false);
// C (M x N, all tiles) -= A B'  with A: M x K (lda), B: N x K (ldb)
int launch_gemm_nt_update(double* C, int64_t ldc, const double* A, int64_t lda, const double* B,
                          int64_t ldb, int M, int N, int K, hipStream_t st, int nbatch = 1, int64_t bstride = 0);

int run_mfma_f64_peak(int iters, int num_cus, float* tflops);

// ---- dense Cholesky -----------------------------------------------------------------------------
// One frontal matrix of a level-batched ("variable batched") partial factorisation: h x h, column-major with ld = h,
// at base + off; only its first w columns are factored, the trailing block receives the Schur update.
struct VbDesc { int64_t off; int h, w, col0, pad; };

struct PotrfWork {
    int* d_info = nullptr;   // device int: 0 ok, >0 first failing pivot (1-based, LAPACK convention)
    int* h_info = nullptr;   // pinned host mirror
    double* d_dinv = nullptr; // inverses of the 16x16 diagonal blocks of the current panel (potf2 -> trsm)
    // persistent tile kernel: ticket / abort / per-block-row progress words, inverses of every column's diagonal blocks
    void* d_ctl = nullptr;
    double* d_linv_all = nullptr;
    int linv_tiles = 0;
    // inverses of the 128 x 128 diagonal blocks of the last factor produced by the tile kernel (column-major, one after the
    // other); valid for the matrix `minv_of` of order `minv_n` (0: not available)
    double* d_minv = nullptr;
    int minv_n = 0;
    const double* minv_of = nullptr;
    // round 6 (trsv512.hip): 512 x 512 inverses of the diagonal blocks (per 512-block M then M'), formed from d_minv after the
    // factorisation (launch_block_inverse512); valid for matrix `m512_of` of order `m512_n`.  d_gran512: the wide solves' granules
    double* d_m512 = nullptr;
    double* d_m512_scratch = nullptr;
    unsigned long long* d_gran512 = nullptr;
    int m512_blocks = 0;
    unsigned m512_launches = 0;       // formation launches so far (their stage counters only ever grow)
    int m512_n = 0;
    int m512_shape_n = 0;             // the order whose zero pattern d_m512 currently has
    const double* m512_of = nullptr;
    // look-ahead: the bulk of each trailing update runs on `side` while the next panel is factored on the main stream
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> ev_panel, ev_bulk;
};
#ifdef MI355KKT_DEBUG          // developer aids, include/mi355kkt_debug.h
int set_wide_ts(long long* dptr);    // 16 int64 stamps (100 MHz) per workgroup written by trsv_wide_kernel (nullptr: off)
int set_tile_ts(long long* dptr);    // 8 int64 stamps per tile written by potrf_tiles_kernel (nullptr: off)
int set_potf2_ts(long long* dptr);   // device buffer of 48 timestamps written by potf2_la_kernel (nullptr: off)
int set_syrk_skip(int v);            // ablation switch of the SYRK (results are wrong when != 0)
#endif
int potrf_work_init(PotrfWork& w);
int potrf_work_reserve(PotrfWork& w, int n);   // state of the persistent tile kernel (allocated lazily otherwise)
void potrf_work_free(PotrfWork& w);
// In-place lower Cholesky of the n x n column-major matrix A (only tril referenced/overwritten).
// Asynchronous on `st`; *w.d_info is updated on device.  Returns 0 or a negative error code.
int launch_potrf(double* A, int64_t lda, int n, PotrfWork& w, hipStream_t st);
int launch_potrf_partial(double* F, int64_t ld, int h, int ncols, PotrfWork& w, hipStream_t st);
// the same for `nfronts` fronts of different sizes in one set of launches (blockIdx.z = front; grids sized for maxh/maxw);
// w.d_info / w.d_dinv must hold nfronts ints / nfronts*2048 doubles; info[z] = global column + 1 of a failing pivot
int launch_potrf_partial_vb(double* base, const VbDesc* d_desc, int nfronts, int maxh, int maxw, PotrfWork& w, hipStream_t st);
int launch_syrk_nt_update_vb(double* base, const VbDesc* d_desc, int nfronts, int k0, int maxh, hipStream_t st);
// one persistent launch for all big fronts of a level: tickets = 4 ints (front, i, j, 0) in the order (j, front, i); prog_off /
// linv_off: per front offsets into the level's progress words / 2048-double inverse blocks
int launch_potrf_tiles_vb(double* base, const VbDesc* d_desc, int nfronts, const void* d_tickets, int ntickets,
                          const int* d_prog_off, const int* d_linv_off, void* d_ctl, unsigned* d_prog, int nprog, double* d_linv,
                          int* d_info, hipStream_t st, bool prezeroed = false);
size_t potrf_tile_ctl_bytes();
// nbatch matrices `bstride` doubles apart; w.d_info / w.d_dinv must hold nbatch ints / nbatch*2048 doubles
int launch_potrf_batched(double* A, int64_t lda, int n, int nbatch, int64_t bstride, PotrfWork& w, hipStream_t st);
int potrf_work_init_batched(PotrfWork& w, int nbatch);

// ---- level-2 pieces of solve() --------------------------------------------------------------------
// zs := w .* z;  y[0:n) += G' (w .* zs)   (G m x n); work >= m doubles
int launch_gemv_t_scaled(const double* G, int64_t ldg, int m, int n, const double* w,
                         const double* z, double* zs, double* y, double* work, hipStream_t st,
                         int nbatch = 1, int64_t sG = 0);   // batched: vectors are contiguous (stride m / n)
// z := alpha * w .* (G x) + beta * zs    (z may alias zs)
int launch_gemv_n_scaled(const double* G, int64_t ldg, int m, int n, const double* w,
                         const double* x, const double* zs, double* z, double alpha, double beta,
                         double* work, hipStream_t st, int nbatch = 1, int64_t sG = 0);
size_t gemv_work_doubles(int m, int n);
// Gx := G x and GTz := G' z in one pass over G (blas2.hip); work: gemv_nt_work_doubles(m, n) doubles per problem
int launch_gemv_nt_fused(const double* G, int64_t ldg, int m, int n, const double* x, const double* z, double* Gx, double* GTz,
                         double* work, hipStream_t st, int nbatch = 1, int64_t sG = 0);
size_t gemv_nt_work_doubles(int m, int n);
// single right-hand side, single launch (needs ceil(n/128) co-resident workgroups: callers check against #CUs);
// gran: 256 u64 per 128-block, zeroed once at allocation; epoch: a fresh non-zero value per launch; err: device int
// copies the strictly lower triangle, transposed, into the strictly upper triangle (needed by the transposed persistent solve)
int launch_mirror_lower(double* A, int64_t lda, int n, hipStream_t st);
// trans != 0 solves L' x = b and REQUIRES the mirrored upper triangle (launch_mirror_lower after the factorisation)
// minv != nullptr: inverses of the 128 x 128 diagonal blocks (per block M then M', 2 x 16384 doubles) from the tile Cholesky
struct TrsvJob { const double* L; int64_t ld; int n; int pad; double* x; };   // one system of a batched launch
int launch_mirror_lower_jobs(const TrsvJob* d_jobs, int njobs, int nmax, hipStream_t st);   // the same for a list of matrices
constexpr int TRSV_JOB_STRIDE = 64;    // granule blocks reserved per job (orders up to 8192)
int launch_trsv_persistent(const double* L, int64_t ldl, int n, double* x, int trans, unsigned int epoch, int* err,
                           hipStream_t st, unsigned long long* gran, const double* minv = nullptr,
                           const TrsvJob* jobs = nullptr, int njobs = 0);
// round 4: the same solve as two pipelined sweeps with two workgroups per block row (blas2.hip, trsv_pair_kernel): needs the
// 128 x 128 inverses (minv), n % 128 == 0, 2 n / 128 co-resident workgroups and 3 n / 128 granule blocks
int launch_trsv_pair(const double* L, int64_t ldl, int n, double* x, int trans, unsigned int epoch, int* err, hipStream_t st,
                     unsigned long long* gran, const double* minv);
// round 6: 512-row hops over all compute units (trsv512.hip).  trsv_wide_rows: rows per workgroup for this order, 0 = not served
// (n >= 1024, a multiple of 128); launch_block_inverse512 after the tile Cholesky of L (needs its 128 x 128
// inverses); launch_trsv_wide: same contract as launch_trsv_pair
int trsv_wide_rows(int n, int num_cus, bool any_order = false);   // any_order: n need not be a multiple of 128 (the sparse root)
int launch_block_inverse512(const double* L, int64_t ldl, int n, PotrfWork& w, hipStream_t st);
int launch_trsv_wide(const double* L, int64_t ldl, int n, double* x, int trans, unsigned int epoch, int* err, hipStream_t st,
                     PotrfWork& w, int rows, int num_cus);
// x := L^-1 x (trans=0) or L^-T x (trans=1), L lower n x n, nrhs right-hand sides (ldx)
int launch_trsm_lower(const double* L, int64_t ldl, int n, double* X, int64_t ldx, int nrhs,
                      int trans, hipStream_t st, int nbatch = 1, int64_t sL = 0, int64_t sX = 0);


// ---- sparse Cholesky (sparse_chol.hip) -------------------------------------------------------------
struct SparseSymbolic {
    int n = 0, m = 0, ns = 0, nlevels = 0;
    int order_method = 0;                         // 1 nested dissection, 2 approximate minimum degree (ordering.h)
    int64_t nnzL = 0;
    double flops = 0.0;
    std::vector<int> perm, iperm;                 // perm[new] = old
    std::vector<int> sn_first;                    // ns + 1
    std::vector<int64_t> sn_rowptr;               // ns + 1
    std::vector<int> sn_rows, sn_parent, sn_level, level_ptr, level_sn;
    std::vector<int64_t> panel_off, upd_off, relmap_off;
    std::vector<int> upd_ld, level_nsmall;
    std::vector<char> big;
    std::vector<VbDesc> vb;                       // big fronts, level by level (same order as level_sn's big part)
    std::vector<int> vb_ptr, vb_maxh, vb_maxw;    // per level
    // persistent tile kernel over the big fronts of a level: tickets (front, i, j, 0), per level ranges, per front offsets
    std::vector<int> tv_tickets, tv_ptr, tv_prog_off, tv_linv_off, tv_nprog;
    int tv_prog_max = 0, tv_linv_max = 0;
    int vb_maxcount = 0;
    std::vector<int> heavy, heavy_ptr, heavy_maxhu, heavy_maxw;   // per level: supernodes with large off-diagonal panels
    std::vector<int64_t> ea_off;                                  // extend-add tile-boundary tables of the children of big fronts
    std::vector<int> ea_lb;
    std::vector<int> wide, wide_ptr;                              // per level: supernodes wider than 256 columns (dense multi-workgroup solves)
    int wide_threshold = 128;   // sp_wide_threshold() as read ONCE by the symbolic analysis: the plan and the kernels' views must agree
    int64_t store_doubles = 0;
    std::vector<int> child_ptr, child_list, relmap;
    std::vector<int64_t> asm_slot, asm_ptr;       // numeric assembly: one entry per structural nonzero of S
    std::vector<int> asm_a, asm_b, asm_r;
};
struct SparseEngine {
    SparseSymbolic sym;
    int n = 0, m = 0;
    int *d_sn_first = nullptr, *d_sn_rows = nullptr, *d_child_ptr = nullptr, *d_child_list = nullptr, *d_relmap = nullptr,
        *d_level_sn = nullptr, *d_asm_a = nullptr, *d_asm_b = nullptr, *d_asm_r = nullptr, *d_perm = nullptr,
        *d_gri = nullptr, *d_gci = nullptr, *d_gnzmap = nullptr, *d_info = nullptr, *h_info = nullptr, *d_upd_ld = nullptr,
        *d_heavy = nullptr, *d_hci = nullptr, *d_hmap = nullptr, *d_iperm = nullptr;
    VbDesc* d_vb = nullptr;
    int *d_tv_tickets = nullptr, *d_tv_prog_off = nullptr, *d_tv_linv_off = nullptr;
    unsigned* d_tv_prog = nullptr;
    void* d_tv_ctl = nullptr;
    // one zero-initialised state buffer per factorisation: per level a TileCtl, the progress words and the per-front info words
    // of the persistent tile kernel (d_tv_ctl / d_tv_prog alias into it)
    unsigned char* d_tv_state = nullptr;
    size_t tv_state_bytes = 0;
    std::vector<size_t> tv_ctl_at, tv_prog_at;      // byte offsets per level
    size_t tv_info_at = 0;                          // byte offset of the info words (one per big front, all levels)
    double* d_tv_linv = nullptr;
    PotrfWork pw_vb;
    int64_t *d_sn_rowptr = nullptr, *d_panel_off = nullptr, *d_upd_off = nullptr, *d_relmap_off = nullptr,
            *d_asm_slot = nullptr, *d_asm_ptr = nullptr, *d_gcp = nullptr, *d_grp = nullptr, *d_rem_off = nullptr,
            *d_hrp = nullptr;
    double *d_gv = nullptr, *d_hv = nullptr, *d_rem = nullptr, *d_panels = nullptr, *d_upd = nullptr, *d_xp = nullptr,
           *d_rem_multi = nullptr;
    int rem_multi_cols = 0;
    // hand-off state of the persistent dense triangular solve (blas2.hip) used for the wide supernodes: borrowed from the handle
    unsigned long long* t_gran = nullptr;
    int* t_err = nullptr;
    unsigned int* t_epoch = nullptr;
    int t_njobs_max = 0;                       // jobs the borrowed flags / granules have room for
    int t_num_cus = 0;                         // compute units of the device (co-residency limit of the two-sweep solves)
    int dense_root_level = -1;                 // level whose single big front is factored by the dense tile kernel (-1: none)
    int64_t* d_ea_off = nullptr;
    int* d_ea_lb = nullptr;
    std::vector<int> lvl_small;                // per level: supernodes that take the one-wave kernels (sp_fwd_small_kernel)
    int64_t* d_zero_off = nullptr;             // the panels as chunks (sp_zero_chunks_kernel clears them before every factorisation)
    int* d_zero_len = nullptr;
    int n_zero_chunks = 0;
    TrsvJob* d_wide_jobs = nullptr;            // one per wide supernode, in the order of sym.wide (x = d_xp + first column)
    int* d_wide = nullptr;                     // sym.wide on the device
    std::vector<int> wide_maxw;                // per level
};
int symbolic_analyze(SparseSymbolic& S, int n, int m, const int64_t* gcp, const int64_t* gri, const int64_t* hcp,
                     const int64_t* hri);
int sparse_engine_create(SparseEngine& E, int n, int m, const int64_t* gcp, const int64_t* gri, const double* gv,
                         const int64_t* hcp, const int64_t* hri, const double* hv);
void sparse_engine_free(SparseEngine& E);
int sparse_engine_factor(SparseEngine& E, const double* d_di, hipStream_t st, int* info);
int sparse_engine_solve(SparseEngine& E, double* d_x, hipStream_t st);
int sparse_engine_forward(SparseEngine& E, const double* d_in, double* d_out_perm, hipStream_t st);   // E.d_xp = L^-1 P b
int sparse_engine_backward(SparseEngine& E, double* d_out, hipStream_t st);
int sparse_engine_forward_rows(SparseEngine& E, const double* d_A, int64_t lda, int nrhs, double* d_out, hipStream_t st);
// rows of a sparse A (CSR on the device) as right-hand sides, `chunk` at a time
int sparse_engine_forward_rows_csr(SparseEngine& E, const int64_t* d_rp, const int* d_ci, const double* d_v, int nrhs, double* d_out,
                                   hipStream_t st, int chunk = 256);                           // d_out = P' L^-T E.d_xp
int sparse_engine_product(SparseEngine& E, int which, int trans, const double* d_in, double* d_out, hipStream_t st);
int sparse_engine_products(SparseEngine& E, const double* d_x, const double* d_z, double* d_Gx, double* d_GTz, double* d_Px,
                           hipStream_t st);
int sparse_engine_gemv_t(SparseEngine& E, const double* d_w, const double* d_z, double* d_zs, double* d_zss, double* d_x,
                         hipStream_t st);
int sparse_engine_gemv_n(SparseEngine& E, const double* d_w, const double* d_x, const double* d_zs, double* d_z,
                         hipStream_t st);

// ---- second-order-cone scaling --------------------------------------------------------------------
struct ConeLayout {
    int ml = 0, nq = 0, vlen = 0, n_small = 0, n_large = 0;
    int *d_off = nullptr, *d_dim = nullptr, *d_voff = nullptr;         // per cone (original numbering)
    int *d_small_ids = nullptr, *d_large_ids = nullptr;                // cones of dimension <= 32 / > 32
    int *d_s_off = nullptr, *d_s_dim = nullptr, *d_s_voff = nullptr;   // compacted descriptors of the small cones
    double* d_s_beta = nullptr;                                        // beta gathered for the small cones
    // semidefinite blocks
    int ns = 0, lq_rows = 0, cdim_packed = 0, rlen = 0, s_maxn = 0;
    int *d_sdim = nullptr, *d_soff = nullptr, *d_spoff = nullptr, *d_sroff = nullptr;
    std::vector<int> h_sdim, h_soff, h_spoff, h_sroff;                 // host copies (the MFMA path is launched per large block)
    mutable double* d_cong = nullptr;                                  // scratch of launch_sdp_congruence, grown on demand
    mutable size_t cong_doubles = 0;
};
int launch_sdp_congruence(const double* d_rti, int m, const double* in, int64_t ldi, double* out, int64_t ldo, int ncols,
                          double extra, double* scratch, size_t scratch_doubles, hipStream_t st);
int cone_layout_build_s(ConeLayout& cl, int lq_rows, const std::vector<int>& s);
// out(packed rows of the 's' blocks, ncols columns) = extra * pack(rti' mat(in) rti)
int launch_sdp_scale_pack(const ConeLayout& cl, const double* in, int64_t ldi, double* out, int64_t ldo, int ncols,
                          const double* d_rti, double extra, hipStream_t st);
int launch_sdp_unpack(const ConeLayout& cl, const double* packed, double* out, hipStream_t st);
int cone_layout_build(ConeLayout& cl, int ml, const std::vector<int>& q);
void cone_layout_free(ConeLayout& cl);
int cone_layout_set_beta(ConeLayout& cl, const double* d_beta, hipStream_t st);
// out = extra * W^-T in on the l + q rows, for ncols columns (in/out may alias)
int launch_cone_scale(const ConeLayout& cl, const double* in, int64_t ldi, double* out, int64_t ldo, int ncols,
                      const double* d_di, const double* d_v, const double* d_beta, double extra, hipStream_t st);
// The same for a batch of problems of one shape, each with its own scaling: problem b reads in + b sIn, writes out + b sOut
// (in == out allowed), with di + b cdim ('l' rows: the first ml entries), v + b max(sumq, 1), beta + b max(nq, 1).
// qoff / qdim: device arrays, row offset and dimension of every second-order cone (shared by all problems); large_ids: the
// nlarge cones of more than 32 rows.
int launch_batch_cone_scale(const double* in, int64_t ldi, int64_t sIn, double* out, int64_t ldo, int64_t sOut, int ncols,
                            int nbatch, int cdim, int ml, int nq, int sumq, const int* d_qoff, const int* d_qdim,
                            const int* d_large_ids, int nlarge, const double* d_di, const double* d_v, const double* d_beta,
                            hipStream_t st);

// ---- device-resident LP-cone coneqp loop for a batch (batch_ipm.hip) --------------------------------
struct IpmState {
    int n = 0, m = 0, p = 0;
    int correction = 1;        // options['use_correction'] (coneprog.py:1781): 0 drops the Mehrotra term ds o dz of the second solve
    // equality constraints A x = b (single-problem entry point only; p = 0 in the batched mode), [B][p]
    double *b = nullptr, *y = nullptr, *ry = nullptr, *dy = nullptr, *Ax = nullptr, *y_out = nullptr, *resy0 = nullptr;
    double* ATy = nullptr;     // [B][n]
    // problem data and iterates, [B][n] / [B][m]
    double *q = nullptr, *h = nullptr, *x = nullptr, *s = nullptr, *z = nullptr;
    double *rx = nullptr, *rz = nullptr, *dx = nullptr, *dz = nullptr, *ds = nullptr;
    double *lmbda = nullptr, *d = nullptr, *di = nullptr, *ws3 = nullptr;
    double *Gx = nullptr, *GTz = nullptr, *Px = nullptr;
    double *x_out = nullptr, *s_out = nullptr, *z_out = nullptr;
    // per problem scalars [B]
    double *gap = nullptr, *resx0 = nullptr, *resz0 = nullptr, *step = nullptr, *sigma = nullptr;
    double *pcost = nullptr, *dcost = nullptr, *gap_out = nullptr;
    int *active = nullptr, *status = nullptr, *iters = nullptr, *freeze = nullptr;
    int* nactive = nullptr;    // one word: problems still iterating after the stopping test
};
void ipm_launch_start(const IpmState& S, int B, hipStream_t st);
void ipm_launch_residual(const IpmState& S, int B, int it, int maxiters, double abstol, double reltol, double feastol,
                         hipStream_t st);
void ipm_launch_info(const IpmState& S, const int* d_info, int it, int B, hipStream_t st);
void ipm_launch_rhs(const IpmState& S, int B, int i01, hipStream_t st);
void ipm_launch_post(const IpmState& S, int B, int i01, hipStream_t st);
void ipm_launch_update(const IpmState& S, int B, hipStream_t st);

// ---- device-resident conelp loop, 'l' + 'q' cones (conelp_ipm.hip) ---------------------------------------
enum LpScalar {
    LP_TAU = 0, LP_KAPPA, LP_DG, LP_DGI, LP_LG, LP_RT, LP_GAP, LP_SIGMA, LP_STEP, LP_MU, LP_DTAU, LP_DKAPPA, LP_WKAPPA3,
    LP_TT, LP_TK, LP_TS, LP_TZ, LP_RESX0, LP_RESY0, LP_RESZ0, LP_Z1Z1, LP_PCOST, LP_DCOST, LP_RELGAP, LP_PRES, LP_DRES,
    LP_PINFRES, LP_DINFRES, LP_GAP_OUT, LP_WTAU, LP_WKAPPA, LP_WTAU2, LP_WKAPPA2, LP_NSC
};
struct LpBuf { double *x, *y, *z, *s; int itau, ikappa; };   // one (x, y, z, tau, s, kappa) sextuple of f6 / res
struct LpState {
    int n = 0, m = 0, p = 0, ml = 0, nq = 0;          // m = ml + sum(q) + sum(s_k^2)
    const int *qoff = nullptr, *qdim = nullptr;       // per cone: offset into the cone vectors, dimension
    // 's' blocks (cone_ops_s.h): block k is sdim[k] x sdim[k], full symmetric storage, at soff[k] of the cone vectors, at
    // sloff[k] of lmbda (compact layout: sdim[k] entries), at soff[k] - lq of r / rti / sw1..3, at sloff[k] - lq of sigs / sigz
    int ns = 0, lq = 0, ldim = 0;                     // lq = ml + sum(q); ldim = lq + sum(s) = length of lmbda
    int nthreads = 256, lds_doubles = 0;              // workgroup size of the loop kernels; dynamic LDS (Jacobi staging)
    int smin = 0, smax = 0, swmax = 0;                // smallest / largest block order; largest order handled one block per wave
    double* jww = nullptr;                            // 16 per-wave scratch areas for the blocks handled one per wave
    const int *sdim = nullptr, *soff = nullptr, *sloff = nullptr;
    double *r = nullptr, *rti = nullptr, *sw1 = nullptr, *sw2 = nullptr, *sw3 = nullptr, *jw = nullptr, *sigs = nullptr,
           *sigz = nullptr;
    double *c = nullptr, *x = nullptr, *dx = nullptr, *rx = nullptr, *x1 = nullptr, *GTz = nullptr, *ATy = nullptr,
           *x_out = nullptr, *wx = nullptr, *wx2 = nullptr;                          // [n]
    double *b = nullptr, *y = nullptr, *dy = nullptr, *ry = nullptr, *y1 = nullptr, *Ax = nullptr, *y_out = nullptr,
           *wy = nullptr, *wy2 = nullptr;                                            // [p]
    double *h = nullptr, *s = nullptr, *z = nullptr, *ds = nullptr, *dz = nullptr, *rz = nullptr, *z1 = nullptr, *th = nullptr,
           *lmbda = nullptr, *lmbdasq = nullptr, *d = nullptr, *di = nullptr, *ws3 = nullptr, *Gx = nullptr, *s_out = nullptr,
           *z_out = nullptr, *t1 = nullptr, *t2 = nullptr, *wz3 = nullptr, *ws = nullptr, *wz = nullptr, *ws2 = nullptr,
           *wz2 = nullptr;                                                           // [m]
    double *v = nullptr, *beta = nullptr;             // W['v'] concatenated [sum(q)], W['beta'] [nq]
    double* sc = nullptr;                             // [LP_NSC]
    int *active = nullptr, *status = nullptr, *iters = nullptr, *init_optimal = nullptr, *nactive = nullptr;
};
void lp_launch_unit_scaling(const LpState& S, hipStream_t st);
void lp_launch_init_primal(const LpState& S, hipStream_t st, int given = 0);
void lp_launch_init_dual(const LpState& S, double abstol, double reltol, hipStream_t st, int have_primal = 0, int have_dual = 0);
void lp_launch_residual(const LpState& S, int it, int maxiters, double abstol, double reltol, double feastol, hipStream_t st);
void lp_launch_singular(const LpState& S, const int* d_info, int it, hipStream_t st);
void lp_launch_scale1(const LpState& S, hipStream_t st);
void lp_launch_build(const LpState& S, const LpBuf& D, const LpBuf& W, int i01, int save, hipStream_t st);
void lp_launch_copy(const LpState& S, const LpBuf& dst, const LpBuf& src, hipStream_t st);
void lp_launch_add(const LpState& S, const LpBuf& dst, const LpBuf& src, hipStream_t st);
void lp_launch_f6pre(const LpState& S, const LpBuf& X, hipStream_t st);
void lp_launch_f6post(const LpState& S, const LpBuf& X, hipStream_t st);
void lp_launch_res_a(const LpState& S, const LpBuf& U, hipStream_t st);
void lp_launch_res_b(const LpState& S, const LpBuf& U, const LpBuf& V, hipStream_t st);
void lp_launch_step(const LpState& S, const LpBuf& D, int i01, hipStream_t st);
void lp_launch_update(const LpState& S, const LpBuf& D, hipStream_t st);
void lp_launch_symm(const LpState& S, double* z, hipStream_t st);   // no-op without 's' blocks
int sdp_op_debug_launch(int op, int m, int arg, int wave_team, double* x, double* y, double* r, double* rti, double* lam, double* w,
                        double* out, hipStream_t st);

// ---- device-resident coneqp loop for one problem, 'l' + 'q' cones (coneqp_ipm.hip) ------------------------
enum QpScalar {
    QP_GAP = 0, QP_SIGMA, QP_STEP, QP_MU, QP_RESX0, QP_RESY0, QP_RESZ0, QP_PCOST, QP_DCOST, QP_RELGAP, QP_PRES, QP_DRES,
    QP_GAP_OUT, QP_NSC
};
struct QpBuf { double *x, *y, *z, *s; };            // one (x, y, z, s) quadruple of f4 / res
struct QpState {
    int n = 0, m = 0, p = 0, ml = 0, nq = 0;          // m = ml + sum(q) + sum(s_k^2)
    // batched mode (mi355kkt_batch_* with second-order cones): nbatch problems of one shape in lock step, one workgroup each
    // (blockIdx.x); every per-problem array below then holds nbatch slices back to back — [n] fields with stride n, [p] with
    // max(p, 1), [m] with m, v with max(sum(q), 1), beta with max(nq, 1), sc with QP_NSC, active / status / iters with 1 — and
    // the kernels shift their copy of the state to their problem first (qp_select, coneqp_ipm.hip).  'l' + 'q' cones only.
    int nbatch = 1;
    int correction = 1;                               // options['use_correction'] (coneprog.py:1781, :2377, :2426)
    const int *qoff = nullptr, *qdim = nullptr;
    // 's' blocks (cone_ops_s.h): block k is sdim[k] x sdim[k], full symmetric storage, at soff[k] of the cone vectors, at
    // sloff[k] of lmbda (compact layout: sdim[k] entries), at soff[k] - lq of r / rti / sw1..3, at sloff[k] - lq of sigs / sigz
    int ns = 0, lq = 0, ldim = 0;                     // lq = ml + sum(q); ldim = lq + sum(s) = length of lmbda
    int nthreads = 256, lds_doubles = 0;              // workgroup size of the loop kernels; dynamic LDS (Jacobi staging)
    int smin = 0, smax = 0, swmax = 0;                // smallest / largest block order; largest order handled one block per wave
    double* jww = nullptr;                            // 16 per-wave scratch areas for the blocks handled one per wave
    const int *sdim = nullptr, *soff = nullptr, *sloff = nullptr;
    double *r = nullptr, *rti = nullptr, *sw1 = nullptr, *sw2 = nullptr, *sw3 = nullptr, *jw = nullptr, *sigs = nullptr,
           *sigz = nullptr;
    double *q = nullptr, *x = nullptr, *dx = nullptr, *rx = nullptr, *Px = nullptr, *GTz = nullptr, *ATy = nullptr,
           *x_out = nullptr, *wx = nullptr, *wx2 = nullptr;                          // [n]
    double *b = nullptr, *y = nullptr, *dy = nullptr, *ry = nullptr, *Ax = nullptr, *y_out = nullptr, *wy = nullptr,
           *wy2 = nullptr;                                                           // [p]
    double *h = nullptr, *s = nullptr, *z = nullptr, *ds = nullptr, *dz = nullptr, *rz = nullptr, *lmbda = nullptr,
           *lmbdasq = nullptr, *d = nullptr, *di = nullptr, *ws3 = nullptr, *Gx = nullptr, *s_out = nullptr, *z_out = nullptr,
           *t1 = nullptr, *t2 = nullptr, *wz3 = nullptr, *ws = nullptr, *wz = nullptr, *ws2 = nullptr, *wz2 = nullptr;   // [m]
    double *v = nullptr, *beta = nullptr;
    double* sc = nullptr;                             // [QP_NSC]
    int *active = nullptr, *status = nullptr, *iters = nullptr, *nactive = nullptr;
};
void qp_launch_unit_scaling(const QpState& S, hipStream_t st);
void qp_launch_start(const QpState& S, hipStream_t st, int given = 0);   // given: s, z hold the caller's starting point
void qp_launch_residual(const QpState& S, int it, int maxiters, double abstol, double reltol, double feastol, hipStream_t st);
void qp_launch_singular(const QpState& S, const int* d_info, int it, hipStream_t st);
void qp_launch_build(const QpState& S, const QpBuf& D, const QpBuf& W, int i01, int save, hipStream_t st);
void qp_launch_copy(const QpState& S, const QpBuf& dst, const QpBuf& src, hipStream_t st);
void qp_launch_add(const QpState& S, const QpBuf& dst, const QpBuf& src, hipStream_t st);
void qp_launch_f4pre(const QpState& S, const QpBuf& X, hipStream_t st);
void qp_launch_f4post(const QpState& S, const QpBuf& X, hipStream_t st);
void qp_launch_res_a(const QpState& S, const QpBuf& U, hipStream_t st);
void qp_launch_res_b(const QpState& S, const QpBuf& U, const QpBuf& V, hipStream_t st);
void qp_launch_step(const QpState& S, const QpBuf& D, int i01, hipStream_t st);
void qp_launch_update(const QpState& S, const QpBuf& D, hipStream_t st);
void qp_launch_symm(const QpState& S, double* z, hipStream_t st);   // no-op without 's' blocks

}  // namespace mi355kkt

/* mi355kkt.h -- C ABI of the MI355X-native KKT-solve backend for CVXOPT's cone solvers.
 *
 * The library replaces, for ONE hot path, what the reference reaches through its Python plug-in hook
 *     f = kktsolver(W);  f(x, y, z)        (reference src/python/coneprog.py:323-344, :1658-1685)
 * i.e. the closures built by the factories
 *     misc.kkt_chol2   src/python/misc.py:1352-1567      misc.kkt_chol  src/python/misc.py:1213-1349
 *     misc.kkt_ldl     src/python/misc.py:1055-1125      misc.kkt_ldl2  src/python/misc.py:1128-1210
 * and everything underneath them on that path (misc_solvers.scale src/C/misc_solvers.c:85-244,
 * base.gemm/base.syrk src/C/base.c:476/:742, blas.syrk/trsv/trsm/gemv src/C/blas.c:3039/:1806/:3742/:872,
 * lapack.potrf/potrs src/C/lapack.c:1471/:1553).  Nothing like this ABI exists in the reference (it is
 * in-process Python + Fortran BLAS); INTEGRATION.md shows the binding a cvxopt maintainer would add.
 *
 * Conventions (same as the reference, src/C/cvxopt.h:46-69): FP64, column-major, 0-based.
 * The KKT system solved is
 *     [ H   A'  G'   ] [ ux ]   [ bx ]
 *     [ A   0   0    ] [ uy ] = [ by ]        W = Nesterov-Todd scaling,
 *     [ G   0  -W'W  ] [ uz ]   [ bz ]
 * and solve() overwrites (x, y, z) = (bx, by, bz) with (ux, uy, W uz) exactly as the hook requires.
 *
 * Return codes: 0 ok; >0 LAPACK-style "leading minor of order info is not positive definite"
 * (the Python layer raises ArithmeticError(info), reference src/C/lapack.c:32-34); <0 errors below.
 * All functions are plain C, take plain pointers and sizes, and are thread-compatible per handle.
 * No C++ exception crosses this boundary: a failing host-side allocation (std::bad_alloc) inside the library comes back as
 * MI355KKT_ENOMEM, any other C++ exception as MI355KKT_EHIP, with the text in mi355kkt_last_error().
 */
#ifndef MI355KKT_H
#define MI355KKT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355KKT_OK 0
#define MI355KKT_EINVAL (-1)  /* bad argument (ValueError / TypeError in the Python layer)  */
#define MI355KKT_EHIP (-2)    /* HIP runtime failure; see mi355kkt_last_error()              */
#define MI355KKT_ENOMEM (-3)
#define MI355KKT_ENOTIMPL (-4)

/* factorisation flavour requested by the host-side factory (all share the device engine; the
 * flavour fixes the regularisation semantics and which reference factory is mirrored) */
#define MI355KKT_CHOL2 0 /* misc.kkt_chol2: S = H + G'W^-1W^-T G, K = A S^-1 A'  (misc.py:1352)      */
#define MI355KKT_CHOL 1  /* misc.kkt_chol : same reduced system (QR elimination of A in the ref.)   */
#define MI355KKT_LDL 2   /* misc.kkt_ldl  : 3x3 quasi-definite LDL', static order z|x|y, + kktreg   */
#define MI355KKT_LDL2 3  /* misc.kkt_ldl2 : 2x2 reduced LDL'                                         */

typedef struct mi355kkt_solver mi355kkt_solver;

/* Nesterov-Todd scaling, flat arrays (reference W dict: coneprog.py:327-334).  Host or device
 * pointers depending on the entry point.  Unused parts may be NULL. */
typedef struct mi355kkt_scaling {
    const double* d;    /* ml            W['d']   */
    const double* di;   /* ml            W['di']  */
    const double* v;    /* sum(q)        W['v'][k] concatenated                         */
    const double* beta; /* nq            W['beta']                                      */
    const double* r;    /* sum(s_k^2)    W['r'][k] column-major, concatenated           */
    const double* rti;  /* sum(s_k^2)    W['rti'][k]                                    */
} mi355kkt_scaling;

/* ---- library / device ------------------------------------------------------------------------ */
int mi355kkt_version(void);
const char* mi355kkt_last_error(void);
int mi355kkt_device_count(void);
/* fills name (<= len bytes), number of compute units, global memory in bytes */
int mi355kkt_device_info(int device, char* name, int len, int* num_cus, size_t* mem_bytes);

/* ---- raw device memory helpers (so a host language without a HIP binding can keep data in HBM) -- */
int mi355kkt_dev_malloc(void** ptr, size_t bytes);
int mi355kkt_dev_free(void* ptr);
int mi355kkt_memcpy_h2d(void* dst, const void* src, size_t bytes);
int mi355kkt_memcpy_d2h(void* dst, const void* src, size_t bytes);
int mi355kkt_memcpy_d2d(void* dst, const void* src, size_t bytes);
int mi355kkt_device_synchronize(void);
/* Cross-process sharing of device memory on one node (the batch scatter / gather without a collective: SURVEY 8(e) "alternative
 * without xGMI collectives"; cvxopt_amd.batch.ShardedBatch(transport="ipc")).  export: a 64-byte handle of the ALLOCATION that
 * contains dptr, and dptr's byte offset in it (dptr may point into a pooled allocation).  open, in ANOTHER process: maps that
 * allocation (peer access over xGMI when it lives on another device) and returns its base address there; close unmaps it. */
int mi355kkt_ipc_export(const void* dptr, void* handle64, int64_t* offset, int64_t* alloc_bytes);
int mi355kkt_ipc_open(const void* handle64, void** base);
int mi355kkt_ipc_close(void* base);

/* ---- solver handle: mirrors `factor = misc.kkt_<name>(G, dims, A[, mnl][, kktreg])` ------------- */
/* dims = {'l': ml, 'q': q[0..nq), 's': s[0..ns)};  n variables, p equalities.  G is cdim x n with
 * cdim = ml + sum(q) + sum(s_k^2). */
int mi355kkt_create(mi355kkt_solver** out, int device, int kind, int n, int p, int ml, int nq, const int* q,
                    int ns, const int* s);
void mi355kkt_destroy(mi355kkt_solver* h);

/* constants captured at factory time (copied to HBM; caller keeps ownership of the host arrays) */
int mi355kkt_set_G_dense(mi355kkt_solver* h, const double* G, int64_t ldG);         /* cdim x n  */
int mi355kkt_set_G_csc(mi355kkt_solver* h, const int64_t* colptr, const int64_t* rowind,
                       const double* values);                                        /* CCS, cdim x n */
/* rows [row0, row0 + nrows) of the dense G := src (host, column-major): the Jacobian rows Df that cvxprog.cp / cpl stack on
 * top of G in every iteration (misc.py:1265-1266, :1412-1416); the handle is then created with dims['l'] = mnl + l. */
int mi355kkt_set_G_rows(mi355kkt_solver* h, int row0, int nrows, const double* src, int64_t ldsrc);
int mi355kkt_set_A_dense(mi355kkt_solver* h, const double* A, int64_t ldA);         /* p x n     */
/* Sparse mode (reference misc.py:1405-1462 sparse branch of kkt_chol2 -> cholmod.symbolic/numeric/solve,
 * src/C/cholmod.c:273-558): G (cdim x n) and H (n x n, lower triangle used; NULL = 0) in CCS.  Runs the symbolic
 * analysis (ordering, supernodes) once; factor()/solve() then use the supernodal multifrontal device engine.
 * LP cone, p = 0 only. */
int mi355kkt_set_sparse_problem(mi355kkt_solver* h, const int64_t* gcolptr, const int64_t* growind, const double* gvalues,
                                const int64_t* hcolptr, const int64_t* hrowind, const double* hvalues);
/* The same with `extra_rows` (0 or p) more rows of G below the cdim cone rows: the rows of A with unit scaling, i.e.
 * S = H + G'D^2 G + A'A -- the reference's fallback for a singular S on the first factorisation (misc.py:1433-1447); in
 * sparse mode the pattern of S grows, so this runs a new symbolic analysis.  The handle is in singular mode afterwards. */
int mi355kkt_set_sparse_problem_aug(mi355kkt_solver* h, const int64_t* gcolptr, const int64_t* growind, const double* gvalues,
                                    const int64_t* hcolptr, const int64_t* hrowind, const double* hvalues, int extra_rows);
/* A (p x n) in CSR (rowptr[p+1], colind[nnz], values[nnz]) for the sparse engine: kept sparse on the device (CSR + its
 * transpose); the Schur complement K = A S^-1 A' (misc.py:1464-1487, cholmod.spsolve src/C/cholmod.c:583-654) is formed from
 * sparse right-hand sides, 256 rows of A per pass of the supernodal forward solve.  No dense copy of A exists.
 * rowptr[0] must be 0 and rowptr non-decreasing, column indices in [0, n) (else MI355KKT_EINVAL); columns of a row may come in
 * any order, repeated (row, column) entries are summed -- the same matrix as the dense upload of the same triplets. */
int mi355kkt_set_A_csr(mi355kkt_solver* h, const int64_t* rowptr, const int64_t* colind, const double* values);
int mi355kkt_sparse_stats(const mi355kkt_solver* h, int64_t* nnzL, int* nsupernodes, int* nlevels, double* flops);
/* the fill-reducing ordering chosen by the symbolic analysis (csrc/ordering.cpp): 1 nested dissection, 2 approximate
 * minimum degree; MI355KKT_EINVAL when the handle is not in sparse mode */
int mi355kkt_sparse_ordering(const mi355kkt_solver* h);
/* device-resident variants: the solver borrows the pointers (no copy); they must outlive the handle */
int mi355kkt_set_G_device(mi355kkt_solver* h, const double* dG, int64_t ldG);
int mi355kkt_set_A_device(mi355kkt_solver* h, const double* dA, int64_t ldA);

/* H (= P for coneqp; NULL means H = 0, as in conelp).  Only the lower triangle is referenced
 * (reference coneprog.py:1475-1477).  Copied to HBM; call again whenever H changes. */
int mi355kkt_set_H_dense(mi355kkt_solver* h, const double* H, int64_t ldH);
int mi355kkt_set_H_device(mi355kkt_solver* h, const double* dH, int64_t ldH);
/* set_H_dense without the wait: the host buffer is pinned in place (hipHostRegister, cached while the same buffer is
 * passed again), the copy runs on the handle's copy stream and the next factor() overlaps it with the scaled SYRK
 * (S = Gs'Gs, then S += tril(H)).  This is what makes "re-upload H at every factor(W, H)" -- the only safe reading of the
 * hook when the caller may have changed H in place (cvxprog.py:526-537) -- cost next to nothing at n = 8192.
 * The caller keeps H alive and unmodified until that factor() returns and alive until the next set_H_* / destroy.
 * Only the page-aligned interior of a contiguous H (ldH == n) is pinned and copied asynchronously; the partial pages at its ends
 * are copied synchronously.  An H of less than 4 MB is copied synchronously (set_H_dense): nothing of the caller's heap is pinned for it -- pinning small
 * matrices where they lie (pages shared with the rest of the caller's heap) ended in GPU memory faults in long-lived processes
 * (the round-4 abort, reproduced and isolated in round 5). */
int mi355kkt_set_H_dense_async(mi355kkt_solver* h, const double* H, int64_t ldH);
/* diagonal regularisation of kkt_ldl (reference misc.py:1095-1098): K[x,x] += reg, K[y,y] -= reg,
 * K[z,z] = -1 - reg.  0 disables it. */
int mi355kkt_set_kktreg(mi355kkt_solver* h, double reg);
/* Per-handle options (no process-wide switches):
 *   "use_correction"  1 (default) / 0: options['use_correction'] of solvers.coneqp (coneprog.py:1781; 0 drops the Mehrotra
 *                     term ds o dz from the second right-hand side, :2377, :2426) for mi355kkt_coneqp* on this handle;
 *   "ldl_refinement"  steps of iterative refinement against the 3 x 3 system in solve() of the MI355KKT_LDL / _LDL2 flavours
 *                     (default 2, 0 = the plain reduced solve; not applied with kktreg);
 *   "qr_refinement"   steps of the same refinement for the flavours WITHOUT it (MI355KKT_CHOL / _CHOL2), applied only to the solves
 *                     of a factorisation whose reduced matrix is ill conditioned: (max L_ii / min L_ii)^2 >= 1e8, read back with
 *                     the info word; from 1e10 (conelp without H) the factor itself is repaired by CholeskyQR2, and where
 *                     chol(Gs'Gs) breaks down by a shifted Cholesky + two repair passes (shifted CholeskyQR3).  This is how
 *                     misc.kkt_qr (reference misc.py:1570-1699: two QR factorisations, error proportional to cond(W^-T G)) is
 *                     mapped onto the Cholesky engine (cond squared); 0 (default): off.
 * Unknown names: MI355KKT_EINVAL. */
int mi355kkt_set_option(mi355kkt_solver* h, const char* name, double value);
/* options['show_progress'] of the reference drivers (coneprog.py:2161-2208, :984-990) for the device-resident loops
 * mi355kkt_conelp / mi355kkt_coneqp: fn is called once per iteration, right after the stopping test, with
 * values = pcost, dcost, gap, pres, dres [, kappa/tau for conelp]; NULL switches it off (the default). */
typedef void (*mi355kkt_progress_fn)(int iteration, int nvalues, const double* values, void* user);
int mi355kkt_set_progress(mi355kkt_solver* h, mi355kkt_progress_fn fn, void* user);

/* factor(W, H): NT scaling + assembly + Cholesky/LDL' on the device.  `W` holds HOST pointers.
 * Returns 0, or info > 0 when a pivot is not positive (=> ArithmeticError(info)). */
int mi355kkt_factor(mi355kkt_solver* h, const mi355kkt_scaling* W);
/* same with DEVICE pointers in W (nothing crosses PCIe except the info word) */
int mi355kkt_factor_device(mi355kkt_solver* h, const mi355kkt_scaling* W);

/* solve(x, y, z): in place on HOST vectors of length n, p, cdim */
int mi355kkt_solve(mi355kkt_solver* h, double* x, double* y, double* z);
/* in place on DEVICE vectors; asynchronous on the solver's stream until mi355kkt_sync() */
int mi355kkt_solve_device(mi355kkt_solver* h, double* dx, double* dy, double* dz);
int mi355kkt_sync(mi355kkt_solver* h);

/* 1 if the first factorisation hit a singular S and switched to S + A'A (reference misc.py:1433-1447) */
int mi355kkt_is_singular_mode(const mi355kkt_solver* h);

/* Timings of the last factor()/solve() in milliseconds, measured with HIP events on the solver's
 * stream: out[0] scale+assemble (SYRK), out[1] Cholesky of S, out[2] Schur complement K (p > 0),
 * out[3] whole factor (device), out[4] whole last solve (device), out[5] the syrk_tn_kernel launch alone
 * (the dominant kernel; bench.py's roofline line).  Returns number written. */
int mi355kkt_get_timings(mi355kkt_solver* h, float* out, int n);
/* copies the factored S (lower Cholesky factor L in tril) to a host n x n buffer -- tests only */
int mi355kkt_get_factor(mi355kkt_solver* h, double* L, int64_t ldL);

/* ---- operator form of the matrices held by a handle ---------------------------------------------------------
 * out = op(M) x with host vectors: which = 0: G (cdim x n; 's' rows unpacked), 1: A (p x n), 2: H (n x n, symmetric from
 * tril(H)); trans != 0: the transpose.  These are the products behind the reference's Gf / Af / fP closures
 * (coneprog.py:531-550, :1843-1844, :1896-1916): handing conelp / coneqp callables built on them keeps G, A, P in HBM
 * and off the host's GEMV path for every cone type (cvxopt_amd.solvers). */
int mi355kkt_product(mi355kkt_solver* h, int which, int trans, const double* x, double* out);

/* ---- device-resident interior-point loop -----------------------------------------
 * The coneqp loop of coneprog.py:2044-2547 for dims = {'l': ml} (equality constraints A x = b allowed with the dense
 * engine: bv, y of length p; NULL when p = 0), run around this handle's factor/solve with the
 * iterates, the Nesterov-Todd scaling (misc.py:284-287, :444-464) and the step bookkeeping (coneprog.py:2376-2456)
 * resident in HBM: per iteration one word ("still active?") and the factorisation info cross PCIe.  G (set_G_*) and
 * optionally H = P (set_H_*), or the sparse problem (set_sparse_problem), must be set.  q: n, hv: ml (host).  Outputs (host): x (n), s, z (ml),
 * *status (1 optimal, 2 unknown: iteration limit, 3 unknown: singular KKT matrix), *iters, *pcost, *dcost, *gap.
 * Returns 0; <0 on error; 1 if the initial factorisation failed (the ValueError of coneprog.py:2065-2066). */
int mi355kkt_coneqp_lp(mi355kkt_solver* h, const double* q, const double* hv, const double* bv, int maxiters,
                       double abstol, double reltol, double feastol, double* x, double* y, double* s, double* z, int* status,
                       int* iters, double* pcost, double* dcost, double* gap);

/* The conelp loop (coneprog.py:586-1436: self-dual embedding, default starting point) for dims = {'l': ml, 'q': [...]}
 * around this handle (no H; equality constraints with the dense engine).  refinement < 0: the reference's default
 * (0 for the LP cone, 1 with second-order cones, :502-507).  c: n, hv: cdim, bv: p (host).
 * Outputs (host): x, y, s, z scaled like the reference's return values; *status: 1 optimal, 2 unknown (iteration limit),
 * 3 unknown (singular KKT matrix), 4 primal infeasible (y, z = certificate), 5 dual infeasible (x, s = certificate);
 * stats[10] = gap, relative gap, primal objective, dual objective, primal / dual infeasibility, residual as primal /
 * dual infeasibility certificate, ts, tz of the starting point (1e300 stands for the reference's None).
 * Returns 0; <0 on error; 1 if the initial factorisation failed (the ValueError of coneprog.py:690-691). */
int mi355kkt_conelp(mi355kkt_solver* h, const double* c, const double* hv, const double* bv, int maxiters, double abstol,
                    double reltol, double feastol, int refinement, double* x, double* y, double* s, double* z, int* status,
                    int* iters, double* stats);
/* The same with caller-supplied starting points (primalstart / dualstart of solvers.conelp, coneprog.py:696-739): have_primal -> x, s
 * hold primalstart on entry, have_dual -> y, z hold dualstart (a missing dualstart['y'] is zero; the caller has checked that the given
 * s / z lie in the interior of the cone: the reference's ValueError).  What is not given is constructed as in the default start;
 * as in the reference, only a constructed vector is shifted into the interior and the "starting point is optimal" exit belongs to
 * the fully constructed start. */
int mi355kkt_conelp_init(mi355kkt_solver* h, const double* c, const double* hv, const double* bv, int maxiters, double abstol,
                         double reltol, double feastol, int refinement, int have_primal, int have_dual, double* x, double* y,
                         double* s, double* z, int* status, int* iters, double* stats);

/* The coneqp loop (coneprog.py:2044-2547) for dims = {'l': ml, 'q': [...]} around this handle (H = P optional; equality
 * constraints with the dense engine); refinement < 0: the reference's default (0 for the LP cone, 1 with second-order
 * cones, :1862-1865).  q: n, hv: cdim, bv: p (host).  Outputs (host): x, y, s, z; *status: 1 optimal, 2 unknown
 * (iteration limit), 3 unknown (singular KKT matrix); stats[6] = gap, relative gap (1e300 = None), primal objective,
 * dual objective, primal infeasibility, dual infeasibility.  Returns 0; <0 on error; 1 if the KKT matrix of the starting
 * point is singular (the ValueError of coneprog.py:2065-2066). */
int mi355kkt_coneqp(mi355kkt_solver* h, const double* q, const double* hv, const double* bv, int maxiters, double abstol,
                    double reltol, double feastol, int refinement, double* x, double* y, double* s, double* z, int* status,
                    int* iters, double* stats);
/* The same with a caller-supplied starting point (initvals of solvers.coneqp, coneprog.py:2109-2149): have_init != 0 -> x, y, s, z
 * hold it on entry (the caller fills in the reference's defaults x = 0, y = 0, s = z = e for missing entries and has checked that
 * s and z are in the interior of the cone); the W = I factorisation / solve of the default start is skipped. */
int mi355kkt_coneqp_init(mi355kkt_solver* h, const double* q, const double* hv, const double* bv, int maxiters, double abstol,
                         double reltol, double feastol, int refinement, int have_init, double* x, double* y, double* s, double* z,
                         int* status, int* iters, double* stats);

/* ---- batched mode: nbatch independent dense LP-cone problems of one shape (BASELINE configs[4]) ---------
 * No reference API exists for this; per problem it is exactly factor()/solve() of the
 * kkt_chol2 hook with p = 0: S_b = H_b + G_b' diag(di_b)^2 G_b = L_b L_b', launched as batched kernels
 * (one blockIdx.z slice per problem).  Arrays are packed problem after problem, column-major inside. */
typedef struct mi355kkt_batch mi355kkt_batch;
int mi355kkt_batch_create(mi355kkt_batch** out, int device, int nbatch, int n, int ml);
/* "use_correction" as mi355kkt_set_option, for mi355kkt_batch_coneqp* on this batch */
int mi355kkt_batch_set_option(mi355kkt_batch* b, const char* name, double value);
void mi355kkt_batch_destroy(mi355kkt_batch* b);
int mi355kkt_batch_set_problem(mi355kkt_batch* b, const double* G, const double* H, int is_device);
int mi355kkt_batch_factor(mi355kkt_batch* b, const double* di, int is_device, int* info);
int mi355kkt_batch_solve(mi355kkt_batch* b, double* x, double* z, int is_device);
/* residual products of the IPM loop (coneprog.py:2170-2186): Gx = G x, GTz = G' z, Hx = H x per problem */
int mi355kkt_batch_products(mi355kkt_batch* b, const double* x, const double* z, double* Gx, double* GTz, double* Hx,
                            int is_device);
float mi355kkt_batch_last_factor_ms(const mi355kkt_batch* b);
/* The whole coneqp loop (coneprog.py:2044-2547, dims = {'l': ml}, no equality constraints) for every problem of the
 * batch with iterates, scaling and step bookkeeping resident in HBM; after set_problem().  q: [nbatch][n],
 * h: [nbatch][ml]; q, h and every output array may be HOST or DEVICE pointers (copied with hipMemcpyDefault; a sharded
 * batch keeps them in the rank's HBM between the RCCL scatter and gather).  Outputs: x [nbatch][n], s, z [nbatch][ml], status [nbatch] (1 optimal, 2 unknown:
 * iteration limit, 3 unknown: singular KKT matrix), iters, pcost, dcost, gap [nbatch]; *iterations_run = lock-step
 * iterations executed.  Returns 0; <0 on error; 1 if the initial factorisation failed (Rank([P; G]) < n, the
 * ValueError of coneprog.py:2065-2066). */
int mi355kkt_batch_coneqp(mi355kkt_batch* b, const double* q, const double* h, int maxiters, double abstol, double reltol,
                          double feastol, double* x, double* s, double* z, int* status, int* iters, double* pcost,
                          double* dcost, double* gap, int* iterations_run);
/* Batches with p equality constraints per problem, A_b x = b_b (reference misc.py:1464-1487, :1513-1563 per problem:
 * Asct_b = L_b^-1 A_b', K_b = Asct_b' Asct_b = L_K L_K'; a singular S_b in any problem at the first factorisation switches the
 * whole batch to S + A'A, misc.py:1433-1447).  A: [nbatch] blocks of p x n, column-major; bvec, y: [nbatch][p].  A failing pivot
 * of K_b is reported as n + pivot in that problem's info word. */
int mi355kkt_batch_create_eq(mi355kkt_batch** out, int device, int nbatch, int n, int ml, int p);
int mi355kkt_batch_set_A(mi355kkt_batch* b, const double* A, int is_device);
int mi355kkt_batch_solve_eq(mi355kkt_batch* b, double* x, double* y, double* z, int is_device);
int mi355kkt_batch_coneqp_eq(mi355kkt_batch* b, const double* q, const double* h, const double* bvec, int maxiters, double abstol,
                             double reltol, double feastol, double* x, double* y, double* s, double* z, int* status, int* iters,
                             double* pcost, double* dcost, double* gap, int* iterations_run);
/* Batches with second-order cones: dims = {'l': nl, 'q': q[0..nq)} for every problem, G_b is cdim x n with
 * cdim = nl + sum(q) (wherever the functions above say ml, read cdim), p equality constraints (0: none).  Per problem this is
 * the reference's kkt_chol with p = 0 / its Schur-complement twin (misc.py:1213-1349): Gs_b = W_b^-T G_b
 * (misc_solvers.c:144-183 per cone), S_b = H_b + Gs_b' Gs_b.  factor_cones: di [nbatch][cdim] (the first nl entries of every
 * slice), v [nbatch][sum(q)], beta [nbatch][nq] — W['di'], W['v'], W['beta'] of every problem; solve_eq / products / set_A as
 * above.  mi355kkt_batch_coneqp[_eq] on such a batch runs the reference's coneqp loop with the 'q' branches of
 * compute_scaling / update_scaling / max_step / sprod / sinv (misc.py:307-354, :503-573) and the default refinement step
 * (coneprog.py:1862-1865, :2330-2345), one workgroup per problem, h / s / z: [nbatch][cdim]. */
int mi355kkt_batch_create_cones(mi355kkt_batch** out, int device, int nbatch, int n, int nl, int nq, const int* q, int p);
int mi355kkt_batch_factor_cones(mi355kkt_batch* b, const double* di, const double* v, const double* beta, int is_device, int* info);

/* ---- stand-alone device operators (each is one stage of factor()/solve(); used by the per-kernel
 * parity tests and by the profiler).  All pointers are DEVICE pointers; calls are synchronous.
 * (Test hooks -- stateless host executions of device-side code, plans, orderings -- are declared in mi355kkt_test.h;
 * developer switches that can change results exist only in -DMI355KKT_DEBUG builds: mi355kkt_debug.h.) ---- */
/* S(lower) = H(lower) + G' diag(di)^2 G ;  di or H may be NULL */
int mi355kkt_op_syrk_scaled(const double* dG, int64_t ldG, int m, int n, const double* ddi, const double* dH,
                            int64_t ldH, double* dS, int64_t ldS, float* ms);
/* host-only: symbolic analysis of the sparse engine (nested-dissection ordering perm[new] = old, supernodal nnz(L)) */
int mi355kkt_op_symbolic(int n, int m, const int64_t* gcolptr, const int64_t* growind, const int64_t* hcolptr,
                         const int64_t* hrowind, int* perm, int64_t* nnzL, int* nsupernodes, int* nlevels);
/* X(:, 0:ncols) := W^-T X on the 'l' and 'q' rows, in place (misc_solvers.scale, trans='T', inverse='I') */
int mi355kkt_op_cone_scale(int ml, int nq, const int* q, double* dX, int64_t ldX, int ncols, const double* ddi,
                           const double* dv, const double* dbeta, float* ms);
/* issue-bound v_mfma_f64_16x16x4_f64 microbenchmark (measured FP64 matrix peak of this device) */
int mi355kkt_op_mfma_f64_peak(int iters, float* tflops);
/* in-place lower Cholesky; *info as LAPACK dpotrf */
int mi355kkt_op_potrf(double* dA, int64_t ldA, int n, int* info, float* ms);
/* X := L^-1 X (trans = 0) or L^-T X (trans = 1) */
int mi355kkt_op_trsm_lower(const double* dL, int64_t ldL, int n, double* dX, int64_t ldX, int nrhs, int trans,
                           float* ms);
/* zs := w .* z, y += (diag(w) G)' zs   and   z := w .* (G x) - zs   (the two products of solve()) */
int mi355kkt_op_gemv_t_scaled(const double* dG, int64_t ldG, int m, int n, const double* dw, const double* dz,
                              double* dzs, double* dy, float* ms);
int mi355kkt_op_gemv_n_scaled(const double* dG, int64_t ldG, int m, int n, const double* dw, const double* dx,
                              const double* dzs, double* dz, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* MI355KKT_H */

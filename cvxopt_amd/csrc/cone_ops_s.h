// Cone operations of the interior-point loops for the 's' (positive semidefinite) blocks: the 's' branches of
// misc.compute_scaling (src/python/misc.py:356-417), misc.update_scaling (:575-634), misc.scale (:118-164),
// misc_solvers.scale2 (src/C/misc_solvers.c:345-395), sprod (:699-766), sinv (:840-880), max_step (:1097-1150) restated
// for one block of order m held column-major with leading dimension m.
//
// Storage convention of the device loops: every cone vector keeps its 's' blocks as FULL symmetric matrices (both
// triangles valid).  The reference only guarantees the lower triangle and repairs the rest with misc.symm where a
// consumer needs it (coneprog.py:2405-2409); with both triangles kept, sdot (misc_solvers.c:1019-1033) is the plain dot
// product of the unpacked storage and sgemv (misc.py:801-832: trisc, gemv, triusc) is the plain gemv, so the residual
// kernels need no special case.  lmbda keeps the reference's compact layout: m entries per 's' block.
//
// Written once in SPMD form over a "team" P (tid, nt, sync, sum, max): the device loops instantiate it with one 256-thread
// workgroup (ParWG, cone_ops.h), the CPU parity tests with a team of one (ParHost) against the reference's own functions
// (tests/test_cone_ops_cpu.py).  Every function ends with a team barrier; scratch lives in global memory (L2 resident
// for the block orders of interest), so no block order is excluded by the LDS size.
//
// LAPACK's dsyevd / dsyevr / dgesvd are replaced by a one-sided (Hestenes) Jacobi iteration with a round-robin pair
// ordering: all m/2 column pairs of a round rotate concurrently.  Singular / eigen vectors are unique only up to signs
// (and order within clusters), so the scaling matrices r, rti differ from the reference's by a signed permutation D of
// their columns (r D, rti D, lmbda permuted alike); every quantity of the original space -- x, s, z, the step lengths,
// the objectives, the iteration count -- is invariant under D.  Singular values are sorted like LAPACK's (descending),
// eigenvalues ascending.
#pragma once
#include <math.h>

namespace mi355kkt {

struct ParHost {   // team of one (host tests)
    __host__ __device__ int tid() const { return 0; }
    __host__ __device__ int nt() const { return 1; }
    __host__ __device__ void sync() const {}
    __host__ __device__ double sum(double v) const { return v; }
    __host__ __device__ double max(double v) const { return v; }
    template <class = void>
    __host__ __device__ void jacobi(double* G, double* V, int m, double* jw) const;
};

// scratch doubles s_jacobi / s_sort need for blocks of order <= maxm with a team of nt threads
__host__ __device__ inline size_t s_jw_doubles(int maxm, int nt) {
    const size_t np = (size_t)(maxm + 1) / 2 + 1;
    const size_t nw = np > (size_t)nt ? np : (size_t)nt;
    return 3 * nw + 3 * np + 3 * (size_t)maxm + 16;
}

// ---- small dense helpers ---------------------------------------------------------------------------------------
// C := op(A) op(B), all m x m; C must not alias A or B.  4 x 4 register tiles per thread: a quarter of the loads per
// multiply-add of the element-per-thread form (the single workgroup is bound by load latency, not by arithmetic).
// sym: the result is known to be symmetric -- tiles strictly above the diagonal are skipped, the lower triangle mirrored.
template <class P>
__host__ __device__ inline void s_gemm_tiles(const P& par, double* C, const double* A, bool tA, const double* B, bool tB, int m,
                                             bool sym) {
    const int tm = (m + 3) / 4, ntile = tm * tm;
    for (int t = par.tid(); t < ntile; t += par.nt()) {
        const int i0 = (t % tm) * 4, j0 = (t / tm) * 4;
        if (sym && j0 > i0 + 3) continue;
        double acc[4][4];
        for (int ii = 0; ii < 4; ++ii)
            for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = 0.0;
        const bool full = (i0 + 4 <= m) && (j0 + 4 <= m);
        for (int k = 0; k < m; ++k) {
            double a[4], b[4];
            if (full) {
                for (int ii = 0; ii < 4; ++ii) a[ii] = tA ? A[k + (size_t)(i0 + ii) * m] : A[i0 + ii + (size_t)k * m];
                for (int jj = 0; jj < 4; ++jj) b[jj] = tB ? B[j0 + jj + (size_t)k * m] : B[k + (size_t)(j0 + jj) * m];
            } else {
                for (int ii = 0; ii < 4; ++ii)
                    a[ii] = (i0 + ii < m) ? (tA ? A[k + (size_t)(i0 + ii) * m] : A[i0 + ii + (size_t)k * m]) : 0.0;
                for (int jj = 0; jj < 4; ++jj)
                    b[jj] = (j0 + jj < m) ? (tB ? B[j0 + jj + (size_t)k * m] : B[k + (size_t)(j0 + jj) * m]) : 0.0;
            }
            for (int ii = 0; ii < 4; ++ii)
                for (int jj = 0; jj < 4; ++jj) acc[ii][jj] += a[ii] * b[jj];
        }
        for (int ii = 0; ii < 4; ++ii)
            for (int jj = 0; jj < 4; ++jj) {
                const int i = i0 + ii, j = j0 + jj;
                if (i >= m || j >= m) continue;
                if (!sym) C[i + (size_t)j * m] = acc[ii][jj];
                else if (i >= j) {
                    C[i + (size_t)j * m] = acc[ii][jj];
                    C[j + (size_t)i * m] = acc[ii][jj];
                }
            }
    }
    par.sync();
}
template <class P>
__host__ __device__ inline void s_gemm(const P& par, double* C, const double* A, bool tA, const double* B, bool tB, int m) {
    s_gemm_tiles(par, C, A, tA, B, tB, m, false);
}
// C := op(A) B for a result known to be symmetric
template <class P>
__host__ __device__ inline void s_gemm_sym(const P& par, double* C, const double* A, bool tA, const double* B, int m) {
    s_gemm_tiles(par, C, A, tA, B, false, m, true);
}
template <class P>
__host__ __device__ inline void s_copy(const P& par, double* dst, const double* src, int n) {
    for (int e = par.tid(); e < n; e += par.nt()) dst[e] = src[e];
    par.sync();
}
// misc.symm (misc_solvers.c:604-631): upper triangle := lower triangle
template <class P>
__host__ __device__ inline void s_symm(const P& par, double* X, int m) {
    const int mm = m * m;
    for (int e = par.tid(); e < mm; e += par.nt()) {
        const int i = e % m, j = e / m;
        if (i < j) X[e] = X[j + i * m];
    }
    par.sync();
}
template <class P>
__host__ __device__ inline void s_add_diag(const P& par, double* X, int m, double a) {
    for (int i = par.tid(); i < m; i += par.nt()) X[i * (m + 1)] += a;
    par.sync();
}
// X := diag(y) (the 's' part of "ds := lmbdasq", "s := lmbda": coneprog.py:1264-1273, :1404-1413)
template <class P>
__host__ __device__ inline void s_set_diag(const P& par, double* X, const double* y, int m) {
    const int mm = m * m;
    for (int e = par.tid(); e < mm; e += par.nt()) X[e] = (e % m == e / m) ? y[e % m] : 0.0;
    par.sync();
}

// Cholesky factor of a symmetric positive definite block, lower triangle in place, strict upper triangle zeroed
// (lapack.potrf followed by the "blas.scal(0.0, Ls, offset = i*m, n = i)" loop of misc.py:385-386).  Returns 0, or j + 1
// when the pivot of column j is not positive.
template <class P>
__host__ __device__ inline int s_potrf(const P& par, double* A, int m) {
    int fail = 0;
    for (int j = 0; j < m; ++j) {
        const double djj = A[j + j * m];
        par.sync();
        if (!(djj > 0.0)) {
            if (!fail) fail = j + 1;
            continue;
        }
        const double d = sqrt(djj);
        for (int i = j + par.tid(); i < m; i += par.nt()) A[i + j * m] = (i == j) ? d : A[i + j * m] / d;
        par.sync();
        const int t = m - j - 1;
        for (int e = par.tid(); e < t * t; e += par.nt()) {
            const int i = j + 1 + e % t, k = j + 1 + e / t;
            if (i >= k) A[i + k * m] -= A[i + j * m] * A[k + j * m];
        }
        par.sync();
    }
    const int mm = m * m;
    for (int e = par.tid(); e < mm; e += par.nt())
        if (e % m < e / m) A[e] = 0.0;
    par.sync();
    return fail;
}

// ---- one-sided Jacobi ---------------------------------------------------------------------------------------------
// pair i of round r of the round-robin tournament on M (even) players
__host__ __device__ inline void s_pair(int i, int r, int M, int& p, int& q) {
    if (i == 0) {
        p = M - 1;
        q = r;
    } else {                       // (r + i) mod (M - 1), (r - i) mod (M - 1): both operands < 2 (M - 1), no division
        p = r + i;
        if (p >= M - 1) p -= M - 1;
        q = r + M - 1 - i;
        if (q >= M - 1) q -= M - 1;
    }
}

// The rotations of the one-sided Jacobi iteration, generic team version (barrier-separated phases: partial dot products of
// every pair over row chunks, one rotation per pair, rotation of the columns).  jw: s_jw_doubles(m, nt) doubles of scratch.
template <class P>
__host__ __device__ inline void s_jacobi_rotations(const P& par, double* G, double* V, int m, double* jw) {
    const int M = m + (m & 1), np = M / 2;
    int nchunk = par.nt() / np;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > (m + 7) / 8) nchunk = (m + 7) / 8;      // at least ~8 rows per partial sum
    const int nw = np * nchunk;
    double* part = jw;                                   // [nw][3] partial (alpha, beta, gamma)
    double* cs = jw + 3 * (size_t)(nw > par.nt() ? nw : par.nt());   // [np][3]: c, s, rotated
    const double tol = fmax(1e-15, 4.5e-16 * sqrt((double)m));
    for (int sweep = 0; sweep < 60; ++sweep) {
        double myrot = 0.0;
        for (int r = 0; r < M - 1; ++r) {
            for (int w = par.tid(); w < nw; w += par.nt()) {
                const int i = w / nchunk, c = w % nchunk;
                int p, q;
                s_pair(i, r, M, p, q);
                double al = 0.0, be = 0.0, ga = 0.0;
                if (p < m && q < m) {
                    const int r0 = (int)((long long)m * c / nchunk), r1 = (int)((long long)m * (c + 1) / nchunk);
                    const double *gp = G + (size_t)p * m, *gq = G + (size_t)q * m;
                    for (int k = r0; k < r1; ++k) {
                        const double a = gp[k], b = gq[k];
                        al += a * a;
                        be += b * b;
                        ga += a * b;
                    }
                }
                part[3 * w] = al;
                part[3 * w + 1] = be;
                part[3 * w + 2] = ga;
            }
            par.sync();
            for (int i = par.tid(); i < np; i += par.nt()) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int c = 0; c < nchunk; ++c) {
                    al += part[3 * (i * nchunk + c)];
                    be += part[3 * (i * nchunk + c) + 1];
                    ga += part[3 * (i * nchunk + c) + 2];
                }
                double cc = 1.0, ss = 0.0;
                if (fabs(ga) > tol * sqrt(al) * sqrt(be) && al > 0.0 && be > 0.0) {
                    const double zeta = (be - al) / (2.0 * ga);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    cc = 1.0 / sqrt(1.0 + t * t);
                    ss = cc * t;
                    myrot = 1.0;
                }
                cs[3 * i] = cc;
                cs[3 * i + 1] = ss;
            }
            par.sync();
            const int nwork = np * m;
            for (int w = par.tid(); w < nwork; w += par.nt()) {
                const int i = w / m, k = w % m;
                const double cc = cs[3 * i], ss = cs[3 * i + 1];
                if (ss == 0.0) continue;
                int p, q;
                s_pair(i, r, M, p, q);
                const double a = G[k + (size_t)p * m], b = G[k + (size_t)q * m];
                G[k + (size_t)p * m] = cc * a - ss * b;
                G[k + (size_t)q * m] = ss * a + cc * b;
                if (V) {
                    const double va = V[k + (size_t)p * m], vb = V[k + (size_t)q * m];
                    V[k + (size_t)p * m] = cc * va - ss * vb;
                    V[k + (size_t)q * m] = ss * va + cc * vb;
                }
            }
            par.sync();
        }
        if (par.max(myrot) == 0.0) break;
    }
}

template <class>
__host__ __device__ inline void ParHost::jacobi(double* G, double* V, int m, double* jw) const {
    s_jacobi_rotations(*this, G, V, m, jw);
}

// On entry G = B (m x m).  On exit G = B V with mutually orthogonal columns (= U diag(sig)), V orthogonal (accumulated
// when V != nullptr), sig[j] = ||G(:, j)||.  The rotations are the team's own (P::jacobi): s_jacobi_rotations for the
// generic team, one column pair per wave for a workgroup (cone_ops.h, s_jacobi_waves).
template <class P>
__host__ __device__ inline void s_jacobi(const P& par, double* G, double* V, double* sig, int m, double* jw) {
    const int mm = m * m;
    if (V) {
        for (int e = par.tid(); e < mm; e += par.nt()) V[e] = (e % m == e / m) ? 1.0 : 0.0;
    }
    par.sync();
    if (m > 1) par.jacobi(G, V, m, jw);
    for (int j = par.tid(); j < m; j += par.nt()) {
        double a = 0.0;
        for (int k = 0; k < m; ++k) a += G[k + (size_t)j * m] * G[k + (size_t)j * m];
        sig[j] = sqrt(a);
    }
    par.sync();
}

// rank[j] (stored as doubles) = position of key[j] in the sorted order (descending or ascending, ties by index)
template <class P>
__host__ __device__ inline void s_rank(const P& par, const double* key, double* rank, int m, bool descending) {
    for (int j = par.tid(); j < m; j += par.nt()) {
        int rk = 0;
        const double kj = key[j];
        for (int i = 0; i < m; ++i) {
            const double ki = key[i];
            const bool before = descending ? (ki > kj) : (ki < kj);
            if (before || (ki == kj && i < j)) ++rk;
        }
        rank[j] = (double)rk;
    }
    par.sync();
}
// dst(:, rank[j]) := scale_j * src(:, j)
template <class P>
__host__ __device__ inline void s_permute_cols(const P& par, double* dst, const double* src, const double* rank,
                                               const double* colscale, int m) {
    const int mm = m * m;
    for (int e = par.tid(); e < mm; e += par.nt()) {
        const int i = e % m, j = e / m;
        dst[i + (size_t)((int)rank[j]) * m] = src[e] * (colscale ? colscale[j] : 1.0);
    }
    par.sync();
}

// ---- the cone operations, one block -----------------------------------------------------------------------------
// misc.scale, 's' block (misc.py:118-164): X := M' X M (mt) or M X M' (!mt), X symmetric; T: m*m scratch.
//   trans 'N', inverse 'N':  r' X r      -> M = r,   mt = true
//   trans 'T', inverse 'N':  r X r'      -> M = r,   mt = false
//   trans 'N', inverse 'I':  rti X rti'  -> M = rti, mt = false
//   trans 'T', inverse 'I':  rti' X rti  -> M = rti, mt = true
template <class P>
__host__ __device__ inline void s_scale_blk(const P& par, double* X, const double* M, int m, bool mt, double* T) {
    s_gemm(par, T, X, false, M, !mt, m);           // T = X M  or  X M'
    s_gemm_sym(par, X, M, mt, T, m);               // X = M' T or  M T
}
// misc_solvers.sprod, diag = 'N' (misc_solvers.c:716-741): X := (X Y + Y X) / 2 = (T + T') / 2 with T = X Y (X, Y symmetric)
template <class P>
__host__ __device__ inline void s_sprod_blk(const P& par, double* X, const double* Y, int m, double* T) {
    const int mm = m * m;
    s_gemm(par, T, X, false, Y, false, m);
    for (int e = par.tid(); e < mm; e += par.nt()) X[e] = 0.5 * (T[e] + T[(e / m) + (size_t)(e % m) * m]);
    par.sync();
}
// sprod, diag = 'D' (misc_solvers.c:743-762): X_ij *= (y_i + y_j) / 2;   sinv (:858-876): X_ij /= (y_i + y_j) / 2
template <class P>
__host__ __device__ inline void s_sprod_diag_blk(const P& par, double* X, const double* y, int m, bool inverse) {
    const int mm = m * m;
    for (int e = par.tid(); e < mm; e += par.nt()) {
        const double g = 0.5 * (y[e % m] + y[e / m]);
        X[e] = inverse ? X[e] / g : X[e] * g;
    }
    par.sync();
}
// misc_solvers.scale2, 's' block (misc_solvers.c:376-392): X_ij /= sqrt(l_i) sqrt(l_j)  (inverse: *=), both triangles
// ("the inverse operation will be applied to nonsymmetric matrices")
template <class P>
__host__ __device__ inline void s_scale2_blk(const P& par, const double* lam, double* X, int m, bool inverse) {
    const int mm = m * m;
    for (int e = par.tid(); e < mm; e += par.nt()) {
        const double c = sqrt(lam[e % m]) * sqrt(lam[e / m]);
        X[e] = inverse ? X[e] * c : X[e] / c;
    }
    par.sync();
}
// Frobenius norm of a block (the shift that makes X + c I positive semidefinite)
template <class P>
__host__ __device__ inline double s_fro(const P& par, const double* X, int m) {
    double a = 0.0;
    const int mm = m * m;
    for (int e = par.tid(); e < mm; e += par.nt()) a += X[e] * X[e];
    return sqrt(par.sum(a));
}
// smallest eigenvalue of the symmetric block X (max_step without sigma: dsyevr 'N', range 1..1, misc_solvers.c:1138-1145).
// X is left alone; T1: m*m scratch; sig: m doubles.
template <class P>
__host__ __device__ inline double s_min_eig_blk(const P& par, const double* X, int m, double* T1, double* sig, double* jw) {
    const double c = s_fro(par, X, m);
    const int mm = m * m;
    for (int e = par.tid(); e < mm; e += par.nt()) T1[e] = X[e] + ((e % m == e / m) ? c : 0.0);
    par.sync();
    s_jacobi(par, T1, (double*)nullptr, sig, m, jw);
    double lo = 1e300;
    for (int j = par.tid(); j < m; j += par.nt()) lo = fmin(lo, sig[j] - c);
    return -par.max(-lo);
}
// eigenvalue decomposition (max_step with sigma: dsyevd 'V', misc_solvers.c:1131-1136): X := eigenvectors (columns),
// sig := eigenvalues, ascending.  T1: m*m scratch.  With the shift c = 2 ||X||_F the matrix X + c I is positive definite with
// condition number <= 3, its one-sided Jacobi iteration ends with G = (X + c I) V = V diag(sigma), so the eigenvectors are the
// normalised columns of G and no V has to be accumulated; eigenvalues sigma - c (absolute accuracy eps ||X||_F, LAPACK's).
template <class P>
__host__ __device__ inline void s_eig_blk(const P& par, double* X, double* sig, int m, double* T1, double* /*T2*/, double* jw) {
    const double c = 2.0 * s_fro(par, X, m);
    const int mm = m * m;
    double* sv = jw + s_jw_doubles(m, par.nt()) - 3 * (size_t)m - 8;      // [m] singular values, [m] ranks, [m] 1 / sigma
    double* rank = sv + m;
    double* inv = rank + m;
    if (c == 0.0) {                                                       // X = 0: eigenvalues 0, eigenvectors I
        for (int e = par.tid(); e < mm; e += par.nt()) X[e] = (e % m == e / m) ? 1.0 : 0.0;
        for (int j = par.tid(); j < m; j += par.nt()) sig[j] = 0.0;
        par.sync();
        return;
    }
    for (int e = par.tid(); e < mm; e += par.nt()) T1[e] = X[e] + ((e % m == e / m) ? c : 0.0);
    par.sync();
    s_jacobi(par, T1, (double*)nullptr, sv, m, jw);
    s_rank(par, sv, rank, m, false);
    for (int j = par.tid(); j < m; j += par.nt()) inv[j] = 1.0 / sv[j];
    par.sync();
    s_permute_cols(par, X, T1, rank, inv, m);
    for (int j = par.tid(); j < m; j += par.nt()) sig[(int)rank[j]] = sv[j] - c;
    par.sync();
}

// misc.compute_scaling, 's' block (misc.py:374-417): sk, zk symmetric positive definite (left alone); on exit
//     r' sk^-1 r = diag(lam)^-1,  r' zk r = diag(lam),  rti = r^-T.
// With Ls Ls' = sk, Lz Lz' = zk and the SVD Lz' Ls = U diag(lam) V':  r = Lz^-T U diag(lam)^1/2 (= Ls V diag(lam)^-1/2,
// the form update_scaling uses, misc.py:584-585: a product instead of the triangular solve), rti = Lz U diag(lam)^-1/2.
// T1, T2, T3: m*m scratch each.  Returns 0 or the failing pivot + 1 of either Cholesky factorisation.
template <class P>
__host__ __device__ inline int s_compute_scaling_blk(const P& par, const double* sk, const double* zk, double* r, double* rti,
                                                     double* lam, int m, double* T1, double* T2, double* T3, double* jw) {
    const int mm = m * m;
    s_copy(par, T1, sk, mm);
    int fail = s_potrf(par, T1, m);                        // T1 = Ls
    s_copy(par, T2, zk, mm);
    const int f2 = s_potrf(par, T2, m);                    // T2 = Lz
    if (!fail) fail = f2;
    s_gemm(par, T3, T2, true, T1, false, m);               // T3 = Lz' Ls -> G = U diag(sigma)
    double* sv = jw + s_jw_doubles(m, par.nt()) - 3 * (size_t)m - 8;
    double* rank = sv + m;
    s_jacobi(par, T3, r /* V */, sv, m, jw);
    s_rank(par, sv, rank, m, true);
    // rti = Lz U diag(lam)^-1/2 = Lz G diag(sigma^-3/2);  r = Ls V diag(sigma^-1/2)   (columns in sorted order)
    s_gemm(par, rti, T2, false, T3, false, m);             // Lz G
    s_gemm(par, T3, T1, false, r, false, m);               // Ls V
    par.sync();
    for (int e = par.tid(); e < mm; e += par.nt()) {
        const int j = e / m;
        const double sg = sv[j];
        T1[e % m + (size_t)((int)rank[j]) * m] = T3[e] / sqrt(sg);
        T2[e % m + (size_t)((int)rank[j]) * m] = rti[e] / (sg * sqrt(sg));
    }
    par.sync();
    s_copy(par, r, T1, mm);
    s_copy(par, rti, T2, mm);
    for (int j = par.tid(); j < m; j += par.nt()) lam[(int)rank[j]] = sv[j];
    par.sync();
    return fail;
}

// misc.update_scaling, 's' block (misc.py:592-634).  Ls, Lz: the factors of the updated variables in the old scaling
// (coneprog.py:1364-1395; general matrices, destroyed).  r := r Ls V diag(lam+)^-1/2, rti := rti Lz U diag(lam+)^-1/2 with
// the SVD Lz' Ls = U diag(lam+) V'.  T1, T2: m*m scratch each.
template <class P>
__host__ __device__ inline void s_update_scaling_blk(const P& par, double* Ls, double* Lz, double* r, double* rti, double* lam,
                                                     int m, double* T1, double* T2, double* jw) {
    const int mm = m * m;
    s_gemm(par, T1, r, false, Ls, false, m);               // r := r Ls
    s_copy(par, r, T1, mm);
    s_gemm(par, T1, rti, false, Lz, false, m);             // rti := rti Lz
    s_copy(par, rti, T1, mm);
    s_gemm(par, T1, Lz, true, Ls, false, m);               // T1 = Lz' Ls -> G = U diag(sigma), V in T2
    double* sv = jw + s_jw_doubles(m, par.nt()) - 3 * (size_t)m - 8;
    double* rank = sv + m;
    s_jacobi(par, T1, T2, sv, m, jw);
    s_rank(par, sv, rank, m, true);
    s_gemm(par, Ls, r, false, T2, false, m);               // r V
    s_gemm(par, Lz, rti, false, T1, false, m);             // rti G  (= rti U diag(sigma))
    for (int e = par.tid(); e < mm; e += par.nt()) {
        const int j = e / m;
        const double sg = sv[j];
        r[e % m + (size_t)((int)rank[j]) * m] = Ls[e] / sqrt(sg);
        rti[e % m + (size_t)((int)rank[j]) * m] = Lz[e] / (sg * sqrt(sg));
    }
    par.sync();
    for (int j = par.tid(); j < m; j += par.nt()) lam[(int)rank[j]] = sv[j];
    par.sync();
}

}  // namespace mi355kkt

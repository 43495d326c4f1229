"""Seeded synthetic problem generators for the BASELINE.json configs.

Pure NumPy; returns column-major float64 ndarrays.  Callers wrap them in cvxopt matrices
(`matrix(np.asfortranarray(a))`) or hand them to the C-ABI directly.
"""
import numpy as np


def dense_qp(n, m, seed=0, p=0):
    """Dense LP-cone QP:  min 1/2 x'Px + q'x  s.t.  Gx <= h, Ax = b.

    B~N(0,1)/sqrt(n), P = B'B + 1e-2 I, q~N(0,1), G~N(0,1)^{m x n}, x0~N(0,1),
    h = G x0 + U(0.1,1)  (strictly feasible, bounded);  A~N(0,1)^{p x n}, b = A x0.
    """
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((n, n)) / np.sqrt(n)
    P = B.T @ B + 1e-2 * np.eye(n)
    q = rng.standard_normal(n)
    G = rng.standard_normal((m, n))
    x0 = rng.standard_normal(n)
    h = G @ x0 + rng.uniform(0.1, 1.0, m)
    out = dict(P=np.asfortranarray(P), q=q, G=np.asfortranarray(G), h=h, dims={'l': m, 'q': [], 's': []})
    if p:
        A = rng.standard_normal((p, n))
        out['A'] = np.asfortranarray(A)
        out['b'] = A @ x0
    return out


def socp(n, ncones, r, seed=0, ml=0):
    """Cone LP with `ncones` second-order cones of dimension r (+ ml linear inequalities).

    G~N(0,1); interior s0, z0 (v0 = ||v1|| + U(.5,1.5) per cone); h = G x0 + s0, c = -G' z0
    => primal and dual strictly feasible, bounded.
    """
    rng = np.random.default_rng(seed)
    cdim = ml + ncones * r
    G = rng.standard_normal((cdim, n))
    x0 = rng.standard_normal(n)

    def interior():
        u = np.empty(cdim)
        u[:ml] = rng.uniform(0.5, 1.5, ml)
        for k in range(ncones):
            o = ml + k * r
            u[o + 1:o + r] = rng.standard_normal(r - 1)
            u[o] = np.linalg.norm(u[o + 1:o + r]) + rng.uniform(0.5, 1.5)
        return u
    s0, z0 = interior(), interior()
    return dict(c=-(G.T @ z0), G=np.asfortranarray(G), h=G @ x0 + s0,
                dims={'l': ml, 'q': [r] * ncones, 's': []})


def random_scaling(dims, seed=0, spread=2.0):
    """A valid Nesterov-Todd scaling W (as plain ndarrays) for dims; d spans 10^+-spread."""
    rng = np.random.default_rng(seed)
    ml = dims['l']
    d = 10.0 ** rng.uniform(-spread, spread, ml)
    W = {'d': d, 'di': 1.0 / d, 'v': [], 'beta': [], 'r': [], 'rti': []}
    for mk in dims['q']:
        v1 = rng.standard_normal(mk - 1) * rng.uniform(0.1, 2.0)
        v = np.concatenate(([np.sqrt(1.0 + v1 @ v1)], v1))      # v'Jv = 1, v0 > 0
        W['v'].append(v)
        W['beta'].append(float(10.0 ** rng.uniform(-spread / 2, spread / 2)))
    for mk in dims['s']:
        R = rng.standard_normal((mk, mk)) + 2.0 * np.eye(mk)
        W['r'].append(np.asfortranarray(R))
        W['rti'].append(np.asfortranarray(np.linalg.inv(R).T))
    return W


def grid_laplacian(k, shift=1e-2):
    """7-point Laplacian of the k x k x k grid + shift I (scipy.sparse CSC): the structured stand-in for the sparse config."""
    import scipy.sparse as sp
    e = np.ones(k)
    T = sp.diags([-e[:-1], 2 * e, -e[:-1]], [-1, 0, 1])
    I = sp.eye(k)
    return (sp.kron(sp.kron(T, I), I) + sp.kron(sp.kron(I, T), I) + sp.kron(sp.kron(I, I), T) + shift * sp.eye(k ** 3)).tocsc()


def tet_mesh_laplacian(n, seed=0, shift=1e-2):
    """Graph Laplacian of the Delaunay tetrahedralisation of n random points in the unit cube + shift I (scipy.sparse
    CSC, about 15.5 off-diagonals per row): an unstructured finite-element stiffness pattern, the class the sparse
    config's SuiteSparse matrix belongs to."""
    import scipy.sparse as sp
    from scipy.spatial import Delaunay
    t = Delaunay(np.random.default_rng(seed).random((n, 3))).simplices
    r = np.concatenate([t[:, a] for a in range(4) for b in range(4) if a != b])
    c = np.concatenate([t[:, b] for a in range(4) for b in range(4) if a != b])
    A = sp.coo_matrix((np.ones(len(r)), (r, c)), shape=(n, n)).tocsc()
    A.data[:] = 1.0                      # duplicates were summed: back to a 0/1 adjacency matrix
    return (sp.diags(np.asarray(A.sum(1)).ravel() + shift) - A).tocsc()


def tet_mesh_elasticity(nnodes, seed=0, shift=1e-2):
    """3 degrees of freedom per node on the Delaunay tetrahedralisation of `nnodes` random points in the unit cube: the stiffness
    matrix of the pin-jointed network of its edges (every edge a spring of unit stiffness along its direction d plus a weak
    isotropic part: block (d d' + 0.1 I) on the two diagonal positions, its negative off the diagonal) + shift I.  Symmetric
    positive definite, order 3 nnodes, irregular 3 x 3 block pattern with ~47 entries per row -- the structure (vector unknowns on an
    unstructured 3-D mesh) of the structural-mechanics matrices of the SuiteSparse collection the sparse config names, which no
    Laplacian has (scipy.sparse CSC)."""
    import scipy.sparse as sp
    from scipy.spatial import Delaunay
    pts = np.random.default_rng(seed).random((nnodes, 3))
    t = Delaunay(pts).simplices
    e = np.concatenate([t[:, [a, b]] for a in range(4) for b in range(a + 1, 4)])
    e = np.unique(np.sort(e, axis=1), axis=0)                # every edge once, (i < j)
    d = pts[e[:, 1]] - pts[e[:, 0]]
    d /= np.linalg.norm(d, axis=1)[:, None]
    B = d[:, :, None] * d[:, None, :] + 0.1 * np.eye(3)[None, :, :]        # one 3 x 3 block per edge
    i3, j3 = 3 * e[:, 0], 3 * e[:, 1]
    a, b = np.meshgrid(np.arange(3), np.arange(3), indexing='ij')
    rows, cols, vals = [], [], []
    for (r0, c0, sgn) in ((i3, i3, 1.0), (j3, j3, 1.0), (i3, j3, -1.0), (j3, i3, -1.0)):    # (B is symmetric: no transposes needed)
        rows.append((r0[:, None, None] + a[None]).ravel())
        cols.append((c0[:, None, None] + b[None]).ravel())
        vals.append((sgn * B).ravel())
    n = 3 * nnodes
    K = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsc()
    K = ((K + K.T) * 0.5 + shift * sp.eye(n)).tocsc()
    K.sort_indices()
    return K


def read_matrix_market(path, shift=0.0):
    """A symmetric positive definite matrix from a Matrix-Market file (coordinate real / integer, general / symmetric) as
    scipy.sparse CSC, for `bench.py --workload sparse --mtx FILE` -- the door for the SuiteSparse matrix BASELINE configs[3] names
    on the day its file is reachable (there is no network in the build).  A `pattern` file carries no values: it gets the graph
    Laplacian of its (symmetrised) pattern, L = D - A off the diagonal, plus the identity (positive definite).  Anything else is
    symmetrised ((A + A')/2); `shift` adds shift * max|diag| * I for files that are only semidefinite.  A result whose diagonal
    is not strictly positive cannot be positive definite and is refused here, with the remedy, instead of deep in the
    factorisation."""
    import scipy.io
    import scipy.sparse as sp
    info = scipy.io.mminfo(path)                          # (rows, cols, entries, format, field, symmetry)
    field = str(info[4]).lower()
    A = sp.csc_matrix(scipy.io.mmread(path))
    if A.shape[0] != A.shape[1]:
        raise ValueError("%s: %d x %d is not square" % (path, A.shape[0], A.shape[1]))
    if field == "pattern":
        n = A.shape[0]
        B = ((abs(A) + abs(A).T) > 0).astype(np.float64).tolil()
        B.setdiag(0.0)
        B = B.tocsc()
        B.eliminate_zeros()
        deg = np.asarray(B.sum(axis=1)).ravel()
        A = (sp.diags(deg + 1.0) - B).tocsc()
    else:
        if A.dtype.kind not in "fiu":
            raise ValueError("%s: real matrices only (field %r)" % (path, field))
        A = A.astype(np.float64)
        A = ((A + A.T) * 0.5).tocsc()
    if shift:
        A = (A + shift * float(np.max(np.abs(A.diagonal()))) * sp.eye(A.shape[0])).tocsc()
    dmin = float(A.diagonal().min()) if A.shape[0] else 1.0
    if not dmin > 0.0:
        raise ValueError("%s: smallest diagonal entry %.3g is not positive, the matrix cannot be positive definite; for a "
                         "semidefinite / indefinite file pass --mtx-shift S (adds S * max|diag| * I)" % (path, dmin))
    A.sort_indices()
    return A

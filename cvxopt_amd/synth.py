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

"""`conelp` / `coneqp` with every heavy product on the MI355X, for all cone types, through the reference's own
operator + kktsolver plug-in API.

The reference drivers accept G, A (and P) as Python callables instead of matrices (`conelp`: coneprog.py:521-550,
`coneqp`: :1838-1917) as long as `kktsolver` is a callable too.  Here both are device backed:

    * by default the whole loop runs on the device, all three cone types, starting points and the reference's options
      included (`cvxopt_amd.conelp_device` / `coneqp_device`; `device_loop=False` forces the host-driver path below),
    * `kktsolver`     = the GPU KKT engine of `cvxopt_amd.kkt` (factor + solves in HBM),
    * `G`, `A`, `P`   = closures around `mi355kkt_product`: y := alpha op(M) x + beta y with M resident in HBM
                        (for 's' cones with the trisc/triusc convention of misc.sgemv, misc.py:801-832),

so the host loop only does the O(n + cdim) cone-vector bookkeeping of the unmodified reference, and the iterates are
the reference's.  (For LP-cone problems `cvxopt_amd.coneqp_lp` / `conelp_lp` move that bookkeeping to the device too.)

    import cvxopt_amd.solvers as gsolvers
    sol = gsolvers.conelp(c, G, h, dims)            # same signature and result dict as cvxopt.solvers.conelp
    sol = gsolvers.coneqp(P, q, G, h, dims, A, b)
    sol = gsolvers.socp(c, Gl, hl, Gq, hq)          # the reference's own argument packing
    sol = gsolvers.sdp(c, Gl, hl, Gs, hs)           # 's' cones: the device-resident conelp loop (all three cone types)
"""
import numpy as np

from . import kkt as _kkt


def _dims_of(h, dims):
    if dims is None:
        return {'l': h.size[0], 'q': [], 's': []}
    return {'l': int(dims['l']), 'q': [int(k) for k in dims['q']], 's': [int(k) for k in dims['s']]}


def _trisc(x, dims):
    """copy of x with the strictly lower triangles of the 's' blocks doubled and the upper triangles zeroed
    (misc_solvers.c:887-938, what misc.sgemv applies before a product with G')."""
    if not dims['s']:
        return x
    x = x.copy()
    ind = dims['l'] + sum(dims['q'])
    for m in dims['s']:
        X = x[ind:ind + m * m].reshape(m, m, order='F')
        x[ind:ind + m * m] = (2.0 * np.tril(X, -1) + np.diag(np.diag(X))).reshape(-1, order='F')
        ind += m * m
    return x


def _operators(eng, dims):
    def view(v):
        a = np.asarray(v)
        return a.reshape(-1, order='F')

    def combine(y, t, alpha, beta):
        yv = view(y)
        if beta == 0.0:
            yv[:] = alpha * t
        else:
            yv[:] = alpha * t + beta * yv

    def Gop(x, y, alpha=1.0, beta=0.0, trans='N'):
        xv = view(x)
        if trans == 'T':
            xv = _trisc(xv, dims)
        combine(y, eng.product(0, trans == 'T', xv), alpha, beta)

    def Aop(x, y, alpha=1.0, beta=0.0, trans='N'):
        combine(y, eng.product(1, trans == 'T', view(x)), alpha, beta)

    def Pop(x, y, alpha=1.0, beta=0.0):
        combine(y, eng.product(2, False, view(x)), alpha, beta)
    return Gop, Aop, Pop


def _as_cvxopt(sol):
    """device-loop result (NumPy vectors) -> the reference's types (cvxopt 'd' matrices)"""
    from cvxopt import matrix
    out = dict(sol)
    for k in ('x', 'y', 's', 'z'):
        if out.get(k) is not None:
            out[k] = matrix(np.asarray(out[k], dtype=float).reshape(-1, 1)) if len(out[k]) else matrix(0.0, (0, 1))
    return out


def _options(kwargs, dims, lp, kktsolver):
    """solver options read and VALIDATED as the reference does (kwargs['options'] over solvers.options; conelp:
    coneprog.py:425-455 + :502-509, coneqp: :1770-1801 + :1862-1869): same defaults, same ValueErrors, in the same order --
    the kktsolver name is checked between the tolerances and `refinement`, as there (:458-466 / :1805-1813).
    Returns (options for the device loop, kktreg, debug, resolved kktsolver name or None for a user callable)."""
    from cvxopt import solvers
    o = kwargs.get('options', solvers.options)
    num = (float, int)
    kktreg = o.get('kktreg', None)
    if kktreg is not None and (not isinstance(kktreg, num) or kktreg < 0.0):
        raise ValueError("options['kktreg'] must be a nonnegative scalar")                 # :429-433 / :1774-1778
    maxiters = o.get('maxiters', 100)
    if not isinstance(maxiters, int) or maxiters < 1:
        raise ValueError("options['maxiters'] must be a positive integer")                 # :435-437 / :1783-1785
    abstol = o.get('abstol', 1e-7)
    if not isinstance(abstol, num):
        raise ValueError("options['abstol'] must be a scalar")
    reltol = o.get('reltol', 1e-6)
    if not isinstance(reltol, num):
        raise ValueError("options['reltol'] must be a scalar")
    if reltol <= 0.0 and abstol <= 0.0:
        raise ValueError("at least one of options['reltol'] and options['abstol'] must be positive")   # :447-449 / :1795-1797
    feastol = o.get('feastol', 1e-7)
    if not isinstance(feastol, num) or feastol <= 0.0:
        raise ValueError("options['feastol'] must be a positive scalar")                   # :451-453 / :1799-1801
    ks_name = _resolve_kktsolver(kktsolver, dims, lp)
    refinement = o.get('refinement', None)
    if refinement is None:
        refinement = 1 if (dims['q'] or dims['s']) else 0                                  # :503-507 / :1862-1865
    elif not isinstance(refinement, int) or refinement < 0:
        raise ValueError("options['refinement'] must be a nonnegative integer")            # :508-509 / :1867-1869
    out = dict(maxiters=maxiters, abstol=abstol, reltol=reltol, feastol=feastol, refinement=refinement, kktreg=kktreg,
               show_progress=bool(o.get('show_progress', True)))
    if not lp:
        out['use_correction'] = bool(o.get('use_correction', True))                        # :1781 (coneqp only)
    return out, kktreg, bool(o.get('debug', False)), ks_name


def _resolve_kktsolver(kktsolver, dims, lp):
    """None -> the reference's default (coneprog.py:458-462 / :1805-1809); strings are validated like :463-466 / :1810-1813,
    an explicit 'chol2' with second-order or semidefinite cones raises like misc.kkt_chol2 (misc.py:1381-1384);
    a callable is the documented plug-in API and is handed to the reference driver untouched (returns None here)."""
    if kktsolver is None:
        if dims['q'] or dims['s']:
            return 'qr' if lp else 'chol'
        return 'chol2'
    if isinstance(kktsolver, str):
        valid = ('ldl', 'ldl2', 'qr', 'chol', 'chol2') if lp else ('ldl', 'ldl2', 'chol', 'chol2')
        if kktsolver not in valid:
            raise ValueError("'%s' is not a valid value for kktsolver" % kktsolver)
        if kktsolver == 'chol2' and (dims['q'] or dims['s']):
            raise ValueError("kktsolver option 'kkt_chol2' is implemented only for problems with no second-order or "
                             "semidefinite cone constraints")
        return kktsolver
    return None


def conelp(c, G, h, dims=None, A=None, b=None, primalstart=None, dualstart=None, kktsolver=None, device_loop='auto',
           **kwargs):
    """cvxopt.solvers.conelp on the MI355X, same signature and result dict.  The whole loop runs on the device
    (`mi355kkt_conelp_init`, all three cone types, primalstart / dualstart included, coneprog.py:696-739).  The reference driver
    runs on the host with G, A as device operators and the GPU kktsolver ('chol' | 'chol2' | 'ldl' | 'ldl2') only for
    `device_loop=False`, options['debug'], extra keyword arguments (xnewcopy ...) or a problem without cone rows."""
    # (the reference drivers are taken from their defining module: `cvxopt.solvers.conelp = cvxopt_amd.solvers.conelp` is a
    #  legitimate way to drop this backend in, and must not make the host-driver path call itself)
    from cvxopt import coneprog, spmatrix
    dims = _dims_of(h, dims)
    n = c.size[0]
    if kktsolver is not None and not isinstance(kktsolver, str):
        # a user kktsolver(W): the reference driver and the user's code, nothing of ours
        return coneprog.conelp(c, G, h, dims, A=A, b=b, primalstart=primalstart, dualstart=dualstart, kktsolver=kktsolver,
                              **kwargs)
    extra = set(kwargs) - {'options'}
    o, kktreg, debug, ks_name = _options(kwargs, dims, True, kktsolver)
    if device_loop and not extra and not debug and (dims['l'] + sum(dims['q']) + sum(dims['s'])) > 0:
        return _as_cvxopt(_kkt.conelp_device(c, G, h, dims, A, b, kktsolver=ks_name, primalstart=primalstart,
                                             dualstart=dualstart, **o))
    Am = A if A is not None else spmatrix([], [], [], (0, n))
    ks = _kkt.kktsolver_lp(G, dims, Am, kind=ks_name, kktreg=kktreg)
    eng = ks.engine
    try:
        eng._set_H(None)                       # decides dense / sparse mode and places G in HBM
        Gop, Aop, _ = _operators(eng, dims)
        kw = {}
        if A is not None:
            kw = {'A': Aop, 'b': b}
        return coneprog.conelp(c, Gop, h, dims, primalstart=primalstart, dualstart=dualstart, kktsolver=ks, **kw, **kwargs)
    finally:
        eng.close()


def coneqp(P, q, G=None, h=None, dims=None, A=None, b=None, initvals=None, kktsolver=None, device_loop='auto',
           **kwargs):
    """cvxopt.solvers.coneqp on the MI355X, same signature and result dict.  The whole loop runs on the device
    (`mi355kkt_coneqp_init`, all three cone types, initvals and options['use_correction'] included, coneprog.py:2109-2149, :1781);
    the reference driver runs on the host with P, G, A as device operators only for `device_loop=False`, options['debug'], extra
    keyword arguments or a problem without cone rows."""
    from cvxopt import coneprog, spmatrix, matrix
    n = q.size[0]
    if G is None:
        G, h = spmatrix([], [], [], (0, n)), matrix(0.0, (0, 1))
    dims = _dims_of(h, dims)
    if kktsolver is not None and not isinstance(kktsolver, str):
        # a user kktsolver(W): the reference driver and the user's code, nothing of ours
        return coneprog.coneqp(P, q, G, h, dims, A=A, b=b, initvals=initvals, kktsolver=kktsolver, **kwargs)
    extra = set(kwargs) - {'options'}
    o, kktreg, debug, ks_name = _options(kwargs, dims, False, kktsolver)
    if device_loop and not extra and not debug and (dims['l'] + sum(dims['q']) + sum(dims['s'])) > 0:
        return _as_cvxopt(_kkt.coneqp_device(P, q, G, h, dims, A, b, kktsolver=ks_name, initvals=initvals, **o))
    Am = A if A is not None else spmatrix([], [], [], (0, n))
    ks = _kkt.kktsolver_qp(G, dims, Am, P, kind=ks_name, kktreg=kktreg)
    eng = ks.engine
    # P is constant by the contract of coneqp (coneprog.py:1440-1477): the hook need not re-upload it at every factor(W, P)
    const_H = _kkt.options.get('assume_constant_H')
    _kkt.options['assume_constant_H'] = True
    try:
        eng._set_H(P)
        Gop, Aop, Pop = _operators(eng, dims)
        kw = {}
        if A is not None:
            kw = {'A': Aop, 'b': b}
        return coneprog.coneqp(Pop, q, Gop, h, dims, initvals=initvals, kktsolver=ks, **kw, **kwargs)
    finally:
        _kkt.options['assume_constant_H'] = const_H
        eng.close()


_EXTERNAL = {'lp': ('glpk', 'mosek'), 'socp': ('mosek',), 'sdp': ('dsdp',), 'qp': ('mosek',)}


def _no_external_solver(kwargs, which):
    """`solver=` of solvers.lp / qp / socp / sdp selects GLPK / MOSEK / DSDP in the reference (coneprog.py:2807, :2877, :3332,
    :3893, :4343): bridges to other codes, outside this backend.  Every OTHER value -- None, 'default' (what modeling.op.solve
    passes, modeling.py:2627), any unknown string -- means conelp / coneqp there, and so it does here."""
    solver = kwargs.pop('solver', None)
    if solver in _EXTERNAL[which]:
        raise ValueError("cvxopt_amd.solvers: solver=%r selects an external solver of the reference; only the default "
                         "(conelp / coneqp) runs on the device" % (solver,))
    return kwargs


def _stack(blocks, n):
    """[Gl; G_1; ...; G_N] as one matrix: sparse when every block is, dense otherwise (coneprog.py:3303-3318, :4085-4098)"""
    from cvxopt import matrix, sparse, spmatrix
    if not blocks:
        return spmatrix([], [], [], (0, n))
    if all(isinstance(B, spmatrix) for B in blocks):
        return sparse(blocks)
    return matrix([matrix(B) for B in blocks])


def _stack_start(start, ml, sizes, lkey, bkey):
    """primalstart / dualstart of socp / sdp ('sl' + 'sq'|'ss' blocks, 'zl' + 'zq'|'zs') as one stacked cone vector"""
    from cvxopt import matrix
    v = matrix(0.0, (ml + sum(sizes), 1))
    if ml:
        v[:ml] = start[lkey]
    ind = ml
    for k, sz in enumerate(sizes):
        v[ind:ind + sz] = start[bkey][k][:]
        ind += sz
    return v


def _split_cone_vectors(sol, ml, sizes, suffix, shape):
    """conelp's 's', 'z' -> the reference's per-block entries ('sl', 'sq'|'ss', 'zl', 'zq'|'zs'; coneprog.py:3351-3377,
    :4129-4153); `shape(k)` is the size of block k in the result."""
    from cvxopt import matrix
    for key in ('s', 'z'):
        v = sol.pop(key)
        if v is None:
            sol[key + 'l'], sol[key + suffix] = None, None
            continue
        sol[key + 'l'] = v[:ml]
        parts, ind = [], ml
        for k, sz in enumerate(sizes):
            parts.append(matrix(v[ind:ind + sz], shape(k)))
            ind += sz
        sol[key + suffix] = parts
    return sol


def socp(c, Gl=None, hl=None, Gq=None, hq=None, A=None, b=None, primalstart=None, dualstart=None, **kwargs):
    """cvxopt.solvers.socp's argument convention (coneprog.py:3013-3378: stacks Gl, Gq[k] and calls conelp)."""
    from cvxopt import matrix
    kwargs = _no_external_solver(kwargs, 'socp')
    n = c.size[0]
    Gq, hq = list(Gq or []), list(hq or [])
    if len(Gq) != len(hq):
        raise TypeError("'hq' must be a list of %d dense 'd' matrices" % len(Gq))
    ml = Gl.size[0] if Gl is not None else 0
    q = [Gk.size[0] for Gk in Gq]
    G = _stack(([Gl] if Gl is not None else []) + Gq, n)
    h = matrix([matrix(v) for v in ([hl] if Gl is not None else []) + hq]) if (ml or q) else matrix(0.0, (0, 1))
    ps = ds = None
    if primalstart:
        ps = {'x': primalstart['x'], 's': _stack_start(primalstart, ml, q, 'sl', 'sq')}
    if dualstart:
        ds = {'z': _stack_start(dualstart, ml, q, 'zl', 'zq')}
        if A is not None and A.size[0]:
            ds['y'] = dualstart['y']
    sol = conelp(c, G, h, {'l': ml, 'q': q, 's': []}, A, b, primalstart=ps, dualstart=ds, **kwargs)
    return _split_cone_vectors(sol, ml, q, 'q', lambda k: (q[k], 1))


def sdp(c, Gl=None, hl=None, Gs=None, hs=None, A=None, b=None, primalstart=None, dualstart=None, **kwargs):
    """cvxopt.solvers.sdp's argument convention (coneprog.py:3566-4153): Gs[k] is m_k^2 x n (column j = vec of the j-th
    coefficient matrix), hs[k] is m_k x m_k; stacked into one cone LP with dims['s'] = [m_k]; 'ss', 'zs' come back as
    m_k x m_k matrices.  The whole loop runs on the device ('s' blocks included, csrc/cone_ops_s.h), primalstart / dualstart
    included (stacked here and handed to `mi355kkt_conelp_init`)."""
    from cvxopt import matrix
    kwargs = _no_external_solver(kwargs, 'sdp')
    n = c.size[0]
    Gs, hs = list(Gs or []), list(hs or [])
    ms = [int(round(Gk.size[0] ** 0.5)) for Gk in Gs]
    for k, (m, Gk) in enumerate(zip(ms, Gs)):
        if m * m != Gk.size[0]:
            raise TypeError("the squareroot of the number of rows in 'Gs[%d]' is not an integer" % k)
    if len(hs) != len(ms):
        raise TypeError("'hs' must be a list of %d dense or sparse 'd' matrices" % len(ms))
    for k, (m, hk) in enumerate(zip(ms, hs)):
        if hk.size != (m, m):
            raise TypeError("hs[%d] has size (%d,%d).  Expected size is (%d,%d)." % (k, hk.size[0], hk.size[1], m, m))
    ml = Gl.size[0] if Gl is not None else 0
    sizes = [m * m for m in ms]
    G = _stack(([Gl] if Gl is not None else []) + Gs, n)
    h = matrix(([matrix(hl)] if Gl is not None else []) + [matrix(hk)[:] for hk in hs]) if (ml or ms) else matrix(0.0, (0, 1))
    ps = ds = None
    if primalstart:
        ps = {'x': primalstart['x'], 's': _stack_start(primalstart, ml, sizes, 'sl', 'ss')}
    if dualstart:
        ds = {'z': _stack_start(dualstart, ml, sizes, 'zl', 'zs')}
        if A is not None and A.size[0]:
            ds['y'] = dualstart['y']
    sol = conelp(c, G, h, {'l': ml, 'q': [], 's': ms}, A, b, primalstart=ps, dualstart=ds, **kwargs)
    return _split_cone_vectors(sol, ml, sizes, 's', lambda k: (ms[k], ms[k]))


def lp(c, G, h, A=None, b=None, **kwargs):
    """cvxopt.solvers.lp (coneprog.py:2562: conelp with dims = {'l': m})."""
    return conelp(c, G, h, {'l': h.size[0], 'q': [], 's': []}, A, b, **_no_external_solver(kwargs, 'lp'))


def qp(P, q, G=None, h=None, A=None, b=None, **kwargs):
    """cvxopt.solvers.qp (coneprog.py:4258: coneqp with dims = {'l': m})."""
    return coneqp(P, q, G, h, None, A, b, **_no_external_solver(kwargs, 'qp'))


class _gpu_factories(object):
    """while active, cvxopt.misc.kkt_{chol,chol2,ldl,ldl2,qr} are the GPU factories (cvxprog resolves the kktsolver names
    at call time, cvxprog.py:527-537, :1877-1887); restores the previous binding unless install() was already in force"""

    def __enter__(self):
        self._was_installed = bool(_kkt._saved)
        _kkt.install()

    def __exit__(self, *exc):
        if not self._was_installed:
            _kkt.uninstall()
        return False


def cpl(c, F, G=None, h=None, dims=None, A=None, b=None, kktsolver=None, **kwargs):
    """cvxopt.solvers.cpl (cvxprog.py:35: convex problem with linear objective): the reference driver on the host, every
    `factor(W, H, Df)` / solve on the device through the kktsolver names ('ldl', 'ldl2', 'chol', 'chol2'; default as the
    reference: 'chol' with q / s cones, 'chol2' otherwise).  The nonlinear-constraint rows (mnl, Df) are refreshed on the
    device at every call (`mi355kkt_set_G_rows`)."""
    from cvxopt import cvxprog
    with _gpu_factories():
        return cvxprog.cpl(c, F, G, h, dims, A, b, kktsolver=kktsolver, **kwargs)


def cp(F, G=None, h=None, dims=None, A=None, b=None, kktsolver=None, **kwargs):
    """cvxopt.solvers.cp (cvxprog.py:1359) with the GPU factories behind the kktsolver names."""
    from cvxopt import cvxprog
    with _gpu_factories():
        return cvxprog.cp(F, G, h, dims, A, b, kktsolver=kktsolver, **kwargs)


def gp(K, F, g, G=None, h=None, A=None, b=None, kktsolver=None, **kwargs):
    """cvxopt.solvers.gp (cvxprog.py:1967: geometric program, calls cp) with the GPU factories."""
    from cvxopt import cvxprog
    with _gpu_factories():
        return cvxprog.gp(K, F, g, G, h, A, b, kktsolver=kktsolver, **kwargs)

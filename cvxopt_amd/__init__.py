"""cvxopt_amd -- MI355X-native KKT-solve backend for CVXOPT's cone solvers.

Public surface (mirrors the reference's plug-in layer, src/python/misc.py:1055-1699):

    from cvxopt_amd import kkt_chol, kkt_chol2, kkt_ldl, kkt_ldl2      # factories
    sol = cvxopt.solvers.coneqp(P, q, G, h, kktsolver=cvxopt_amd.kktsolver_qp(G, dims, A, P))
    cvxopt_amd.install()      # or: rebind cvxopt.misc.kkt_* so kktsolver='chol'|'chol2'|'ldl'|'ldl2' route to the GPU

    sol = cvxopt_amd.coneqp_lp(P, q, G, h)    # LP-cone QP with the whole interior-point loop resident on the GPU
    sol = cvxopt_amd.conelp_lp(c, G, h)       # LP (conelp's self-dual loop) likewise

The compute lives in libmi355kkt.so (hand-written HIP for gfx950 behind the C ABI of
include/mi355kkt.h).  There is no CPU fallback: importing works everywhere, but creating a solver
without the library or without a GPU raises.
"""
from .kkt import (kkt_chol, kkt_chol2, kkt_ldl, kkt_ldl2, kkt_qr, install, uninstall,   # noqa: F401
                  kktsolver_qp, kktsolver_lp, coneqp_lp, conelp_lp, conelp_device, coneqp_device)
from . import synth   # noqa: F401

__version__ = "0.1.0"

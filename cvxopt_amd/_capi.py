"""ctypes binding of include/mi355kkt.h (the C ABI of libmi355kkt.so).

Fails loudly: no library -> ImportError at first use; no GPU -> RuntimeError from create().
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# $CVXOPT_AMD_LIB: an alternative build of the SAME library (tools/asan_host.sh: host AddressSanitizer / UBSan build)
LIB_PATH = os.environ.get("CVXOPT_AMD_LIB") or os.path.join(_HERE, "libmi355kkt.so")

EINVAL, EHIP, ENOMEM, ENOTIMPL = -1, -2, -3, -4
CHOL2, CHOL, LDL, LDL2 = 0, 1, 2, 3

_lib = None

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_i64_p = C.POINTER(C.c_int64)
c_float_p = C.POINTER(C.c_float)


class Scaling(C.Structure):
    _fields_ = [("d", c_double_p), ("di", c_double_p), ("v", c_double_p), ("beta", c_double_p),
                ("r", c_double_p), ("rti", c_double_p)]


# name -> (restype, argtypes); this table is also what tests check against include/mi355kkt.h + include/mi355kkt_test.h
# (mi355kkt_debug_*: only in -DMI355KKT_DEBUG builds, bound lazily by debug_signatures() for tools/dev scripts)
SIGNATURES = {
    "mi355kkt_version": (C.c_int, []),
    "mi355kkt_last_error": (C.c_char_p, []),
    "mi355kkt_device_count": (C.c_int, []),
    "mi355kkt_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, c_int_p, C.POINTER(C.c_size_t)]),
    "mi355kkt_dev_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "mi355kkt_dev_free": (C.c_int, [C.c_void_p]),
    "mi355kkt_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi355kkt_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi355kkt_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi355kkt_ipc_export": (C.c_int, [C.c_void_p, C.c_void_p, c_i64_p, c_i64_p]),
    "mi355kkt_ipc_open": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "mi355kkt_ipc_close": (C.c_int, [C.c_void_p]),
    "mi355kkt_device_synchronize": (C.c_int, []),
    "mi355kkt_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  c_int_p, C.c_int, c_int_p]),
    "mi355kkt_destroy": (None, [C.c_void_p]),
    "mi355kkt_set_G_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mi355kkt_set_G_csc": (C.c_int, [C.c_void_p, c_i64_p, c_i64_p, c_double_p]),
    "mi355kkt_set_sparse_problem": (C.c_int, [C.c_void_p, c_i64_p, c_i64_p, c_double_p, c_i64_p, c_i64_p, c_double_p]),
    "mi355kkt_set_sparse_problem_aug": (C.c_int, [C.c_void_p, c_i64_p, c_i64_p, c_double_p, c_i64_p, c_i64_p, c_double_p, C.c_int]),
    "mi355kkt_set_A_csr": (C.c_int, [C.c_void_p, c_i64_p, c_i64_p, c_double_p]),
    "mi355kkt_sparse_stats": (C.c_int, [C.c_void_p, c_i64_p, c_int_p, c_int_p, c_double_p]),
    "mi355kkt_sparse_ordering": (C.c_int, [C.c_void_p]),
    "mi355kkt_set_A_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mi355kkt_set_G_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mi355kkt_set_A_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mi355kkt_set_H_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mi355kkt_set_H_dense_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mi355kkt_set_progress": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355kkt_set_H_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mi355kkt_set_kktreg": (C.c_int, [C.c_void_p, C.c_double]),
    "mi355kkt_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "mi355kkt_batch_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "mi355kkt_factor": (C.c_int, [C.c_void_p, C.POINTER(Scaling)]),
    "mi355kkt_factor_device": (C.c_int, [C.c_void_p, C.POINTER(Scaling)]),
    "mi355kkt_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355kkt_solve_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355kkt_sync": (C.c_int, [C.c_void_p]),
    "mi355kkt_is_singular_mode": (C.c_int, [C.c_void_p]),
    "mi355kkt_get_timings": (C.c_int, [C.c_void_p, c_float_p, C.c_int]),
    "mi355kkt_get_factor": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "mi355kkt_set_G_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64]),
    "mi355kkt_product": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "mi355kkt_coneqp_lp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                     C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_int_p, c_int_p,
                                     c_double_p, c_double_p, c_double_p]),
    "mi355kkt_conelp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                  C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_int_p, c_int_p,
                                  c_double_p]),
    "mi355kkt_coneqp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                  C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_int_p, c_int_p,
                                  c_double_p]),
    "mi355kkt_conelp_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                       C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_int_p,
                                       c_int_p, c_double_p]),
    "mi355kkt_coneqp_init": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                       C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_int_p, c_int_p,
                                       c_double_p]),
    "mi355kkt_batch_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int]),
    "mi355kkt_batch_destroy": (None, [C.c_void_p]),
    "mi355kkt_batch_set_problem": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mi355kkt_batch_factor": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_int_p]),
    "mi355kkt_batch_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mi355kkt_batch_products": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mi355kkt_batch_last_factor_ms": (C.c_float, [C.c_void_p]),
    "mi355kkt_batch_coneqp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.c_void_p, C.c_void_p, C.c_void_p, c_int_p, c_int_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, c_int_p]),
    "mi355kkt_batch_create_eq": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mi355kkt_batch_create_cones": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, C.c_int]),
    "mi355kkt_batch_factor_cones": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, c_int_p]),
    "mi355kkt_batch_set_A": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "mi355kkt_batch_solve_eq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mi355kkt_batch_coneqp_eq": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                           C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, c_int_p, c_int_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, c_int_p]),
    "mi355kkt_op_syrk_scaled": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_int64, C.c_void_p, C.c_int64, c_float_p]),
    "mi355kkt_op_symbolic": (C.c_int, [C.c_int, C.c_int, c_i64_p, c_i64_p, c_i64_p, c_i64_p, c_int_p, c_i64_p, c_int_p, c_int_p]),
    "mi355kkt_test_symbolic_plan": (C.c_int64, [C.c_int, C.c_int, c_i64_p, c_i64_p, c_i64_p, c_i64_p, c_i64_p, C.c_int64]),
    "mi355kkt_op_cone_scale": (C.c_int, [C.c_int, C.c_int, c_int_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, c_float_p]),
    "mi355kkt_test_cone_op_host": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355kkt_test_sdp_op_host": (C.c_int, [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5),
    "mi355kkt_test_sdp_op_device": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5),
    "mi355kkt_test_sdp_op_host_team": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 5),
    "mi355kkt_test_syrk_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, C.c_int, c_int_p, c_int_p, c_int_p]),
    "mi355kkt_test_ordering": (C.c_int, [C.c_int, c_i64_p, c_i64_p, C.c_int, c_int_p, c_double_p]),
    "mi355kkt_test_throw": (C.c_int, [C.c_int]),
    "mi355kkt_test_touches_brk_heap": (C.c_int, [C.c_void_p, C.c_size_t]),
    "mi355kkt_test_set_knob": (C.c_int, [C.c_char_p, C.c_char_p]),
    "mi355kkt_test_install_abort_dump": (C.c_int, [C.c_char_p]),
    "mi355kkt_test_guard_probe": (C.c_int, [C.c_int, C.c_int, c_double_p]),
    "mi355kkt_test_guard_violations": (C.c_int, []),
    "mi355kkt_op_mfma_f64_peak": (C.c_int, [C.c_int, c_float_p]),
    "mi355kkt_op_potrf": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, c_int_p, c_float_p]),
    "mi355kkt_op_trsm_lower": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                         c_float_p]),
    "mi355kkt_op_gemv_t_scaled": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, c_float_p]),
    "mi355kkt_op_gemv_n_scaled": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, c_float_p]),
}


# include/mi355kkt_debug.h -- present only in a library built with `build.sh --debug` (load it with $CVXOPT_AMD_LIB)
DEBUG_SIGNATURES = {
    "mi355kkt_debug_hwid": (C.c_int, [C.c_void_p, C.c_int]),
    "mi355kkt_debug_potf2_ts": (C.c_int, [C.c_void_p]),
    "mi355kkt_debug_tile_ts": (C.c_int, [C.c_void_p]),
    "mi355kkt_debug_wide_ts": (C.c_int, [C.c_void_p]),
    "mi355kkt_debug_syrk_skip": (C.c_int, [C.c_int]),
}


def set_knob(name, value):
    """include/mi355kkt_test.h: developer / test knob of the library (value None unsets it; name None unsets all)"""
    enc = lambda v: None if v is None else str(v).encode("ascii")
    check(lib().mi355kkt_test_set_knob(enc(name), enc(value)), "mi355kkt_test_set_knob")


def lib():
    """Loads libmi355kkt.so once.  If torch is (or can be) imported it is imported FIRST so that the
    process holds exactly one HIP runtime (torch bundles its own libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("cvxopt_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; "
                          "g.build()'` or `bash cvxopt_amd/csrc/build.sh` (there is NO CPU fallback)" % LIB_PATH)
    if "torch" not in sys.modules and os.environ.get("CVXOPT_AMD_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    for name, (res, args) in DEBUG_SIGNATURES.items():      # a -DMI355KKT_DEBUG build only
        f = getattr(L, name, None)
        if f is not None:
            f.restype = res
            f.argtypes = args
    _lib = L
    return L


def last_error():
    return lib().mi355kkt_last_error().decode("utf-8", "replace")


def check(rc, what):
    """Maps C return codes onto the reference's exception classes (SURVEY.md 8(b) 'Errors')."""
    if rc == 0:
        return
    if rc > 0:
        raise ArithmeticError(int(rc))            # singular / not positive definite (lapack.c:32-34)
    msg = "%s: %s" % (what, last_error())
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    if rc == ENOTIMPL:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


def device_count():
    return lib().mi355kkt_device_count()


PROGRESS_FN = C.CFUNCTYPE(None, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_void_p)


class DeviceBuffer(object):
    """A block of HBM owned through the C ABI (used by tests / bench to keep inputs resident)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().mi355kkt_dev_malloc(C.byref(p), self.nbytes), "dev_malloc")
        self.ptr = p.value

    @classmethod
    def from_array(cls, a):
        import numpy as np
        a = np.ascontiguousarray(a) if not (a.flags.f_contiguous or a.flags.c_contiguous) else a
        b = cls(a.nbytes)
        check(lib().mi355kkt_memcpy_h2d(b.ptr, a.ctypes.data, a.nbytes), "memcpy_h2d")
        return b

    def to_array(self, shape, dtype="float64", order="F"):
        import numpy as np
        out = np.empty(shape, dtype=dtype, order=order)
        assert out.nbytes <= self.nbytes
        check(lib().mi355kkt_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes), "memcpy_d2h")
        return out

    def free(self):
        if self.ptr:
            lib().mi355kkt_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

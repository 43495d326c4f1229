"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol declared in
include/mi355kkt.h (the boundary) and include/mi355kkt_test.h (the test hooks), the ctypes table covers exactly that set, the
production library holds NO mi355kkt_debug_* switch, and the host mirror fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols(names=("mi355kkt.h", "mi355kkt_test.h")):
    out = set()
    for name in names:
        src = open(os.path.join(ROOT, "include", name)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        out |= set(re.findall(r"\b(mi355kkt_[a-zA-Z0-9_]+)\s*\(", src))
    return sorted(out)


def test_header_declares_the_expected_surface():
    syms = header_symbols()
    for must in ("mi355kkt_create", "mi355kkt_destroy", "mi355kkt_set_G_dense", "mi355kkt_set_A_dense",
                 "mi355kkt_set_H_dense", "mi355kkt_factor", "mi355kkt_solve", "mi355kkt_factor_device",
                 "mi355kkt_solve_device", "mi355kkt_set_kktreg"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from cvxopt_amd import _capi
    L = _capi.lib()
    for s in header_symbols():
        assert hasattr(L, s), "libmi355kkt.so does not export %s" % s
    assert L.mi355kkt_version() >= 100


def test_ctypes_table_matches_header():
    from cvxopt_amd import _capi
    assert sorted(_capi.SIGNATURES.keys()) == header_symbols()


def test_boundary_header_is_free_of_test_and_debug_entry_points():
    """nothing in the production header (or binary) can change results behind the caller's back"""
    from cvxopt_amd import _capi
    boundary = header_symbols(("mi355kkt.h",))
    assert not [s for s in boundary if "_debug_" in s or "_test_" in s]
    assert all(s.startswith("mi355kkt_test_") for s in header_symbols(("mi355kkt_test.h",)))
    if os.path.basename(_capi.LIB_PATH) == "libmi355kkt.so":           # (a --debug build is loaded only through $CVXOPT_AMD_LIB)
        L = _capi.lib()
        for s in header_symbols(("mi355kkt_debug.h",)):
            assert s.startswith("mi355kkt_debug_") and not hasattr(L, s), "%s is exported by the production library" % s


def test_the_library_does_not_read_knobs_from_the_environment(monkeypatch):
    """kernel selection / ordering knobs are set through mi355kkt_test_set_knob only (csrc/knobs.h)"""
    import subprocess
    import sys
    so = os.path.join(ROOT, "cvxopt_amd", "libmi355kkt.so")
    strings = subprocess.run(["strings", "-a", so], capture_output=True, text=True).stdout if os.path.exists(so) else ""
    if strings:
        assert "MI355KKT_ROCTX" in strings                  # the one tracing switch
        src = "".join(open(os.path.join(ROOT, "cvxopt_amd", "csrc", f)).read()
                      for f in os.listdir(os.path.join(ROOT, "cvxopt_amd", "csrc")) if f.endswith((".hip", ".cpp", ".h")))
        calls = re.findall(r"(?<![_a-zA-Z])getenv\(\s*\"?([A-Za-z0-9_]*)", src)
        assert sorted(set(calls)) == ["MI355KKT_ROCTX", "name"], calls      # knobs.cpp: getenv(name) under MI355KKT_DEBUG only
    # functional: an ordering knob in the environment changes nothing, the same knob through the API does
    from scipy.sparse import diags
    from cvxopt_amd import _capi
    n = 400
    A = (diags([1.0, 1.0], [1, 20], shape=(n, n)) + diags([1.0, 1.0], [1, 20], shape=(n, n)).T).tocsc()
    A.sort_indices()

    def method():
        perm = np.zeros(n, dtype=np.int32)
        stats = np.zeros(8)
        rc = _capi.lib().mi355kkt_test_ordering(n, A.indptr.astype(np.int64).ctypes.data_as(_capi.c_i64_p),
                                                A.indices.astype(np.int64).ctypes.data_as(_capi.c_i64_p), 0,
                                                perm.ctypes.data_as(_capi.c_int_p), stats.ctypes.data_as(_capi.c_double_p))
        assert rc == 0
        return int(stats[0])
    base = method()
    other = "amd" if base == 1 else "nd"
    monkeypatch.setenv("MI355KKT_ORDERING", other)
    assert method() == base
    try:
        _capi.set_knob("MI355KKT_ORDERING", other)
        assert method() == (2 if other == "amd" else 1)
    finally:
        _capi.set_knob(None, None)
    assert method() == base


def test_no_torch_types_in_the_abi():
    src = open(os.path.join(ROOT, "include", "mi355kkt.h")).read()
    assert "torch" not in src.lower() and "at::" not in src and "#include <hip" not in src


def test_create_without_gpu_fails_loudly():
    from cvxopt_amd import _capi, kkt
    if _capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError):
        kkt.kkt_chol2(np.zeros((4, 3), order='F'), {'l': 4, 'q': [], 's': []}, np.zeros((0, 3)))
    h = ctypes.c_void_p()
    rc = _capi.lib().mi355kkt_create(ctypes.byref(h), 0, 0, 3, 0, 4, 0, None, 0, None)
    assert rc == _capi.EHIP and not h.value
    assert b"not available" in _capi.lib().mi355kkt_last_error()


def test_argument_errors_mirror_the_reference():
    from cvxopt_amd import kkt
    with pytest.raises(ValueError):                      # misc.py:1381-1384
        kkt.kkt_chol2(np.zeros((5, 3), order='F'), {'l': 2, 'q': [3], 's': []}, np.zeros((0, 3)))
    with pytest.raises(RuntimeError):                    # mnl > 0 (cvxprog.cp / cpl) is accepted; no GPU here -> fails loudly
        kkt.kkt_chol(np.zeros((5, 3), order='F'), {'l': 5, 'q': [], 's': []}, np.zeros((0, 3)), mnl=2)
    with pytest.raises(ValueError):
        kkt._vec(np.zeros(4), 5, "x")
    with pytest.raises(TypeError):
        kkt._vec(np.zeros(4, dtype=np.float32), 4, "x")
    with pytest.raises(ArithmeticError):
        from cvxopt_amd import _capi
        _capi.check(7, "factor")


def test_batch_entry_points_validate_their_arguments_before_touching_the_device():
    """bad shapes are refused with MI355KKT_EINVAL (no GPU needed); well-formed calls fail loudly with MI355KKT_EHIP here"""
    import ctypes as C
    from cvxopt_amd import _capi
    L = _capi.lib()
    h = C.c_void_p()
    EINVAL = -1
    q = (C.c_int * 3)(4, 0, 5)                            # a cone of dimension 0
    assert L.mi355kkt_batch_create_cones(C.byref(h), 0, 4, 8, 2, 3, q, 0) == EINVAL and not h.value
    assert b"positive" in L.mi355kkt_last_error()
    q = (C.c_int * 2)(4, 5)
    assert L.mi355kkt_batch_create_cones(C.byref(h), 0, 4, 8, -1, 2, q, 0) == EINVAL             # negative 'l'
    assert L.mi355kkt_batch_create_cones(C.byref(h), 0, 4, 8, 2, 2, None, 0) == EINVAL           # nq > 0 without q
    assert L.mi355kkt_batch_create_cones(C.byref(h), 0, 0, 8, 2, 2, q, 0) == EINVAL              # empty batch
    assert L.mi355kkt_batch_create_cones(C.byref(h), 0, 4, 8, 2, 2, q, 9) == EINVAL              # p > n
    assert L.mi355kkt_batch_create_eq(C.byref(h), 0, 4, 0, 5, 0) == EINVAL                       # n = 0
    assert L.mi355kkt_batch_factor_cones(None, None, None, None, 0, None) == EINVAL
    assert L.mi355kkt_batch_solve_eq(None, None, None, None, 0) == EINVAL
    if _capi.device_count() <= 0:
        assert L.mi355kkt_batch_create_cones(C.byref(h), 0, 4, 8, 2, 2, q, 0) == _capi.EHIP and not h.value
        from cvxopt_amd.batch import BatchKkt
        with pytest.raises(RuntimeError):                  # there is no CPU fallback
            BatchKkt(np.zeros((2, 8, 11)), dims={'l': 2, 'q': [4, 5], 's': []})


def test_install_and_uninstall_rebind_factories():
    import types
    import cvxopt_amd
    fake = types.SimpleNamespace(kkt_chol=1, kkt_chol2=2, kkt_ldl=3, kkt_ldl2=4, kkt_qr=5, other=6)
    cvxopt_amd.install(fake)
    assert fake.kkt_chol2 is cvxopt_amd.kkt_chol2 and fake.kkt_ldl is cvxopt_amd.kkt_ldl and fake.other == 6
    assert fake.kkt_qr is cvxopt_amd.kkt_qr
    cvxopt_amd.uninstall()
    assert (fake.kkt_chol, fake.kkt_chol2, fake.kkt_ldl, fake.kkt_ldl2) == (1, 2, 3, 4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "cvxopt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), f
                # cvxopt (the reference, installed by the user) is only ever the CALLER: kkt.install() rebinds its factories,
                # solvers.py hands its own drivers device operators; nothing else may import it
                assert "import cvxopt\n" not in txt and "from cvxopt " not in txt or f in ("kkt.py", "solvers.py"), f


def test_no_cpp_exception_crosses_the_c_abi():
    """every non-trivial entry point is a function-try-block: a throwing host path returns an error code (and a message)
    instead of std::terminate-ing the interpreter (header: 'No C++ exception crosses this boundary')"""
    from cvxopt_amd import _capi
    L = _capi.lib()
    assert L.mi355kkt_test_throw(3) == 0
    assert L.mi355kkt_test_throw(0) == _capi.ENOMEM
    assert "out of host memory" in _capi.last_error()
    assert L.mi355kkt_test_throw(1) == _capi.EHIP
    assert "requested by the caller" in _capi.last_error()
    assert L.mi355kkt_test_throw(2) == _capi.EHIP
    with pytest.raises(MemoryError):
        _capi.check(L.mi355kkt_test_throw(0), "test_throw")


def test_every_knob_the_sources_read_is_documented_in_the_test_header():
    """knobs were used in the sources that include/mi355kkt_test.h did not name.  Every dev_knob("...") literal of csrc/
    must appear in the header's list."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "mi355kkt_test.h")).read()
    used = set()
    for path in glob.glob(os.path.join(root, "cvxopt_amd", "csrc", "*")):
        if path.endswith((".hip", ".cpp", ".h")):
            used |= set(re.findall(r'dev_knob\("(MI355KKT_[A-Z0-9_]+)"\)', open(path).read()))
    assert used, "no knob found: the pattern of this test is stale"
    missing = sorted(k for k in used if '"%s"' % k not in header)
    assert not missing, missing


def test_brk_heap_detection_used_by_the_pinned_upload():
    """round 6: mi355kkt_set_H_dense_async never hipHostRegisters memory of the brk heap; the detector behind that decision:
    a small malloc'ed block lies in the heap, an anonymous mmap does not, a huge NumPy array (mmap'ed by glibc) does not"""
    import ctypes as C
    import mmap
    from cvxopt_amd import _capi
    L = _capi.lib()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.free.argtypes = [C.c_void_p]
    p = libc.malloc(1000)
    try:
        assert L.mi355kkt_test_touches_brk_heap(C.c_void_p(p), 1000) == 1
    finally:
        libc.free(p)
    mm = mmap.mmap(-1, 1 << 20)
    buf = (C.c_char * (1 << 20)).from_buffer(mm)
    assert L.mi355kkt_test_touches_brk_heap(C.addressof(buf), 1 << 20) == 0
    del buf
    mm.close()
    big = np.empty(40 << 20, dtype=np.uint8)            # 40 MB: above glibc's largest dynamic mmap threshold (32 MB)
    assert L.mi355kkt_test_touches_brk_heap(C.c_void_p(big.ctypes.data), big.nbytes) == 0
    # a range that straddles the end of the heap counts as touching it
    assert L.mi355kkt_test_touches_brk_heap(C.c_void_p(0), 1 << 62) == 1

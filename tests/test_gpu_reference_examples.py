"""The reference's examples/book scripts (the figures of Boyd & Vandenberghe: lp / qp / socp / sdp / cp / gp on the book's data)
as a drop-in check with RESULTS, not only statuses: every script is run twice from the same seed -- on the host reference and
with cvxopt.solvers' drivers replaced by cvxopt_amd.solvers' (device-resident loops, GPU factories) -- and every dense matrix the
script leaves in its globals (solutions, trade-off curves, fitted coefficients ...) is compared.  Staged sourceless by
oracle/build_ref.sh (byte-compiled scripts + their pickled data files)."""
import contextlib
import io
import marshal
import os
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOOK = os.path.join(ROOT, "oracle", "_ref", "reftests", "examples", "book")
SOLVER_NAMES = ("conelp", "coneqp", "lp", "qp", "socp", "sdp", "cp", "cpl", "gp")


def _examples():
    out = []
    if os.path.isdir(BOOK):
        for chap in sorted(os.listdir(BOOK)):
            for f in sorted(os.listdir(os.path.join(BOOK, chap))):
                if f.endswith(".pyc"):
                    out.append(chap + "/" + f[:-4])
    return out or ["missing"]


def _run(example):
    """executes the staged script in a fresh globals dict (cwd = its directory: the data files are opened by relative name; pylab
    made unimportable: the scripts guard their plots with try / except ImportError); returns its dense 'd' matrices as arrays"""
    import cvxopt
    chap, name = example.split("/")
    with open(os.path.join(BOOK, chap, name + ".pyc"), "rb") as f:
        f.read(16)
        code = marshal.load(f)
    cwd = os.getcwd()
    saved = sys.modules.get("pylab", "absent")
    sys.modules["pylab"] = None
    cvxopt.setseed(11)
    g = {"__name__": "__example__"}
    try:
        os.chdir(os.path.join(BOOK, chap))
        with contextlib.redirect_stdout(io.StringIO()):
            try:
                exec(code, g)
            except NameError as e:                  # (chap6/robls.py plots unconditionally at its very end)
                if "pylab" not in str(e):
                    raise
    finally:
        os.chdir(cwd)
        if saved == "absent":
            sys.modules.pop("pylab", None)
        else:
            sys.modules["pylab"] = saved
    out = {}
    for k, v in g.items():
        if isinstance(v, cvxopt.matrix) and v.typecode == 'd' and not k.startswith("_"):
            out[k] = np.array(v)
    return out


@pytest.mark.parametrize("example", _examples())
def test_book_example_same_results_through_the_backend(ref_cvxopt, example):
    if example == "missing":
        pytest.fail("oracle/_ref/reftests/examples/book missing: run `bash oracle/build_ref.sh` where /root/reference exists")
    from cvxopt import solvers
    import cvxopt_amd.solvers as gs
    old = dict(solvers.options)
    solvers.options['show_progress'] = False
    calls = {}

    def counted(name, fn):
        def f(*a, **k):
            calls[name] = calls.get(name, 0) + 1
            return fn(*a, **k)
        return f
    try:
        t0 = time.perf_counter()
        ref = _run(example)
        t1 = time.perf_counter()
        saved = {n: getattr(solvers, n) for n in SOLVER_NAMES}
        try:
            for n in SOLVER_NAMES:
                setattr(solvers, n, counted(n, getattr(gs, n)))
            got = _run(example)
        finally:
            for n, f in saved.items():
                setattr(solvers, n, f)
        t2 = time.perf_counter()
    finally:
        solvers.options.clear()
        solvers.options.update(old)
    assert set(got) == set(ref)
    worst, which = 0.0, None
    for k in ref:
        assert got[k].shape == ref[k].shape, k
        if ref[k].size == 0:
            continue
        scale = max(1.0, float(np.max(np.abs(ref[k]))))
        err = float(np.max(np.abs(got[k] - ref[k]))) / scale
        if err > worst:
            worst, which = err, k
    print("%s: %d matrices, %s solver calls, worst deviation %.1e (%s); host %.1f s, backend %.1f s"
          % (example, len(ref), calls, worst, which, t1 - t0, t2 - t1))
    # interior-point solutions are accurate to the solvers' own tolerances (abstol 1e-7, reltol 1e-6 on gaps and residuals): the two
    # runs follow the same iterates, so they agree far tighter than that on well-posed problems; 1e-5 leaves room for quantities
    # that amplify the last digits (dual variables of nearly degenerate constraints)
    assert worst <= 1e-5, (which, worst)

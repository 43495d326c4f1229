"""The reference's OWN test-suite for this path, run through the GPU backend (VERDICT r3 "missing" 2): tests/test_custom_kkt.py,
tests/test_examples.py (examples/doc/chap8-10: conelp, coneqp, lp, socp, sdp, mcsdp, l1, l1regls, gp, acent, acent2, l2ac, the
modeling examples) and tests/test_modeling.py (incl. boeing2.mps) of cvxopt, unmodified -- staged sourceless under
oracle/_ref/reftests by oracle/build_ref.sh -- in the two drop-in modes:

    install   cvxopt_amd.install(): the reference's drivers, with misc.kkt_chol / chol2 / ldl / ldl2 / qr rebound to the GPU factories
    solvers   cvxopt.solvers.conelp / coneqp / lp / qp / socp / sdp / cp / cpl / gp replaced by cvxopt_amd.solvers' (device loops)

and, as the control, unpatched (the reference on the host).  Every reference test must pass in every mode, and the modes must
actually have used the GPU (the factories / device loops count their calls)."""
import importlib.machinery
import importlib.util
import marshal
import os
import sys
import unittest

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFTESTS = os.path.join(ROOT, "oracle", "_ref", "reftests")
MODULES = ("test_custom_kkt", "test_examples", "test_modeling")
SOLVER_NAMES = ("conelp", "coneqp", "lp", "qp", "socp", "sdp", "cp", "cpl", "gp")


def _load_module(name):
    path = os.path.join(REFTESTS, "tests", name + ".pyc")
    loader = importlib.machinery.SourcelessFileLoader("reftests_" + name, path)
    spec = importlib.util.spec_from_loader(loader.name, loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


def _exec_example(self, example):
    """TestExamples.exec_example (reference tests/test_examples.py:9-15) for the sourceless staging: same globals dict protocol"""
    with open(os.path.join(REFTESTS, "examples", example + "c"), "rb") as f:
        f.read(16)
        code = marshal.load(f)
    gdict = dict()
    exec(code, gdict)
    return gdict


def _run_all():
    suite = unittest.TestSuite()
    names = []
    for name in MODULES:
        mod = _load_module(name)
        if hasattr(mod, "TestExamples"):
            mod.TestExamples.exec_example = _exec_example
        tests = unittest.defaultTestLoader.loadTestsFromModule(mod)
        suite.addTests(tests)
        names += [t.id() for group in tests for t in group]
    with open(os.devnull, "w") as null:
        old = sys.stdout
        sys.stdout = null                      # the examples print their solutions
        try:
            res = unittest.TextTestRunner(stream=null, verbosity=0).run(suite)
        finally:
            sys.stdout = old
    return res, names


class _Counter(object):
    def __init__(self):
        self.calls = {}

    def wrap(self, name, fn):
        def counted(*a, **k):
            self.calls[name] = self.calls.get(name, 0) + 1
            return fn(*a, **k)
        counted.__name__ = getattr(fn, "__name__", name)
        return counted


@pytest.fixture(scope="module")
def staged(ref_cvxopt):
    if not os.path.isdir(os.path.join(REFTESTS, "tests")):
        pytest.fail("oracle/_ref/reftests missing: run `bash oracle/build_ref.sh` where /root/reference exists")
    ref_cvxopt.solvers.options['show_progress'] = False
    return ref_cvxopt


def _report(res):
    return "\n".join("%s\n%s" % (t.id(), tb) for t, tb in res.failures + res.errors)


def test_reference_suite_unpatched_is_green_here(staged):
    """control: the staged tests pass against the reference itself (oracle/_ref: MKL + the SciPy-backed cholmod shim)"""
    res, names = _run_all()
    assert res.wasSuccessful(), _report(res)
    assert res.testsRun == len(names) >= 20
    assert any("test_ch9_acent" in n for n in names) and any("test_loadfile" in n for n in names)


def test_reference_suite_with_gpu_factories_installed(staged):
    """cvxopt_amd.install(): every string-named kktsolver of the reference drivers is GPU backed"""
    import cvxopt.misc as misc
    from cvxopt_amd import kkt
    cnt = _Counter()
    kkt.install()
    try:
        for f in ("kkt_chol", "kkt_chol2", "kkt_ldl", "kkt_ldl2", "kkt_qr"):
            setattr(misc, f, cnt.wrap(f, getattr(misc, f)))
        res, names = _run_all()
    finally:
        kkt.uninstall()
    assert res.wasSuccessful(), _report(res)
    assert res.testsRun == len(names)
    # the suite reaches the dense LP-cone engine, the sparse engine (modeling / acent: spmatrix G) and the q / s cone flavours
    assert cnt.calls.get("kkt_chol2", 0) >= 5 and (cnt.calls.get("kkt_chol", 0) + cnt.calls.get("kkt_qr", 0)) >= 3, cnt.calls
    assert misc.kkt_chol2.__module__ == "cvxopt.misc"


def test_reference_suite_through_cvxopt_amd_solvers(staged):
    """cvxopt.solvers' drivers replaced by cvxopt_amd.solvers': device-resident conelp / coneqp loops, GPU factories for cp / cpl / gp"""
    from cvxopt import solvers
    import cvxopt_amd.solvers as gs
    cnt = _Counter()
    saved = {n: getattr(solvers, n) for n in SOLVER_NAMES}
    dev = _Counter()
    saved_dev = (gs._kkt.conelp_device, gs._kkt.coneqp_device)
    gs._kkt.conelp_device = dev.wrap("conelp_device", saved_dev[0])
    gs._kkt.coneqp_device = dev.wrap("coneqp_device", saved_dev[1])
    try:
        for n in SOLVER_NAMES:
            setattr(solvers, n, cnt.wrap(n, getattr(gs, n)))
        res, names = _run_all()
    finally:
        for n, f in saved.items():
            setattr(solvers, n, f)
        gs._kkt.conelp_device, gs._kkt.coneqp_device = saved_dev
    assert res.wasSuccessful(), _report(res)
    assert res.testsRun == len(names)
    for n in ("conelp", "coneqp", "lp", "socp", "sdp", "cp"):
        assert cnt.calls.get(n, 0) >= 1, cnt.calls
    assert dev.calls.get("conelp_device", 0) >= 5 and dev.calls.get("coneqp_device", 0) >= 1, dev.calls

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_count():
    try:
        from cvxopt_amd import _capi
        return _capi.device_count()
    except Exception:
        return 0


def pytest_collection_modifyitems(config, items):
    # GPU tests must FAIL (not skip) on a GPU box whose library is missing; on CPU-only hosts they are
    # deselected by `-m "not gpu"`.  If someone runs them anyway without a GPU, skip with a clear reason.
    if _gpu_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def ref_cvxopt():
    """The real reference (oracle/_ref, built by oracle/build_ref.sh)."""
    from oracle import refloader
    if not refloader.available():
        pytest.skip("oracle/_ref not built (bash oracle/build_ref.sh needs /root/reference)")
    cvx = refloader.load()
    cvx.solvers.options['show_progress'] = False
    return cvx


@pytest.fixture(scope="session")
def capi():
    from cvxopt_amd import _capi
    _capi.lib()
    return _capi


def to_matrix(cvx, a):
    return cvx.matrix(np.asfortranarray(np.atleast_2d(a.T).T if a.ndim == 1 else a))

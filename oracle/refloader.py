"""TEST INFRASTRUCTURE ONLY: import the *real* reference (cvxopt built by oracle/build_ref.sh).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this module.
The product (cvxopt_amd/) never imports cvxopt or anything under oracle/; it accepts cvxopt
matrices through the buffer protocol only.
"""
import os
import sys

_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def available():
    return os.path.isdir(os.path.join(_REF, "cvxopt"))


def load():
    """Returns the reference `cvxopt` package (raises ImportError when oracle/_ref is missing)."""
    if not available():
        raise ImportError("oracle/_ref/cvxopt missing: run `bash oracle/build_ref.sh` "
                          "(needs /root/reference; prebuilt files travel to the GPU box)")
    # The reference is linked against MKL's single dynamic library; its default Intel-OpenMP threading
    # layer silently corrupts results when libgomp (torch) lives in the same process.  Must be set
    # before libmkl_rt is first used.
    os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
    if _REF not in sys.path:
        sys.path.insert(0, _REF)
    import cvxopt
    import cvxopt.solvers  # noqa: F401
    import cvxopt.misc     # noqa: F401
    return cvxopt

"""tests/golden/book_examples.npz: every number the reference's examples/book scripts leave in their globals when run on the HOST
reference (oracle/_ref) from a fixed seed -- the expected values of tests/test_gpu_reference_examples.py.

    bash oracle/build_ref.sh && python tests/golden/make_golden_book.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import refloader            # noqa: E402

cvx = refloader.load()
cvx.solvers.options['show_progress'] = False
import test_gpu_reference_examples as t   # noqa: E402

out = t.reference_results()
np.savez_compressed(os.path.join(HERE, "book_examples.npz"), **out)
print("%d variables of %d examples, %d numbers" % (len(out), len({k.split("::")[0] for k in out}), sum(v.size for v in out.values())))

#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- builds the *real* reference (cvxopt @ /root/reference) into
# oracle/_ref/ so tests and bench.py's cpu_baseline leg can run the reference CPU kktsolvers.
#
# Nothing is copied into the tracked tree: the C sources are compiled where they lie under
# /root/reference/src/C (reference build lines: setup.py:92-258, dense modules only), the
# reference's Python drivers are byte-compiled (sourceless .pyc) into oracle/_ref/cvxopt/, and
# oracle/_ref/ is git-ignored.  CHOLMOD/UMFPACK/AMD need SuiteSparse (third-party, not vendored,
# CI pin v7.11.0: .github/workflows/linux_build.yml:12) which is absent here -> the `cholmod`
# module is replaced by oracle/cholmod_shim.py (SciPy-backed, ours).
#
# BLAS/LAPACK: MKL's single dynamic library (LP64 Fortran symbols dpotrf_ ...) at /opt/conda/lib.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${CVXOPT_REFERENCE:-/root/reference}"
OUT="$HERE/_ref/cvxopt"
if [ ! -d "$REF/src/C" ]; then
  echo "build_ref: $REF not present (GPU box?) -- keeping prebuilt $OUT" >&2
  exit 0
fi
PYINC="$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')"
SUFFIX="$(python3 -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')"
MKLDIR="${MKLDIR:-/opt/conda/lib}"
mkdir -p "$OUT"
build() { # name srcs...
  local name=$1; shift
  local tgt="$OUT/$name$SUFFIX"
  local newest; newest=$(ls -t "$@" "$REF/src/C/cvxopt.h" "$REF/src/C/misc.h" | head -1)
  if [ -f "$tgt" ] && [ "$tgt" -nt "$newest" ]; then return; fi
  echo "build_ref: $name"
  gcc -O2 -fPIC -shared -w -I"$PYINC" -I"$REF/src/C" "$@" -o "$tgt" \
      -L"$MKLDIR" -Wl,-rpath,"$MKLDIR" -lmkl_rt -lm
}
build base         "$REF/src/C/base.c" "$REF/src/C/dense.c" "$REF/src/C/sparse.c"
build blas         "$REF/src/C/blas.c"
build lapack       "$REF/src/C/lapack.c"
build misc_solvers "$REF/src/C/misc_solvers.c"
# Python drivers: byte-compile only (no source copies).
python3 - "$REF/src/python" "$OUT" <<'PY'
import sys, os, py_compile
src, out = sys.argv[1], sys.argv[2]
for f in sorted(os.listdir(src)):
    if f.endswith('.py') and f not in ('msk.py',):
        py_compile.compile(os.path.join(src, f), cfile=os.path.join(out, f + 'c'),
                           dfile='cvxopt/' + f, doraise=True, optimize=0)
PY
cp "$HERE/cholmod_shim.py" "$OUT/cholmod.py"
# The reference's OWN test-suite and documentation examples for this path (tests/test_custom_kkt.py, test_examples.py,
# test_modeling.py; examples/doc/chap8-10), staged sourceless like the drivers: byte-compiled .pyc files + the one data file
# test_modeling reads.  tests/test_gpu_reference_suite.py runs them through the GPU backend on the box.
python3 - "$REF" "$HERE/_ref/reftests" <<'PY'
import sys, os, py_compile, shutil
ref, out = sys.argv[1], sys.argv[2]
os.makedirs(os.path.join(out, "tests"), exist_ok=True)
for f in ("test_custom_kkt.py", "test_examples.py", "test_modeling.py"):
    py_compile.compile(os.path.join(ref, "tests", f), cfile=os.path.join(out, "tests", f + "c"), dfile="reftests/tests/" + f,
                       doraise=True, optimize=0)
shutil.copyfile(os.path.join(ref, "tests", "boeing2.mps"), os.path.join(out, "tests", "boeing2.mps"))
for chap in ("chap8", "chap9", "chap10"):
    src = os.path.join(ref, "examples", "doc", chap)
    dst = os.path.join(out, "examples", "doc", chap)
    os.makedirs(dst, exist_ok=True)
    for f in sorted(os.listdir(src)):
        if f.endswith(".py"):
            py_compile.compile(os.path.join(src, f), cfile=os.path.join(dst, f + "c"), dfile="reftests/examples/doc/%s/%s" % (chap, f),
                               doraise=True, optimize=0)
# examples/book (Boyd & Vandenberghe figures: lp / qp / socp / sdp / cp / gp on real data): byte-compiled scripts + their pickled
# data files (*.bin: cvxopt matrices, data not source); tests/test_gpu_reference_examples.py runs each on the host reference and
# through cvxopt_amd.solvers and compares every matrix the script leaves behind
for chap in ("chap4", "chap6", "chap7", "chap8"):
    src = os.path.join(ref, "examples", "book", chap)
    dst = os.path.join(out, "examples", "book", chap)
    os.makedirs(dst, exist_ok=True)
    for f in sorted(os.listdir(src)):
        if f.endswith(".py"):
            py_compile.compile(os.path.join(src, f), cfile=os.path.join(dst, f + "c"), dfile="reftests/examples/book/%s/%s" % (chap, f),
                               doraise=True, optimize=0)
        elif f.endswith(".bin"):
            shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
PY
echo "build_ref: done -> $OUT (+ $HERE/_ref/reftests)"

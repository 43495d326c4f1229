#!/usr/bin/env bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer build of libmi355kkt (the reference has no
# sanitizer CI; this is the backend's).  Device code is compiled as usual (-fno-gpu-sanitize); every host path of the C ABI
# -- argument checks, the symbolic analysis / ordering (csrc/ordering.cpp, sparse_chol.hip), the SYRK work-list builder,
# the host twins of the cone operations -- runs instrumented under the CPU test-suite:
#     bash tools/asan_host.sh            # builds /tmp/mi355kkt_asan/libmi355kkt.so and runs the CPU tests against it
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/cvxopt_amd/csrc"
OUT="${ASAN_OUT:-/tmp/mi355kkt_asan}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
mkdir -p "$OUT"
SAN="-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -g -O1"
pids=()
for f in gemm_f64 potrf blas2 cone_scale sparse_chol batch_ipm conelp_ipm coneqp_ipm capi; do
  ( "$HIPCC" --offload-arch=gfx950 -std=c++17 -fPIC $SAN -I"$ROOT/include" -c "$SRC/$f.hip" -o "$OUT/$f.o" ) &
  pids+=($!)
done
for f in ordering knobs devmem; do
  ( "$HIPCC" -std=c++17 -fPIC $SAN -c "$SRC/$f.cpp" -o "$OUT/$f.o" ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC $SAN -o "$OUT/libmi355kkt.so" "$OUT"/*.o
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
echo "built $OUT/libmi355kkt.so (runtime $RT)"
cd "$ROOT"
# python itself is not instrumented: preload the runtime, do not treat the interpreter's own leaks as failures
CVXOPT_AMD_LIB="$OUT/libmi355kkt.so" CVXOPT_AMD_NO_TORCH_PRELOAD=1 LD_PRELOAD="$RT" \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest -q -m "not gpu" tests/test_capi_load.py tests/test_ordering_cpu.py tests/test_sparse_symbolic_cpu.py \
      tests/test_sparse_plan_cpu.py tests/test_syrk_plan_cpu.py tests/test_cone_ops_cpu.py tests/test_sdp_ops_cpu.py "$@"

#!/usr/bin/env bash
# Host-side ThreadSanitizer build of libmi355kkt: the SPMD 's'-block operations of csrc/cone_ops_s.h run by teams of host
# threads (mi355kkt_test_sdp_op_host_team, a pthread barrier as the team barrier) under TSan -- a missing barrier or a data
# race between the threads of a team shows up here without a GPU.  Second part: the threaded host analysis of the sparse engine.
#     bash tools/tsan_host.sh
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/cvxopt_amd/csrc"
OUT="${TSAN_OUT:-/tmp/mi355kkt_tsan}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
mkdir -p "$OUT"
SAN="-fsanitize=thread -fno-gpu-sanitize -fno-omit-frame-pointer -g -O1"
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
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.tsan-x86_64.so | head -1)"
echo "built $OUT/libmi355kkt.so (runtime $RT)"
cd "$ROOT"
CVXOPT_AMD_LIB="$OUT/libmi355kkt.so" CVXOPT_AMD_NO_TORCH_PRELOAD=1 LD_PRELOAD="$RT" \
  TSAN_OPTIONS="halt_on_error=1 report_signal_unsafe=0 exitcode=66" \
  python -m pytest -q -m "not gpu" tests/test_sdp_ops_cpu.py -k "team" "$@"
# the multi-threaded host analysis of the sparse engine (the two sides of every separator of the nested dissection on separate
# threads, the two ordering candidates side by side, the parallel key sort of the assembly lists).  NumPy's own OpenBLAS
# threads are not instrumented and would be reported as races: one BLAS thread.  (-s: TSan writes to fd 2.)
OPENBLAS_NUM_THREADS=1 MKL_NUM_THREADS=1 OMP_NUM_THREADS=1 \
CVXOPT_AMD_LIB="$OUT/libmi355kkt.so" CVXOPT_AMD_NO_TORCH_PRELOAD=1 LD_PRELOAD="$RT" \
  TSAN_OPTIONS="halt_on_error=1 report_signal_unsafe=0 exitcode=66" \
  python -m pytest -q -s -m "not gpu" tests/test_ordering_cpu.py tests/test_sparse_plan_cpu.py "$@"

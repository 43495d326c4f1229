#!/usr/bin/env bash
# Builds libmi355kkt.so (HIP kernels + C ABI) for gfx950, in-tree, next to the Python package.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libmi355kkt.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
OBJ="$HERE/.obj"; mkdir -p "$OBJ"
pids=()
for f in gemm_f64 potrf blas2 cone_scale sparse_chol batch_ipm conelp_ipm coneqp_ipm capi; do
  src="$HERE/$f.hip"; obj="$OBJ/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/kkt_common.h" -nt "$obj" ] || [ "$HERE/cone_ops.h" -nt "$obj" ] || [ "$HERE/cone_ops_s.h" -nt "$obj" ] || [ "$HERE/ordering.h" -nt "$obj" ] || [ "$HERE/../../include/mi355kkt.h" -nt "$obj" ]; then
    ( "$HIPCC" $FLAGS -c "$src" -o "$obj" ) &
    pids+=($!)
  fi
done
# host-only sources (no device code)
for f in ordering; do
  src="$HERE/$f.cpp"; obj="$OBJ/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$HERE/$f.h" -nt "$obj" ]; then
    ( "$HIPCC" -O3 -std=c++17 -fPIC -Wall -c "$src" -o "$obj" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ"/gemm_f64.o "$OBJ"/potrf.o "$OBJ"/blas2.o "$OBJ"/cone_scale.o "$OBJ"/sparse_chol.o "$OBJ"/batch_ipm.o "$OBJ"/conelp_ipm.o "$OBJ"/coneqp_ipm.o "$OBJ"/capi.o "$OBJ"/ordering.o
echo "built $OUT"

#!/usr/bin/env bash
# Builds libmi355kkt.so (HIP kernels + C ABI) for gfx950, in-tree, next to the Python package.
#
#   build.sh            incremental: an object is recompiled when its source, any header of csrc/ or include/, or this script
#                       is newer than it; prints "compiled K of N objects" so a caller can tell a rebuild from a no-op
#   build.sh --clean    removes every object and the library first (a full rebuild: ~2 min)
#   build.sh --debug    builds cvxopt_amd/libmi355kkt_debug.so with -DMI355KKT_DEBUG (include/mi355kkt_debug.h: developer switches,
#                       environment fall-back of the knobs) into its own object directory; never loaded unless $CVXOPT_AMD_LIB names it
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
CLEAN=0; DEBUG=0
for a in "$@"; do
  case "$a" in
    --clean) CLEAN=1 ;;
    --debug) DEBUG=1 ;;
    *) echo "build.sh: unknown argument $a" >&2; exit 2 ;;
  esac
done
if [ "$DEBUG" = 1 ]; then
  OUT="$HERE/../libmi355kkt_debug.so"; OBJ="$HERE/.obj_debug"; DEFS="-DMI355KKT_DEBUG"
else
  OUT="$HERE/../libmi355kkt.so"; OBJ="$HERE/.obj"; DEFS=""
fi
if [ "$CLEAN" = 1 ]; then rm -rf "$OBJ" "$OUT"; fi
mkdir -p "$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $DEFS"
HIP_SRCS="gemm_f64 potrf blas2 trsv512 cone_scale sparse_chol batch_ipm conelp_ipm coneqp_ipm capi"
HOST_SRCS="ordering knobs devmem"
# every header an object may depend on (coarse on purpose: a header edit rebuilds everything that could include it)
DEPS=("$HERE"/*.h "$HERE/../../include"/*.h "$HERE/build.sh")
stale() {   # $1 = object, $2 = source
  [ ! -f "$1" ] && return 0
  [ "$2" -nt "$1" ] && return 0
  for d in "${DEPS[@]}"; do [ "$d" -nt "$1" ] && return 0; done
  return 1
}
pids=(); compiled=0; total=0
for f in $HIP_SRCS; do
  total=$((total + 1))
  if stale "$OBJ/$f.o" "$HERE/$f.hip"; then
    ( "$HIPCC" $FLAGS -c "$HERE/$f.hip" -o "$OBJ/$f.o" ) &
    pids+=($!); compiled=$((compiled + 1))
  fi
done
for f in $HOST_SRCS; do   # host-only sources (no device code)
  total=$((total + 1))
  if stale "$OBJ/$f.o" "$HERE/$f.cpp"; then
    ( "$HIPCC" -O3 -std=c++17 -fPIC -Wall $DEFS -c "$HERE/$f.cpp" -o "$OBJ/$f.o" ) &
    pids+=($!); compiled=$((compiled + 1))
  fi
done
fail=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || fail=1; }; done
if [ "$fail" = 1 ]; then echo "build.sh: compilation failed" >&2; exit 1; fi
if [ "$compiled" -gt 0 ] || [ ! -f "$OUT" ]; then
  objs=(); for f in $HIP_SRCS $HOST_SRCS; do objs+=("$OBJ/$f.o"); done
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}"
fi
echo "build.sh: compiled $compiled of $total objects -> $OUT"

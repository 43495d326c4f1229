#!/usr/bin/env bash
# round-4 call 17: the GPU suite against the host-AddressSanitizer build of the library (device code as shipped), stderr not captured:
# hunting the heap corruption behind the abort of call r4c16 (glibc aborted inside set_G_csc's allocation, book example chap7/probbounds)
export PYTHONPATH=.
export PYTHONFAULTHANDLER=1
O=gpurun_out/r4c17; mkdir -p $O
RT="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
export CVXOPT_AMD_LIB=$PWD/cvxopt_amd/libmi355kkt_asan.so
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=1:protect_shadow_gap=0:log_path=$PWD/$O/asan
# (torch does not initialise under the preloaded sanitizer runtime: the three tests that need it are left out; no -x)
( LD_PRELOAD="$RT" timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider --deselect tests/test_gpu_maxsize.py --deselect tests/test_gpu_stress.py \
    --deselect tests/test_gpu_lifecycle.py --deselect tests/test_gpu_batch.py::test_batchkkt_takes_problem_data_already_in_hbm \
    --deselect tests/test_gpu_batch.py::test_sharded_batch_on_rccl ) > $O/suite.log 2>&1
echo "suite rc=$? last=$(cat gpurun_out/pytest_last_test.txt 2>/dev/null)" > $O/summary.txt; tail -3 $O/suite.log | cut -c1-300 >> $O/summary.txt
grep -n "ERROR: AddressSanitizer\|SUMMARY: AddressSanitizer" $O/suite.log | head >> $O/summary.txt
ls $O >> $O/summary.txt
for f in $O/asan.*; do [ -f "$f" ] && head -60 "$f" >> $O/summary.txt; done
cat $O/summary.txt

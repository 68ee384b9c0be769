#!/bin/bash
# Host-side sanitizer runs.  CPU ONLY: run in the build container, never through gpurun (the pool
# refuses GPU sanitizers; nothing here touches a device -- the library's DEVICE code is compiled as always, its HOST
# code is instrumented: -fsanitize=... -fno-gpu-sanitize).
#
#   bash scripts/sanitize_host.sh [log=profiles/r05_host_sanitizers.log]
#
#   1. AddressSanitizer + UBSan: libferreus_bbfmm_hip_asan.so (python -m ferreus_rbf_rs_amd.build --sanitize=asan), loaded
#      instead of the product's library (FERREUS_BBFMM_HIP_LIB) with clang's ASan runtime preloaded, under the whole
#      `pytest -m "not gpu"` suite: host-only trees, lists, operators, M2L tables, plans and partitions, DDM, solvers, the
#      pool's concurrent-build and fork tests, the two-process gloo tests.
#   2. ThreadSanitizer: libferreus_bbfmm_hip_tsan.so under the tests that drive the host thread pool (csrc/parallel.hpp)
#      from several threads / handles, and scripts/host_pool_microbench.cpp built with -fsanitize=thread.
#   3. oracle/passes.c under gcc -fsanitize=address,undefined (ORACLE_PASSES_SANITIZE=1, gcc's libasan preloaded) for the
#      tests that drive the C passes of the oracle.
# Leak detection is off (CPython does not free everything at exit); everything else halts on the first report.
set -u
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r05_host_sanitizers.log}
mkdir -p "$(dirname "$LOG")"
: > "$LOG"
say() { echo "$@" | tee -a "$LOG"; }
fail=0

say "== host sanitizers, $(date -u +%Y-%m-%dT%H:%MZ), sources $(python -c 'import bench; print(bench.source_hash())' 2>/dev/null)"
say "== compiler: $(/opt/rocm/bin/hipcc --version | grep -m1 -i 'clang version')"

# ---------------------------------------------------------------------------------------------- 1. ASan + UBSan
python -m ferreus_rbf_rs_amd.build --sanitize=asan > /tmp/sanitize_build_asan.log 2>&1 || { say "asan build FAILED"; tail -20 /tmp/sanitize_build_asan.log | tee -a "$LOG"; exit 1; }
ASAN_LIB=$PWD/ferreus_rbf_rs_amd/libferreus_bbfmm_hip_asan.so
ASAN_RT=$(python -c "from ferreus_rbf_rs_amd import build as b; print(b.sanitizer_runtime('asan'))")
say "== 1. AddressSanitizer + UndefinedBehaviorSanitizer: $(basename "$ASAN_LIB"), runtime $ASAN_RT"
say "   flags: -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-gpu-sanitize -shared-libsan"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:abort_on_error=0:exitcode=99
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:exitcode=98
FERREUS_BBFMM_HIP_LIB=$ASAN_LIB LD_PRELOAD=$ASAN_RT python - >> "$LOG" 2>&1 <<'EOF'
import ferreus_rbf_rs_amd._lib as L
L.load()
maps = open("/proc/self/maps").read()
assert "libferreus_bbfmm_hip_asan.so" in maps and "libclang_rt.asan" in maps
assert "libferreus_bbfmm_hip.so" not in maps
print("   loaded:", L.LIB_PATH, "(the product's library is NOT mapped)")
EOF
[ $? -eq 0 ] || { say "   the sanitized library did not load"; fail=1; }
FERREUS_BBFMM_HIP_LIB=$ASAN_LIB LD_PRELOAD=$ASAN_RT timeout 7200 python -m pytest tests/ -q -m "not gpu" -p no:cacheprovider 2>&1 | grep -v "^\[Gloo\]" | tail -25 | tee -a "$LOG"
rc=${PIPESTATUS[0]}
say "   exit code $rc"
[ "$rc" -eq 0 ] || fail=1

# ---------------------------------------------------------------------------------------------- 2. TSan
python -m ferreus_rbf_rs_amd.build --sanitize=tsan > /tmp/sanitize_build_tsan.log 2>&1 || { say "tsan build FAILED"; tail -20 /tmp/sanitize_build_tsan.log | tee -a "$LOG"; exit 1; }
TSAN_LIB=$PWD/ferreus_rbf_rs_amd/libferreus_bbfmm_hip_tsan.so
TSAN_RT=$(python -c "from ferreus_rbf_rs_amd import build as b; print(b.sanitizer_runtime('tsan'))")
say "== 2. ThreadSanitizer: $(basename "$TSAN_LIB"), runtime $TSAN_RT"
# die_after_fork=0: the fork test starts a pool in the child of a threaded parent on purpose (TSan refuses that by default: a limit
# of its runtime, not a report)
# numpy's OpenBLAS is not instrumented (scripts/tsan_suppressions.txt): single-threaded here, and suppressed by library
export TSAN_OPTIONS=halt_on_error=1:exitcode=97:report_signal_unsafe=0:die_after_fork=0:suppressions=$PWD/scripts/tsan_suppressions.txt
FERREUS_BBFMM_HIP_LIB=$TSAN_LIB LD_PRELOAD=$TSAN_RT OPENBLAS_NUM_THREADS=1 timeout 7200 python -m pytest tests/test_ddm_tree.py tests/test_host_structure.py tests/test_partition_upward.py tests/test_solvers.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -25 | tee -a "$LOG"
rc=${PIPESTATUS[0]}
say "   exit code $rc"
[ "$rc" -eq 0 ] || fail=1
say "   scripts/host_pool_microbench.cpp with -fsanitize=thread (the pool of csrc/parallel.hpp alone, loops from two threads):"
/opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=thread -pthread -I ferreus_rbf_rs_amd/csrc scripts/host_pool_microbench.cpp -o /tmp/host_pool_microbench_tsan >> "$LOG" 2>&1 \
  && BBFMM_HOST_THREADS=8 timeout 1800 /tmp/host_pool_microbench_tsan 2>&1 | tail -8 | tee -a "$LOG"
rc=${PIPESTATUS[0]}
say "   exit code $rc"
[ "$rc" -eq 0 ] || fail=1

# ---------------------------------------------------------------------------------------------- 3. the oracle's C passes
GCC_ASAN=$(gcc -print-file-name=libasan.so)
say "== 3. oracle/passes.c: gcc -O1 -g -fsanitize=address,undefined (runtime $GCC_ASAN)"
ORACLE_PASSES_SANITIZE=1 LD_PRELOAD=$GCC_ASAN python - >> "$LOG" 2>&1 <<'EOF'
from oracle import bbfmm_oracle as O
O.lib()
assert "liboracle_passes_asan.so" in open("/proc/self/maps").read()
print("   loaded: oracle/_build/liboracle_passes_asan.so")
EOF
[ $? -eq 0 ] || { say "   the sanitized oracle did not load"; fail=1; }
ORACLE_PASSES_SANITIZE=1 LD_PRELOAD=$GCC_ASAN timeout 7200 python -m pytest tests/test_oracle_vs_dense.py tests/test_oracle_fixtures.py tests/test_oracle_ddm.py tests/test_albatite.py tests/test_solvers.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -8 | tee -a "$LOG"
rc=${PIPESTATUS[0]}
say "   exit code $rc"
[ "$rc" -eq 0 ] || fail=1

say "== RESULT: $([ $fail -eq 0 ] && echo 'no sanitizer report, every run green' || echo 'FAILURES above')"
exit $fail

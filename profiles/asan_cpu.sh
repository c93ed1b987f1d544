#!/bin/bash
# Sanitizer build of the CPU-side C / C++ (SURVEY 5 "race detection / sanitizers"; GPU ASan and XNACK are not available on this pool):
#   * oracle/c/*.c                                  -> oracle/_build_asan/liboracle.so    (gcc -fsanitize=address,undefined)
#   * the host code inside libroam_hip.so - csrc/blobprune_host.hip (roam_prune_blobs, roam_argsort_np122: csrc/blobprune.h) and
#     csrc/pngdec.hip + csrc/fastinflate.h (roam_png_*) - compiled as plain C++ into variants/libroam_host_asan.so
# and the CPU tests that drive them run against those libraries (python itself is not instrumented: libasan is preloaded).
# usage: bash profiles/asan_cpu.sh [log]      exit code 0 = no report
set -u
cd "$(dirname "$0")/.."
LOG=${1:-profiles/r06_asan_cpu.log}
SAN="-fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -g -O1"
mkdir -p oracle/_build_asan variants
gcc $SAN -fPIC -shared -std=c11 -ffp-contract=off -fno-fast-math -D_GNU_SOURCE -o oracle/_build_asan/liboracle.so oracle/c/*.c -lm || exit 2
g++ $SAN -fPIC -shared -std=c++17 -ffp-contract=off -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Wno-unused-result \
    radarslampy_amd/csrc/blobprune_host.hip radarslampy_amd/csrc/pngdec.hip -o variants/libroam_host_asan.so -lz -lpthread || exit 2
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=66 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1:exitcode=66
export ORACLE_LIB="$PWD/oracle/_build_asan/liboracle.so" ROAM_LIB="$PWD/variants/libroam_host_asan.so" ROAM_LIB_PARTIAL=1
{
  echo "== $(date -u +%FT%TZ)  gcc $(gcc -dumpversion), flags: $SAN"
  python -m pytest -q -x -p no:cacheprovider -m "not gpu" tests/test_oracle_golden.py tests/test_oracle_known_answers.py tests/test_oracle_clique_order.py \
      tests/test_oracle_reference_dump.py tests/test_oracle_tiny_traj.py tests/test_set_table_model.py tests/test_parallel_partition_model.py \
      tests/test_png_native_cpu.py
  rc=$?
  echo "== pytest exit code $rc"
} > "$LOG" 2>&1
grep -c "ERROR: AddressSanitizer\|runtime error:" "$LOG" | sed 's/^/sanitizer reports: /' | tee -a "$LOG"
tail -3 "$LOG"
exit $rc

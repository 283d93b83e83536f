#!/bin/bash
# the CPU restatement under AddressSanitizer + UndefinedBehaviorSanitizer (host only: GPU sanitizers are
# not available on this pool).  Builds oracle/_build/asan/*.so and runs the oracle-facing CPU tests on them.
#   bash tools/asan_cpu_suite.sh            (from the repository root)
set -e -o pipefail
make -C oracle asan
ASAN_LIB=$(gcc -print-file-name=libasan.so)
UBSAN_LIB=$(gcc -print-file-name=libubsan.so)
# Python itself is not instrumented: leak detection off (the interpreter "leaks" by design), the runtime preloaded.
export KPO_BUILD_DIR=_build/asan
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export OMP_NUM_THREADS=${OMP_NUM_THREADS:-4}
LD_PRELOAD="$ASAN_LIB $UBSAN_LIB" python3 -m pytest tests/test_oracle_cpu.py tests/test_host_cpu.py -x -q "$@"

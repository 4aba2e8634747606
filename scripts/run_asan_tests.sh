#!/bin/bash
# CPU sanitizer run: builds the host half of the library, the cycle model, the spada-sim front end and the oracle
# with -fsanitize=address,undefined (make -C spada_sim_amd/csrc asan) and runs the non-GPU test files that exercise them
# under it.  The interpreter itself is not instrumented, so libasan is preloaded and leak detection (which would report
# CPython's own allocations) is off; every other ASan / UBSan finding aborts the test.  tests/test_multi_rank_gloo.py is left out:
# it tests spada_sim_amd/parallel.py (Python over torch.distributed), no native code of this repository.
#   scripts/run_asan_tests.sh [log file]      (default profiles/r05_asan_cpu_tests.txt)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LOG="${1:-$ROOT/profiles/r05_asan_cpu_tests.txt}"
make -s -C "$ROOT/spada_sim_amd/csrc" all asan
make -s -C "$ROOT/oracle" liboracle_spgemm.so asan
ASAN_SO="$(gcc -print-file-name=libasan.so)"
# (libstdc++ next to it: the interpreter is not a C++ program, and ASan resolves its __cxa_throw interceptor when it starts)
export LD_PRELOAD="$ASAN_SO $(gcc -print-file-name=libstdc++.so)"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
export SPADA_LIB_PATH="$ROOT/spada_sim_amd/lib_asan/libspada_spgemm.so"
# (the exchange library as its sanitizer build against the HIP / RCCL / engine test doubles of csrc/mock/: spada_comm_plan and the host
# code of both exchange forms run instrumented, and nothing dlopens the HIP or RCCL runtime under the preloaded libasan)
export SPADA_COMM_LIB_PATH="$ROOT/spada_sim_amd/lib_asan/libspada_comm_mock.so"
export SPADA_COMM_MOCK_LIB_PATH="$ROOT/spada_sim_amd/lib_asan/libspada_comm_mock.so"
export SPADA_ORACLE_LIB_PATH="$ROOT/oracle/liboracle_spgemm_asan.so"
export SPADA_BIN_PATH="$ROOT/spada_sim_amd/lib_asan/spada-sim"
cd "$ROOT"
{
  echo "# $(date -u +%Y-%m-%dT%H:%M:%SZ)  gcc $(gcc -dumpversion)  -fsanitize=address,undefined  LD_PRELOAD=$ASAN_SO"
  echo "# SPADA_LIB_PATH=$SPADA_LIB_PATH  SPADA_ORACLE_LIB_PATH=$SPADA_ORACLE_LIB_PATH  SPADA_BIN_PATH=$SPADA_BIN_PATH"
  python -m pytest tests/test_host_cpu.py tests/test_oracle_golden.py tests/test_cycle_model.py -x -q -m "not gpu" -p no:cacheprovider 2>&1
} | tee "$LOG"

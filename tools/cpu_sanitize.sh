#!/bin/bash
# tools/cpu_sanitize.sh — the host-side C/C++ under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only; GPU
# sanitizers are not available on the pool): the config / NetCDF host code through csim_hosttool
# (tests/test_host_config_snapshot.py) and the oracle's C restatement (tests/test_oracle_golden.py).
# Builds go to $TMP; nothing in the tree is replaced.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TMP=${TMPDIR:-/tmp}/csim_sanitize
mkdir -p "$TMP"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
make -s -C "$R/climate-sim-mpi-cpp_amd/csrc"
( cd "$R/climate-sim-mpi-cpp_amd/driver" &&
  g++ -std=c++17 $SAN -Wall -I../../include -o "$TMP/csim_hosttool" hosttool.cpp config.cpp snapshot.cpp compat.cpp \
      -L../lib -lcsim -Wl,-rpath,"$R/climate-sim-mpi-cpp_amd/lib" -Wl,-rpath,/opt/rocm/lib )
gcc -std=gnu11 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread $SAN -o "$TMP/liboracle_cpu.so" "$R/oracle/cpu_stepper.c" -lm
export ASAN_OPTIONS=halt_on_error=1:detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
cd "$R"
CSIM_HOSTTOOL="$TMP/csim_hosttool" python -m pytest tests/test_host_config_snapshot.py -x -q
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" CSIM_ORACLE_SO="$TMP/liboracle_cpu.so" \
    python -m pytest tests/test_oracle_golden.py tests/test_multirank_gloo.py -x -q -p no:cacheprovider

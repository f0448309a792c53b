#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side native code (GPU sanitizers are not available on this pool): the operator-level host layer
# (libgolap_host.so, device-free plan arms) and the CPU oracle, each rebuilt with -fsanitize=address,undefined into /tmp and run through
# their CPU test files.  usage (repo root, no GPU needed): bash scripts/sanitize_cpu.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/golap_san
mkdir -p $OUT
LIBASAN=$(gcc -print-file-name=libasan.so)
SAN="-O1 -g -std=c++17 -fPIC -Wall -Wextra -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -shared"
make -C $ROOT/gpu_olap_engine_amd/csrc -s
g++ $SAN -o $OUT/libgolap_host_asan.so $ROOT/gpu_olap_engine_amd/host/golap_executor.cpp -I$ROOT/include -L$ROOT/gpu_olap_engine_amd/lib -lgolap_hip \
    -Wl,-rpath,$ROOT/gpu_olap_engine_amd/lib -Wl,--allow-shlib-undefined
(cd $ROOT && GOLAP_HOST_DEV_LIB=$OUT/libgolap_host_asan.so LD_PRELOAD=$LIBASAN ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests/test_host_cpu.py -x -q)
# the oracle is loaded from its own directory: swap the sanitizer build in, run, put the normal build back
cp $ROOT/oracle/libgolap_oracle.so $OUT/libgolap_oracle_orig.so
trap 'cp $OUT/libgolap_oracle_orig.so $ROOT/oracle/libgolap_oracle.so; touch $ROOT/oracle/libgolap_oracle.so' EXIT
g++ $SAN -o $ROOT/oracle/libgolap_oracle.so $ROOT/oracle/golap_oracle.cpp && touch $ROOT/oracle/libgolap_oracle.so
(cd $ROOT && LD_PRELOAD=$LIBASAN ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 python -m pytest tests/test_oracle.py -x -q)
echo "sanitize_cpu: no reports"

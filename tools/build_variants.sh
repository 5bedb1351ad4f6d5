#!/bin/bash
# Diagnostic builds of the library with parts of the fast GEMM disabled (timing only, results are wrong).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/variants
for v in ${@:-NO_EPILOGUE NO_LOADS NO_MFMA}; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -munsafe-fp-atomics -shared -DCTN_DBG_$v \
     -o tools/variants/libctn_$v.so speech_amd/csrc/*.hip &
done
wait
ls -la tools/variants

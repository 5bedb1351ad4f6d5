#!/bin/bash
# Times the prepared-operand GEMM kernels in every diagnostic build under tools/variants (see build_variants2.sh).
cd "$(dirname "$0")/.."
for lib in tools/variants/libctn_*.so; do
  CTN_LIB=$PWD/$lib timeout -k 10 120 python tools/diag_prep_gemm.py ${1:-32} || exit 1
done

#!/bin/bash
# usage: tools/ab_gemm.sh <alt lib under tools/_bin> [bench_gemm args]  -- same-box A/B of the GEMM micro-benchmark: shipped lib vs a variant
alt=$1; shift
echo "== new"; python tools/bench_gemm.py --no-blas --rounds 3 --variants "pp:" "$@" 2>&1 | grep -v amdgpu
echo "== alt $alt"; SPN_LIB=tools/_bin/$alt python tools/bench_gemm.py --no-blas --rounds 3 --variants "pp:" "$@" 2>&1 | grep -v amdgpu
echo "== new"; python tools/bench_gemm.py --no-blas --rounds 3 --variants "pp:" "$@" 2>&1 | grep -v amdgpu

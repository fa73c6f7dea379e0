#!/bin/bash
# A/B of the 256x128 8-wave GEMM tile: tools/bench_big.sh
for big in 0 1 0 1; do CA_GEMM_BIG=$big python tools/bench_gemm.py 2>&1 | grep -v amdgpu | sed "s/\$/ BIG=$big/"; done

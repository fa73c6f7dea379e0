#!/bin/bash
# A/B of the 32-wide double-buffered K stage: tools/bench_kt.sh
for kt in 64 32 64 32; do CA_GEMM_KT=$kt python tools/bench_gemm.py 2>&1 | grep -v amdgpu | sed "s/\$/ KT=$kt/"; done

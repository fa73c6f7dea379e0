#!/bin/bash
# Ablation timing of the attention kernel (results are NOT valid attention outputs): rebuilds the
# attention object with -DCA_ATTN_ABLATE=n and runs the micro-benchmark.
set -e
cd "$(dirname "$0")/.."
for n in 0 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -DCA_ATTN_ABLATE=$n -c controlanimate_amd/csrc/ca_attention.hip -o /tmp/ca_attention_abl.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o controlanimate_amd/csrc/libcontrolanimate_hip.so controlanimate_amd/csrc/ca_gemm.o controlanimate_amd/csrc/ca_norm.o /tmp/ca_attention_abl.o controlanimate_amd/csrc/ca_elementwise.o
  echo "ABLATE=$n: $(python tools/bench_attn.py | grep 'attn images')"
done

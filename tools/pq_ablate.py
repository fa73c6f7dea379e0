"""Timing-only ablations of the 256 x 320 streaming kernel (ca_gemm_pq.h, CA_PQ_ABLATE): where the time of a launch goes.

Build one library per ablation (results of an ablated kernel are WRONG by construction; these libraries are never shipped):

    cd controlanimate_amd/csrc && python -m controlanimate_amd._build   # the other objects
    for v in 1 4 5 8; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -DCA_PQ_ABLATE=$v -c ca_gemm_pp.hip -o /tmp/pp$v.o &&
      hipcc --offload-arch=gfx950 -shared -fPIC -o libca_abl$v.so ca_gemm.o /tmp/pp$v.o ca_norm.o ca_attention.o ca_elementwise.o; done
    python tools/pq_ablate.py; for v in 1 4 5 8; do CA_HIP_LIB=$PWD/controlanimate_amd/csrc/libca_abl$v.so python tools/pq_ablate.py; done

bits: 1 = no global -> LDS units after a block's first two, 2 = no fragment reads (registers keep pseudo-random bit patterns:
full-entropy operands, which by themselves slow the MFMAs down -- tools/gemm_data_power.py), 4 = no MFMAs, 8 = no barriers
inside the K loop, 16 = every tile streams the operands of tile (0, 0) (an L2-resident stream: round 5).  Results of round 3: DESIGN.md section 3.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controlanimate_amd import kernels as K
DEV = "cuda"
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
tag = os.path.basename(os.environ.get("CA_HIP_LIB", "product"))
out = []
for (m, n, k) in [(32768, 640, 2560), (131072, 320, 1280), (32768, 1280, 1280), (8192, 10240, 1280), (32768, 5120, 640)]:
    a = torch.randn(m, k, device=DEV).half(); w = (torch.randn(n, k, device=DEV) * k ** -0.5).half(); b = torch.randn(n, device=DEV)
    K._plan_sink = lab = []; K.gemm(a, w, bias=b); K._plan_sink = None
    ms = timeit(lambda: K.gemm(a, w, bias=b))
    out.append(f"{m}x{n}x{k} {ms*1e3:7.1f}us {2.0*m*n*k/ms/1e9:6.0f}TF {lab[0]}")
for (img, h, ci, co) in [(32, 32, 640, 640), (32, 32, 1280, 1280)]:
    x = torch.randn(img, h, h, ci, device=DEV).half(); w = (torch.randn(co, 3, 3, ci, device=DEV) * (9 * ci) ** -0.5).half(); b = torch.randn(co, device=DEV)
    K._plan_sink = lab = []; K.conv3x3(x, w, bias=b); K._plan_sink = None
    ms = timeit(lambda: K.conv3x3(x, w, bias=b))
    out.append(f"conv{h} {ci}->{co} {ms*1e3:7.1f}us {2.0*img*h*h*co*9*ci/ms/1e9:6.0f}TF {lab[0]}")
print(f"{tag:22s} | " + " | ".join(out), flush=True)

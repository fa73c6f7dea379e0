"""Builds the gfx950 shared library with hipcc (in-tree, no JIT cache).

    python -m controlanimate_amd._build [--force] [--experiments]

The library is `controlanimate_amd/csrc/libcontrolanimate_hip.so`; it is git-ignored but
travels to the GPU box with the repo snapshot.

`--experiments` builds a SECOND library, `csrc/libcontrolanimate_hip_exp.so`, with -DCA_EXPERIMENTS: the same ABI plus
the tuning environment variables (CA_KNOB in ca_common.h) and the kernels that never became defaults
(csrc/experiments/ca_gemm_pp.h, ca_gemm_pp3.h).  It is loaded through CA_HIP_LIB for same-box A/B timing and is never the product path.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libcontrolanimate_hip.so")
SOURCES = ["ca_gemm.hip", "ca_gemm_pp.hip", "ca_gemm_ar.hip", "ca_norm.hip", "ca_attention.hip", "ca_elementwise.hip"]
EXPERIMENT_HEADERS = [os.path.join("experiments", "ca_gemm_pp.h"), os.path.join("experiments", "ca_gemm_pp3.h")]  # -DCA_EXPERIMENTS builds only
HEADERS = ["ca_common.h", "ca_gemm_core.h", "ca_gemm_pp2.h", "ca_gemm_wres.h", "ca_gemm_ps.h", "ca_gemm_pq.h", "ca_gemm_ar.h", "ca_ff_fused.h", "ca_attn_out.h", "ca_tattn_fused.h", "ca_xattn_fused.h", "ca_gemm_seq.h", "ca_conv_wino.h", os.path.join("..", "..", "include", "controlanimate_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         # keep MFMA accumulators in the (unified) VGPR file: without it hipcc parks them in AGPRs and the
         # attention kernel spends ~200 v_accvgpr_read/write per K/V tile on the softmax rescale
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


# per-file extra flags
EXTRA = {"ca_attention.hip": ["-fno-honor-nans"]}  # see vmax3 in ca_common.h


def _extra_env_flags() -> list:
    """CA_HIPCC_FLAGS="-DX=1 ..." appends flags (timing experiments, e.g. the CA_GEMM_ABLATE switches)."""
    return os.environ.get("CA_HIPCC_FLAGS", "").split()


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.sep not in cand or os.path.exists(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(out: str, deps: list[str]) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


LIB_EXP = os.path.join(CSRC, "libcontrolanimate_hip_exp.so")
LIB_STAMPS = os.path.join(CSRC, "libcontrolanimate_hip_stamps.so")  # --experiments --stamps: knobs + s_memtime stamps (tools/*_stamps.py)


def build(force: bool = False, verbose: bool = True, experiments: bool = False, stamps: bool = False) -> str:
    hipcc = _hipcc()
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS + (EXPERIMENT_HEADERS if experiments else [])]
    jobs = []
    objs = []
    objdir = os.path.join(CSRC, "build_stamps" if stamps else "build_exp") if experiments else CSRC
    os.makedirs(objdir, exist_ok=True)
    lib = (LIB_STAMPS if stamps else LIB_EXP) if experiments else LIB
    xflags = ["-DCA_EXPERIMENTS"] if experiments else []
    if stamps:
        xflags.append("-DCA_STAMPS")
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc, *FLAGS, *xflags, *EXTRA.get(s, []), *_extra_env_flags(), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, flush=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, experiments="--experiments" in sys.argv, stamps="--stamps" in sys.argv))

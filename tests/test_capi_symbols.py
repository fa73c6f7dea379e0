"""CPU: the C-ABI library builds/loads and exports exactly what include/controlanimate_hip.h
declares; ctypes struct layouts equal the C layouts (sizeof + every field offset, via gcc)."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "controlanimate_hip.h")


@pytest.fixture(scope="module")
def capi():
    from controlanimate_amd import _build, _capi
    _build.build(verbose=False)
    return _capi


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ca_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(capi):
    lib = capi.lib()
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(capi.SYMBOLS) == names, "binding table and header disagree"
    assert lib.ca_abi_version() == capi.ABI_VERSION


def test_ctypes_structs_match_c_layout(capi):
    structs = {"ca_gemm_args": capi.GemmArgs, "ca_conv_args": capi.ConvArgs, "ca_groupnorm_args": capi.GroupNormArgs,
               "ca_layernorm_args": capi.LayerNormArgs, "ca_attn_args": capi.AttnArgs}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for cname, st in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "l.c"), os.path.join(d, "l")
        open(src, "w").write("\n".join(lines))
        subprocess.check_call(["gcc", "-o", exe, src])
        out = subprocess.check_output([exe], text=True)
    got = dict(l.split() for l in out.strip().splitlines())
    for cname, st in structs.items():
        assert int(got[cname]) == C.sizeof(st), cname
        for fname, _ in st._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(st, fname).offset, f"{cname}.{fname}"


def test_invalid_arguments_are_rejected_without_a_gpu(capi):
    """Argument validation happens before any launch: usable as a no-GPU check of the error path."""
    lib = capi.lib()
    args = capi.GemmArgs(m=0, n=0, k1=0, k2=0)
    rc = lib.ca_gemm(C.byref(args), None)
    assert rc == -1 and b"ca_gemm" in lib.ca_last_error()
    assert lib.ca_groupnorm_partials_floats(32, 4096, 1, 32) == 32 * 16 * 32 * 2   # 256-row chunks
    assert lib.ca_groupnorm_partials_floats(2, 4096, 2, 32) == 1 * 32 * 32 * 2
    assert lib.ca_groupnorm_partials_floats(32, 64, 1, 32) == 32 * 16 * 32 * 2  # small images: 4 rows per chunk


def test_every_entry_point_rejects_bad_arguments_without_a_gpu(capi):
    """The reference raises Python exceptions on malformed inputs; the C ABI returns a negative code and a
    message naming the entry point, never crashes, never launches (all of this runs on the CPU-only builder)."""
    lib = capi.lib()
    fake = C.c_void_p(0x1000)  # never dereferenced on the host: validation fails first

    def expect(rc, who):
        assert rc < 0 and who.encode() in lib.ca_last_error(), (rc, lib.ca_last_error())

    expect(lib.ca_gemm(None, None), "ca_gemm")
    expect(lib.ca_gemm(C.byref(capi.GemmArgs(a=fake, w=fake, c=fake, m=16, n=16, k1=12, lda=16, ldc=16, dtype=1)), None), "ca_gemm")        # K % 8
    expect(lib.ca_gemm(C.byref(capi.GemmArgs(a=fake, w=fake, c=fake, m=16, n=16, k1=16, lda=16, ldc=16, dtype=7)), None), "ca_gemm")        # dtype
    expect(lib.ca_gemm(C.byref(capi.GemmArgs(a=fake, w=fake, c=fake, m=16, n=16, k1=16, lda=16, ldc=16, dtype=1, ln_stats=fake)), None), "ca_gemm")  # ln pair
    # ABI v6: ln_colsum alone = statistics inside the kernel; only the weight-resident K = 320 kernel can (M >= 16384, N % 160 == 0)
    def ln_args(m, n, k, eps=1e-5, **kw):
        return capi.GemmArgs(a=fake, w=fake, c=fake, m=m, n=n, k1=k, lda=k, ldc=n, dtype=1, alpha=1.0, post_scale=1.0, ln_colsum=fake, ln_eps=eps, **kw)
    assert lib.ca_gemm_ln_inline_supported(C.byref(ln_args(131072, 960, 320))) == 1
    assert lib.ca_gemm_ln_inline_supported(C.byref(ln_args(131072, 960, 640))) == 0      # K != 320
    assert lib.ca_gemm_ln_inline_supported(C.byref(ln_args(8192, 960, 320))) == 0        # too few rows for that kernel
    assert lib.ca_gemm_ln_inline_supported(C.byref(ln_args(131072, 328, 320))) == 0      # N % 160
    assert lib.ca_gemm_ln_inline_supported(None) == 0
    expect(lib.ca_gemm(C.byref(ln_args(8192, 960, 320)), None), "ca_gemm")               # not available there: an error, not a silent fallback
    expect(lib.ca_gemm(C.byref(ln_args(131072, 960, 320, eps=0.0)), None), "ca_gemm")    # needs ln_eps
    # ABI v6: row sums for the next LayerNorm out of the 128x320-tile kernel's epilogue; split-K scratch for long-K dense GEMMs
    def plain(m, n, k, **kw):
        return capi.GemmArgs(a=fake, w=fake, c=fake, m=m, n=n, k1=k, lda=k, ldc=n, dtype=1, alpha=1.0, post_scale=1.0, **kw)
    assert lib.ca_gemm_row_sums_parts(C.byref(plain(8192, 1280, 1280))) == 4        # 256 tiles of 128x320
    assert lib.ca_gemm_row_sums_parts(C.byref(plain(32768, 640, 640))) == 8          # ABI v8: the 256x320 kernel, one per 80-column wave quarter
    assert lib.ca_gemm_row_sums_parts(C.byref(plain(2048, 1280, 1280))) == 0        # 64 tiles: another kernel takes it
    assert lib.ca_gemm_row_sums_parts(C.byref(plain(131072, 320, 320))) == 0        # the weight-resident kernel takes it
    assert lib.ca_gemm_row_sums_parts(C.byref(plain(8192, 1280, 1280, geglu=1))) == 0
    expect(lib.ca_gemm(C.byref(plain(2048, 1280, 1280, row_sums_out=fake)), None), "ca_gemm")          # not available: an error
    expect(lib.ca_gemm(C.byref(plain(8192, 1280, 1280, ln_parts=4)), None), "ca_gemm")                 # ln_parts without the sums
    assert lib.ca_gemm_workspace_bytes(C.byref(plain(2048, 1280, 5120))) == 6 * 2048 * 1280 * 4        # 80 K tiles on 160 tiles
    assert lib.ca_gemm_workspace_bytes(C.byref(plain(2048, 1280, 1280))) == 0                          # 20 K tiles: not worth it
    assert lib.ca_gemm_workspace_bytes(None) == 0
    # N = 12 / ldc = 12 pass a "multiple of 4" check but the LDS-staged epilogue stores 16-byte chunks
    expect(lib.ca_gemm(C.byref(capi.GemmArgs(a=fake, w=fake, c=fake, m=16, n=12, k1=16, lda=16, ldc=16, dtype=1)), None), "ca_gemm")
    expect(lib.ca_gemm(C.byref(capi.GemmArgs(a=fake, w=fake, c=fake, m=16, n=16, k1=16, lda=16, ldc=12, dtype=1)), None), "ca_gemm")
    expect(lib.ca_gemm(C.byref(capi.GemmArgs(a=fake, w=fake, c=fake, residual=fake, m=16, n=16, k1=16, lda=16, ldc=16, ld_res=12, dtype=1)), None), "ca_gemm")
    expect(lib.ca_conv3x3(C.byref(capi.ConvArgs(x=fake, w=fake, y=fake, images=1, hin=8, win=8, cin1=16, cout=12, stride=1, dtype=1)), None), "ca_conv3x3")
    expect(lib.ca_conv3x3(None, None), "ca_conv3x3")
    expect(lib.ca_conv3x3(C.byref(capi.ConvArgs(x=fake, w=fake, y=fake, images=1, hin=8, win=8, cin1=12, cout=16, stride=1, dtype=1)), None), "ca_conv3x3")  # Cin % 8
    expect(lib.ca_conv3x3(C.byref(capi.ConvArgs(x=fake, w=fake, y=fake, images=1, hin=8, win=8, cin1=16, cout=16, stride=3, dtype=1)), None), "ca_conv3x3")  # stride
    assert lib.ca_conv3x3_workspace_bytes(None) == 0
    small = capi.ConvArgs(images=32, hin=8, win=8, cin1=1280, cout=1280, stride=1, dtype=1)
    assert lib.ca_conv3x3_workspace_bytes(C.byref(small)) == 6 * 2048 * 1280 * 4        # 160 tiles -> 6 K ranges of fp32 slabs
    big = capi.ConvArgs(images=32, hin=64, win=64, cin1=320, cout=320, stride=1, dtype=1)
    assert lib.ca_conv3x3_workspace_bytes(C.byref(big)) == 0
    expect(lib.ca_groupnorm(None, None), "ca_groupnorm")
    expect(lib.ca_groupnorm(C.byref(capi.GroupNormArgs(x=fake, partials=fake, images=2, hw=16, c1=64, groups=32, frames_per_stat=1, dtype=1)), None), "ca_groupnorm")  # y / gamma / beta missing
    expect(lib.ca_groupnorm_stats(C.byref(capi.GroupNormArgs(x=fake, partials=fake, images=2, hw=16, c1=30, groups=32, frames_per_stat=1, dtype=1)), None),
           "ca_groupnorm_stats")                                                                               # C % 8
    expect(lib.ca_groupnorm_apply(C.byref(capi.GroupNormArgs(x=fake, partials=fake, images=3, hw=16, c1=64, groups=32, frames_per_stat=2, dtype=1)), None),
           "ca_groupnorm_apply")                                                                               # frames_per_stat
    expect(lib.ca_layernorm(C.byref(capi.LayerNormArgs(x=fake, rows=4, c=320, dtype=1)), None), "ca_layernorm")                          # no y / stats
    expect(lib.ca_layernorm(C.byref(capi.LayerNormArgs(x=fake, stats=fake, pos=fake, rows=4, c=320, rows_per_frame=1, frames=1, dtype=1)), None), "ca_layernorm")
    expect(lib.ca_attention(C.byref(capi.AttnArgs(q=fake, k=fake, v=fake, o=fake, head_dim=20, batches=1, heads=1, nq=4, nk=4, inner_count=1,
                                                 kv_inner_count=1, kv_div=1, dtype=1)), None), "ca_attention")      # head_dim % 8
    expect(lib.ca_attention(C.byref(capi.AttnArgs(q=fake, k=fake, v=fake, o=fake, head_dim=40, batches=1, heads=1, nq=4, nk=8, inner_count=1,
                                                 kv_inner_count=1, kv_div=1, q_row=40, k_row=40, o_row=40, dtype=1, causal=1)), None), "ca_attention")  # causal nq != nk
    expect(lib.ca_softmax_rows(fake, fake, 4, 10, 12, 12, 1.0, 1, None), "ca_softmax_rows")                       # cols % 4
    ptrs, cf = (C.c_void_p * 2)(0x1000, 0x2000), (C.c_float * 2)(1.0, 2.0)
    expect(lib.ca_lincomb(fake, ptrs, cf, 9, 64, None), "ca_lincomb")                                              # > 8 terms
    expect(lib.ca_lincomb(fake, ptrs, cf, 2, 0, None), "ca_lincomb")                                               # n = 0
    expect(lib.ca_lincomb(None, ptrs, cf, 2, 64, None), "ca_lincomb")
    # ABI v8
    expect(lib.ca_repeat(fake, C.c_void_p(0x2000), 0, 2, None), "ca_repeat")                                        # bytes
    expect(lib.ca_repeat(fake, C.c_void_p(0x2008), 64, 65, None), "ca_repeat")                                      # times
    expect(lib.ca_repeat(None, fake, 64, 2, None), "ca_repeat")
    expect(lib.ca_ln_finish_sums(fake, 0, 64, 640, 1e-5, fake, None), "ca_ln_finish_sums")                          # parts
    expect(lib.ca_ln_finish_sums(None, 2, 64, 640, 1e-5, fake, None), "ca_ln_finish_sums")
    assert lib.ca_gemm_wants_finished_stats(None) == 0


def test_no_cpu_fallback_when_library_missing(monkeypatch):
    from controlanimate_amd import _capi
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", "/nonexistent/libcontrolanimate_hip.so")
    with pytest.raises(_capi.CAHipUnavailable):
        _capi.lib()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under controlanimate_amd/ may reference it."""
    pkg = os.path.join(ROOT, "controlanimate_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f

"""CPU: which kernel instantiation every GEMM / convolution shape of the benchmark workload (BASELINE config 2:
16 frames 512x512, CFG batch 2, 1 ControlNet) is dispatched to.

The product library has no tuning environment variables (CA_KNOB compiles to its default), so the choice is a pure
function of the arguments; `ca_gemm_plan_name` / `ca_conv3x3_plan_name` report it without a launch and this table pins it.
A threshold edit in `plan_gemm` (csrc/ca_gemm.hip) that silently moves a headline shape to another kernel fails here --
and the network-level parity test at the headline size (tests/test_fullsize_gpu.py::test_config2_full_size_eps_vs_oracle)
is what then has to be re-run on the GPU.
"""
import ctypes as C

import pytest


@pytest.fixture(scope="module")
def capi():
    from controlanimate_amd import _build, _capi
    _build.build(verbose=False)
    return _capi


FAKE = 0x10000  # never dereferenced: the plan only looks at sizes, flags and which pointers are set


def gemm_label(capi, m, n, k, *, geglu=0, res=False, ln=None, row_sums=False, workspace=False, k2=0, frag=False, rowbias=0):
    a = capi.GemmArgs(a=FAKE, w=FAKE, c=FAKE, m=m, n=n, k1=k - k2, k2=k2, lda=k - k2, lda2=k2, ldc=n // 2 if geglu else n,
                      alpha=1.0, post_scale=1.0, dtype=capi.CA_F16, geglu=geglu, rows_per_group=1)
    if k2:
        a.a2 = FAKE
    if res:
        a.residual, a.ld_res = FAKE, n
    if ln == "inline":
        a.ln_colsum, a.ln_eps = FAKE, 1e-5
    elif ln == "stats":
        a.ln_colsum, a.ln_stats, a.ln_eps = FAKE, FAKE, 1e-5
    elif isinstance(ln, int):
        a.ln_colsum, a.ln_stats, a.ln_eps, a.ln_parts = FAKE, FAKE, 1e-5, ln
    if row_sums:
        a.row_sums_out = FAKE
    if workspace:
        a.workspace, a.workspace_bytes = FAKE, 1 << 40
    if frag:  # the fragment-ordered twin of W (ABI v9): layers.LnFold packs one for K = 320, N >= 960
        a.w_frag = FAKE
    if rowbias:
        a.rowbias, a.rows_per_group, a.ld_rowbias = FAKE, rowbias, n
    buf = C.create_string_buffer(64)
    rc = capi.lib().ca_gemm_plan_name(C.byref(a), buf, 64)
    assert rc == 0, capi.lib().ca_last_error()
    return buf.value.decode()


def conv_label(capi, images, h, cin, cout, *, cin2=0, stride=1, upsample=0, workspace=True):
    a = capi.ConvArgs(x=FAKE, w=FAKE, y=FAKE, images=images, hin=h, win=h, cin1=cin - cin2, cin2=cin2, cout=cout, stride=stride,
                      upsample=upsample, alpha=1.0, post_scale=1.0, dtype=capi.CA_F16, rows_per_group=1)
    if cin2:
        a.x2 = FAKE
    if workspace:
        a.workspace, a.workspace_bytes = FAKE, 1 << 40
    buf = C.create_string_buffer(64)
    rc = capi.lib().ca_conv3x3_plan_name(C.byref(a), buf, 64)
    assert rc == 0, capi.lib().ca_last_error()
    return buf.value.decode()


# (M, N, K, keyword flags) -> label.  Rows = the dense launches of one config-2 denoise step (ControlNet + UNet3D,
# `bench.py --shapes`), largest time share first.  Labels: wres160 = weight-resident K = 320 kernel, ps128x320 = persistent
# streaming kernel (round 3), pq256x320 = its 256 x 320 / 128 x 80-wave-tile sibling (round 3), pp128x320 = ping-pong 128 x 320 tiles, BMxBN = k_gemm_dma tiles (_r3: a three-stage LDS ring, round 4; _db: two stages),
# _splitkS = S K ranges + reduce, reg_ = register-staged fallback, ar128x64 = activation-resident K = 320 kernel (round 4: the launches
# that hand over W in fragment order and, for in-kernel LayerNorm statistics, the 8 M bytes of scratch -- what kernels.gemm does).
GEMMS = [
    ((131072, 2560, 320, dict(geglu=1, ln="inline", frag=True, workspace=True)), "ar128x64"),  # FF projection + GEGLU, 64x64 latents
    ((131072, 320, 320, dict(res=True)), "wres160"),                  # to_out / proj_out (+ residual)
    ((131072, 320, 320, dict()), "wres160"),                          # proj_in, to_q (cross)
    ((131072, 960, 320, dict(ln="inline", frag=True, workspace=True)), "ar128x64"),            # q|k|v
    ((131072, 960, 320, dict(ln="inline", frag=True, workspace=True, rowbias=4096)), "ar128x64"),  # temporal q|k|v (+ pe W^T per frame)
    ((131072, 2560, 320, dict(geglu=1, ln="inline")), "wres160"),     # without the fragment-ordered weights: the weight-resident kernel
    ((131072, 960, 320, dict(ln="inline", frag=True)), "wres160"),    # without the statistics scratch: likewise
    ((131072, 960, 320, dict(ln="inline")), "wres160"),
    ((131072, 320, 320, dict(res=True, frag=True)), "wres160"),       # N = 320: five 64-column panels for four waves -- stays
    ((32768, 5120, 640, dict(geglu=1, ln=2)), "ps128x320"),
    ((32768, 640, 640, dict(res=True, row_sums=True)), "pq256x320"),    # (ABI v8: row sums per wave quarter, finished by ca_ln_finish_sums)
    ((32768, 640, 640, dict()), "pq256x320"),
    ((32768, 1920, 640, dict(ln=2)), "128x128"),
    ((8192, 10240, 1280, dict(geglu=1, ln=4)), "ps128x320"),
    ((8192, 1280, 1280, dict(res=True, row_sums=True)), "ps128x320"),
    ((8192, 3840, 1280, dict(ln=4)), "128x128"),
    ((131072, 320, 1280, dict(res=True)), "pq256x320"),               # FF out, 64x64 latents
    ((32768, 640, 2560, dict(res=True)), "pq256x320"),
    ((8192, 1280, 5120, dict(res=True)), "pp128x320"),
    ((2048, 1280, 1280, dict(res=True)), "128x64_r3"),
    ((2048, 1280, 5120, dict(res=True, workspace=True)), "pp128x320_splitk4"),
    ((2048, 1280, 5120, dict(res=True)), "128x64_r3"),                # no scratch handed over: unsplit
    ((2048, 3840, 1280, dict(ln="stats")), "pp128x320"),
    ((2048, 10240, 1280, dict(geglu=1, ln="stats")), "pq256x320"),
    ((131072, 320, 640, dict(k2=320)), "ps128x320"),                  # shortcut over the skip concat (K = 320 + 320)
    ((32, 1280, 320, dict()), "128x64_r3"),                           # time embedding
    ((32768, 640, 640, dict(row_sums=True)), "pq256x320"),
    # folded LayerNorm with FINISHED statistics (what kernels.gemm hands over after ca_gemm_wants_finished_stats): 256 x 320 tiles
    ((32768, 5120, 640, dict(geglu=1, ln="stats")), "pq256x320"),
    ((8192, 10240, 1280, dict(geglu=1, ln="stats")), "pq256x320"),
    ((32768, 1920, 640, dict(ln="stats")), "pq256x320"),
    ((8192, 3840, 1280, dict(ln="stats")), "128x128"),                # 384 tiles = 1.5 rounds of 256: stays
    ((8192, 1280, 5120, dict(res=True)), "pp128x320"),                # 128 tiles of 256 x 320 would leave half the CUs idle
]

# (images, H, Cin, Cout, keyword flags) -> label
CONVS = [
    ((32, 64, 320, 320, dict()), "128x160"),
    ((32, 16, 1280, 1280, dict()), "pp128x320"),
    ((32, 32, 640, 640, dict()), "pq256x320"),
    ((32, 8, 1280, 1280, dict()), "128x128_splitk6"),
    ((32, 8, 1280, 1280, dict(workspace=False)), "128x64_r3"),
    ((32, 16, 2560, 1280, dict(cin2=1280)), "pp128x320"),
    ((32, 64, 640, 320, dict(cin2=320)), "pq256x320"),
    ((32, 64, 640, 640, dict(upsample=0)), "pq256x320"),
    ((32, 32, 1280, 1280, dict()), "pq256x320"),
    ((32, 32, 1920, 640, dict(cin2=640)), "pq256x320"),
    ((32, 64, 960, 320, dict(cin2=320)), "pq256x320"),
    ((32, 64, 320, 320, dict(stride=2)), "pp128x320"),                # Downsample: 32x32 outputs
    ((32, 16, 1280, 1280, dict(upsample=1)), "128x128"),              # Upsample: 32x32 outputs
    ((32, 64, 8, 320, dict()), "reg_128x64"),                         # conv_in (4 latent channels padded to 8)
    ((32, 32, 640, 640, dict(upsample=1)), "128x128"),                # (the 256 x 320 kernel's gather has no upsampling)
]


@pytest.mark.parametrize("shape,label", GEMMS, ids=[f"gemm{m}x{n}x{k}{'_' + '_'.join(sorted(kw)) if kw else ''}" for (m, n, k, kw), _ in GEMMS])
def test_gemm_dispatch(capi, shape, label):
    m, n, k, kw = shape
    assert gemm_label(capi, m, n, k, **kw) == label


@pytest.mark.parametrize("shape,label", CONVS, ids=[f"conv{i}x{h}x{h}_{ci}to{co}{'_' + '_'.join(sorted(kw)) if kw else ''}" for (i, h, ci, co, kw), _ in CONVS])
def test_conv_dispatch(capi, shape, label):
    i, h, ci, co, kw = shape
    assert conv_label(capi, i, h, ci, co, **kw) == label


def test_partial_layernorm_sums_never_reach_a_kernel_that_reads_mean_rstd(capi):
    """ln_parts = P hands [M][P][2] partial sums over; only the tiled epilogue finishes them.  The split-K
    reduce kernel and the weight-resident kernel read ln_stats as (mean, rstd): such a launch must take neither."""
    lib = capi.lib()
    # the shape that would split (2048 x 1280 x 5120 with scratch) stays unsplit with ln_parts
    assert gemm_label(capi, 2048, 1280, 5120, ln=4, workspace=True) == "128x64_r3"
    assert gemm_label(capi, 2048, 1280, 5120, ln="stats", workspace=True) == "pp128x320_splitk4"
    # the weight-resident shape (K = 320, M >= 16384) goes to a kernel whose epilogue finishes partial sums
    assert gemm_label(capi, 131072, 960, 320, ln=1) == "ps128x320"
    assert gemm_label(capi, 131072, 960, 320, ln="stats") == "wres160"
    a = capi.GemmArgs(a=FAKE, w=FAKE, c=FAKE, m=2048, n=1280, k1=5120, lda=5120, ldc=1280, alpha=1.0, post_scale=1.0, dtype=capi.CA_F16,
                      ln_colsum=FAKE, ln_stats=FAKE, ln_eps=1e-5, ln_parts=4)
    assert lib.ca_gemm_workspace_bytes(C.byref(a)) == 0
    a.ln_parts = 0
    assert lib.ca_gemm_workspace_bytes(C.byref(a)) > 0
    # malformed: partial sums without the statistics pointer
    bad = capi.GemmArgs(a=FAKE, w=FAKE, c=FAKE, m=2048, n=1280, k1=1280, lda=1280, ldc=1280, alpha=1.0, post_scale=1.0, dtype=capi.CA_F16,
                        ln_colsum=FAKE, ln_eps=1e-5, ln_parts=4)
    buf = C.create_string_buffer(64)
    assert lib.ca_gemm_plan_name(C.byref(bad), buf, 64) < 0 and b"ln_parts" in lib.ca_last_error()


def test_consumer_of_partial_sums_is_told_when_finished_statistics_are_better(capi):
    """ABI v8: ca_gemm_wants_finished_stats -- the launches whose preferred kernel (256 x 320 tiles) reads (mean, rstd) only."""
    lib = capi.lib()

    def wants(m, n, k, parts, geglu=0):
        a = capi.GemmArgs(a=FAKE, w=FAKE, c=FAKE, m=m, n=n, k1=k, lda=k, ldc=n // 2 if geglu else n, alpha=1.0, post_scale=1.0, dtype=capi.CA_F16,
                          geglu=geglu, rows_per_group=1, ln_colsum=FAKE, ln_stats=FAKE, ln_eps=1e-5, ln_parts=parts)
        return lib.ca_gemm_wants_finished_stats(C.byref(a))

    assert wants(32768, 5120, 640, 2, geglu=1) == 1 and wants(8192, 10240, 1280, 4, geglu=1) == 1 and wants(32768, 1920, 640, 2) == 1
    assert wants(8192, 3840, 1280, 4) == 0          # keeps the 128 x 128 kernel, which finishes the sums itself
    assert wants(32768, 1920, 640, 0) == 0          # already finished
    assert wants(131072, 960, 320, 1) == 0          # K = 320: weight-resident / streaming kernels


def test_activation_resident_kernel_only_takes_what_it_implements(capi):
    """ABI v9: w_frag selects the activation-resident kernel for K = 320, M >= 16384, N % 320 == 0, N >= 960, plain epilogues;
    everything else keeps its old plan, and the statistics scratch is what ca_gemm_workspace_bytes asks for."""
    lib = capi.lib()
    assert gemm_label(capi, 131072, 1280, 320, frag=True) == "ar128x64"
    assert gemm_label(capi, 131072, 960, 320, ln="stats", frag=True) == "ar128x64"           # finished statistics: no scratch needed
    assert gemm_label(capi, 8192, 960, 320, ln="stats", frag=True) != "ar128x64"             # M < 16384
    assert gemm_label(capi, 131072, 960, 640, frag=True, k2=320) != "ar128x64"               # two sources / K = 640
    assert gemm_label(capi, 131072, 960, 320, ln="stats", res=True, frag=True) == "wres160"  # LayerNorm + residual: not implemented there
    assert gemm_label(capi, 131072, 960, 320, frag=True, rowbias=4032) == "wres160"          # row-bias groups must be whole 128-row tiles
    assert gemm_label(capi, 131072, 960, 320, ln=1, frag=True) == "ps128x320"                # partial sums: the kernels that finish them
    a = capi.GemmArgs(a=FAKE, w=FAKE, c=FAKE, m=131072, n=960, k1=320, lda=320, ldc=960, alpha=1.0, post_scale=1.0, dtype=capi.CA_F16,
                      rows_per_group=1, ln_colsum=FAKE, ln_eps=1e-5, w_frag=FAKE)
    assert lib.ca_gemm_workspace_bytes(C.byref(a)) == 131072 * 8
    assert lib.ca_gemm_ln_inline_supported(C.byref(a)) == 1
    a.w_frag = None
    assert lib.ca_gemm_workspace_bytes(C.byref(a)) == 0
    a.w_frag, a.alpha = FAKE, 0.5
    assert lib.ca_gemm_workspace_bytes(C.byref(a)) == 0  # (alpha != 1: the weight-resident kernel keeps the launch)
    # ca_pack_w_frag argument checks
    for args in ((None, 960, 320, 0, FAKE), (FAKE, 960, 640, 0, FAKE), (FAKE, 100, 320, 0, FAKE), (FAKE, 960, 320, 0, 0x10008)):
        assert lib.ca_pack_w_frag(args[0], args[1], args[2], args[3], args[4], None) < 0 and b"ca_pack_w_frag" in lib.ca_last_error()


def test_fused_feed_forward_only_takes_what_it_implements(capi):
    """ABI v9 ca_ff_fused_supported: C = 320, inner 1280, M >= 16384, 16-byte aligned operands, in-kernel statistics need ln_eps."""
    lib = capi.lib()

    def ok(**over):
        kw = dict(x=FAKE, w1_frag=FAKE, bias1=FAKE, colsum1=FAKE, w2_frag=FAKE, bias2=FAKE, residual=FAKE, y=FAKE, lda=320, ldc=320, ld_res=320,
                  m=131072, c=320, inner=1280, ln_eps=1e-5, dtype=capi.CA_F16)
        kw.update(over)
        return lib.ca_ff_fused_supported(C.byref(capi.FfArgs(**kw)))

    assert ok() == 1 and ok(residual=None, ld_res=0) == 1 and ok(bias2=None) == 1 and ok(ln_stats=FAKE, ln_eps=0.0) == 1 and ok(dtype=capi.CA_BF16) == 1
    assert ok(m=8192) == 0 and ok(c=640) == 0 and ok(inner=2560) == 0 and ok(w1_frag=None) == 0 and ok(w2_frag=None) == 0
    assert ok(lda=324) == 0 and ok(x=FAKE + 8) == 0 and ok(ln_eps=0.0) == 0 and ok(dtype=7) == 0
    assert ok(m=1 << 23, lda=320) == 0  # 32-bit byte offsets
    bad = capi.FfArgs(x=FAKE, m=100)
    assert lib.ca_ff_fused(C.byref(bad), None) < 0 and b"ca_ff_fused" in lib.ca_last_error()
    for args in ((None, 320, 1280, FAKE), (FAKE, 640, 1280, FAKE), (FAKE, 320, 1280, 0x10008)):
        assert lib.ca_pack_w2_frag(args[0], args[1], args[2], args[3], None) < 0


def test_fused_temporal_attention_only_takes_what_it_implements(capi):
    """ABI v10 ca_tattn_fused_supported: C = 320, 8 heads, 16 frames, tokens % 8 == 0, >= 16384 rows, aligned operands."""
    lib = capi.lib()

    def ok(**over):
        kw = dict(x=FAKE, w_frag=FAKE, gamma=FAKE, bias_pe=FAKE, o=FAKE, lda=320, ldo=320, ld_bias_pe=320, batch=2, frames=16, tokens=4096,
                  heads=8, c=320, ln_eps=1e-5, scale=40 ** -0.5, dtype=capi.CA_F16)
        kw.update(over)
        return lib.ca_tattn_fused_supported(C.byref(capi.TattnArgs(**kw)))

    assert ok() == 1 and ok(dtype=capi.CA_BF16) == 1 and ok(batch=1, tokens=1024) == 1 and ok(lda=640, tokens=6144) == 1
    assert ok(frames=8) == 0 and ok(frames=32) == 0 and ok(heads=4) == 0 and ok(c=640) == 0 and ok(tokens=4100) == 0
    # ABI v12: 8 and 32 frames with the output stage (the eight-wave kernel); tokens in whole tiles of 128 / frames pixels; 24 frames never
    assert ok(frames=8, w_out_frag=FAKE) == 1 and ok(frames=32, w_out_frag=FAKE, tokens=9216) == 1 and ok(frames=16, w_out_frag=FAKE, residual=FAKE, ld_res=320) == 1
    assert ok(frames=24, w_out_frag=FAKE) == 0 and ok(frames=8, w_out_frag=FAKE, tokens=4104) == 0 and ok(frames=32, w_out_frag=FAKE, tokens=4098) == 0
    assert ok(residual=FAKE, ld_res=320) == 0                      # a residual belongs to the output projection
    assert ok(w_out_frag=FAKE, residual=FAKE, ld_res=324) == 0
    assert ok(batch=1, tokens=64) == 0  # 1024 rows: not worth a launch of 512 blocks
    assert ok(w_frag=None) == 0 and ok(gamma=None) == 0 and ok(bias_pe=None) == 0 and ok(x=FAKE + 8) == 0
    assert ok(lda=324) == 0 and ok(ld_bias_pe=318) == 0 and ok(ln_eps=0.0) == 0 and ok(scale=0.0) == 0 and ok(dtype=7) == 0
    assert ok(batch=64, tokens=1 << 16) == 0  # 32-bit byte offsets
    bad = capi.TattnArgs(x=FAKE, frames=3)
    assert lib.ca_tattn_fused(C.byref(bad), None) < 0 and b"ca_tattn_fused" in lib.ca_last_error()
    for args in ((None, 960, 320, FAKE), (FAKE, 640, 320, FAKE), (FAKE, 960, 320, 0x10008)):
        assert lib.ca_pack_w_tattn(args[0], args[1], args[2], args[3], None) < 0


def test_fused_text_cross_attention_only_takes_what_it_implements(capi):
    """ABI v11 ca_xattn_fused_supported: C = 320, 8 heads, 65..80 keys, tokens % 128 == 0, >= 16384 rows, aligned operands."""
    lib = capi.lib()

    def ok(**over):
        kw = dict(x=FAKE, wq_frag=FAKE, bias=FAKE, kv_frag=FAKE, o=FAKE, lda=320, ldo=320, m=131072, tokens=4096, frames_per_kv=16, kv_mod=2,
                  kv_batches=2, nk=77, heads=8, c=320, ln_eps=1e-5, dtype=capi.CA_F16)
        kw.update(over)
        return lib.ca_xattn_fused_supported(C.byref(capi.XattnArgs(**kw)))

    assert ok() == 1 and ok(dtype=capi.CA_BF16) == 1 and ok(bias=None) == 1 and ok(nk=80) == 1 and ok(m=16384, tokens=1024) == 1 and ok(lda=640) == 1
    assert ok(nk=64) == 0 and ok(nk=81) == 0 and ok(heads=4) == 0 and ok(c=640) == 0 and ok(tokens=4100) == 0 and ok(m=131072 + 128) == 0
    assert ok(m=8192, tokens=1024) == 0 and ok(frames_per_kv=0) == 0
    # image z reads text batch (z // frames_per_kv) % kv_mod: what counts is the largest index USED (32 images, 16 per prompt: 0 and 1)
    assert ok(kv_mod=3) == 1 and ok(kv_mod=32) == 1          # (kv_mod = images is kernels.xattn_fused's default, as attention_cross)
    assert ok(kv_mod=3, frames_per_kv=8) == 0 and ok(kv_mod=32, frames_per_kv=8) == 0 and ok(kv_batches=1) == 0
    assert ok(wq_frag=None) == 0 and ok(kv_frag=None) == 0 and ok(x=FAKE + 8) == 0 and ok(lda=324) == 0 and ok(ln_eps=0.0) == 0 and ok(dtype=7) == 0
    assert ok(m=1 << 23, tokens=4096) == 0  # 32-bit byte offsets
    bad = capi.XattnArgs(x=FAKE, m=5)
    assert lib.ca_xattn_fused(C.byref(bad), None) < 0 and b"ca_xattn_fused" in lib.ca_last_error()
    for args in ((None, 320, 320, FAKE), (FAKE, 640, 320, FAKE), (FAKE, 320, 320, 0x10008)):
        assert lib.ca_xattn_pack_w(args[0], args[1], args[2], args[3], None) < 0
    assert lib.ca_xattn_pack_kv(FAKE, 640, 2, 77, 0, 64, 0.158, capi.CA_F16, FAKE, None) < 0   # 64 keys: not this kernel's
    assert lib.ca_xattn_pack_kv(FAKE, 640, 2, 77, 4, 77, 0.158, capi.CA_F16, FAKE, None) < 0   # rows beyond the batch
    # ABI v13: the IP-Adapter's image-prompt tokens (1..16 of them) ride on the form with the output stage only
    assert ok(kv_frag_ip=FAKE, nk_ip=4, ip_scale=1.0) == 0
    out = dict(w_out_frag=FAKE, bias_out=FAKE, residual=FAKE, ld_res=320)
    assert ok(**out) == 1 and ok(kv_frag_ip=FAKE, nk_ip=4, ip_scale=1.0, **out) == 1 and ok(kv_frag_ip=FAKE, nk_ip=16, ip_scale=0.0, **out) == 1
    assert ok(kv_frag_ip=FAKE, nk_ip=0, ip_scale=1.0, **out) == 0 and ok(kv_frag_ip=FAKE, nk_ip=17, ip_scale=1.0, **out) == 0
    assert ok(kv_frag_ip=FAKE + 8, nk_ip=4, ip_scale=1.0, **out) == 0 and ok(kv_frag_ip=FAKE, nk_ip=4, ip_scale=float("nan"), **out) == 0
    assert ok(nk_ip=4, **out) == 0                                                              # a count without the fragments
    assert lib.ca_xattn_pack_kv(FAKE, 640, 2, 81, 77, 17, 0.158, capi.CA_F16, FAKE, None) < 0  # 17 image-prompt tokens: not taken


def attn_label(capi, *, images, nq, nk, heads, d, kind="self", accumulate=0, causal=0, mask=False):
    """Strides as kernels.attention_spatial / attention_cross / attention_temporal hand them over."""
    c = heads * d
    if kind == "self":          # q | k | v rows of width 3C
        a = capi.AttnArgs(q=FAKE, k=FAKE, v=FAKE, o=FAKE, q_outer=nq * 3 * c, q_row=3 * c, o_outer=nq * c, o_row=c, k_outer=nq * 3 * c, k_row=3 * c,
                          inner_count=1, kv_inner_count=1, kv_div=1, kv_mod=images)
    elif kind == "cross":       # q rows of width C, k | v rows of width 2C, one K/V per 16 frames
        a = capi.AttnArgs(q=FAKE, k=FAKE, v=FAKE, o=FAKE, q_outer=nq * c, q_row=c, o_outer=nq * c, o_row=c, k_outer=nk * 2 * c, k_row=2 * c,
                          inner_count=1, kv_inner_count=1, kv_div=16, kv_mod=2)
    else:                       # temporal: images = b * tokens sequences of `nq` frames, rows tokens * 3C apart
        tokens = images // 2
        a = capi.AttnArgs(q=FAKE, k=FAKE, v=FAKE, o=FAKE, q_outer=nq * tokens * 3 * c, q_inner=3 * c, q_row=tokens * 3 * c, o_outer=nq * tokens * c,
                          o_inner=c, o_row=tokens * c, k_outer=nq * tokens * 3 * c, k_inner=3 * c, k_row=tokens * 3 * c, inner_count=tokens,
                          kv_inner_count=tokens, kv_div=1, kv_mod=images)
    a.batches, a.heads, a.head_dim, a.nq, a.nk, a.scale, a.out_scale = images, heads, d, nq, nk, d ** -0.5, 1.0
    a.accumulate, a.causal, a.dtype = accumulate, causal, capi.CA_F16
    if mask:
        a.key_mask, a.key_mask_stride = FAKE, nk
    buf = C.create_string_buffer(64)
    rc = capi.lib().ca_attention_plan_name(C.byref(a), buf, 64)
    assert rc == 0, capi.lib().ca_last_error()
    return buf.value.decode()


ATTN_TABLE = [
    # the attentions of a config-2 step that are still separate launches (ABI v12: ca_attention_plan_name)
    (dict(images=32, nq=4096, nk=4096, heads=8, d=40), "attn_dma40"),       # spatial self-attention, 64x64 latents
    (dict(images=32, nq=1024, nk=1024, heads=8, d=80), "attn_dma80"),       # 32x32 latents
    (dict(images=32, nq=256, nk=256, heads=8, d=160), "attn_generic"),      # 16x16 latents
    (dict(images=32, nq=64, nk=64, heads=8, d=160), "attn_generic"),        # mid block
    (dict(images=32, nq=4096, nk=77, heads=8, d=40, kind="cross"), "attn_short"),
    (dict(images=32, nq=1024, nk=77, heads=8, d=80, kind="cross"), "attn_short"),
    (dict(images=32, nq=256, nk=77, heads=8, d=160, kind="cross"), "attn_generic"),
    (dict(images=32, nq=4096, nk=4, heads=8, d=40, kind="cross", accumulate=1), "attn_generic"),   # IP-Adapter pass (config 4)
    (dict(images=2 * 1024, nq=16, nk=16, heads=8, d=80, kind="temporal"), "attn_tiny16"),   # motion modules below the 64x64 level
    (dict(images=2 * 256, nq=32, nk=32, heads=8, d=160, kind="temporal"), "attn_tiny32"),   # config 5: 32 frames
    (dict(images=2, nq=77, nk=77, heads=12, d=64, causal=1), "attn_generic"),              # CLIP text encoder
    (dict(images=1, nq=257, nk=257, heads=16, d=80), "attn_dma80"),                       # CLIP ViT-H vision tower
    (dict(images=2, nq=512, nk=512, heads=8, d=40, mask=True), "attn_generic"),            # a key mask keeps the generic kernel
]


@pytest.mark.parametrize("kw,label", ATTN_TABLE, ids=[f"{l}-{k['nq']}x{k['nk']}-d{k['d']}" for k, l in ATTN_TABLE])
def test_attention_dispatch(capi, kw, label):
    assert attn_label(capi, **kw) == label


def test_winograd_route_of_the_deep_small_latent_convolutions(capi):
    """ABI v12: with the Winograd weights and the workspace the 16x16- and 8x8-latent convolutions with >= 1280 input channels report
    "wino_pq256x320"; without either, with a shallow input or stride 2 they keep the direct kernels."""
    def label(images, h, cin, cout, *, cin2=0, wino=True, workspace=True, stride=1, upsample=0, dtype=None):
        a = capi.ConvArgs(x=FAKE, w=FAKE, y=FAKE, images=images, hin=h, win=h, cin1=cin - cin2, cin2=cin2, cout=cout, stride=stride,
                          upsample=upsample, alpha=1.0, post_scale=1.0, dtype=capi.CA_F16 if dtype is None else dtype, rows_per_group=1)
        if cin2:
            a.x2 = FAKE
        if wino:
            a.w_wino = FAKE
        if workspace:
            a.workspace, a.workspace_bytes = FAKE, 1 << 40
        buf = C.create_string_buffer(64)
        assert capi.lib().ca_conv3x3_plan_name(C.byref(a), buf, 64) == 0, capi.lib().ca_last_error()
        return buf.value.decode(), int(capi.lib().ca_conv3x3_workspace_bytes(C.byref(a)))
    assert label(32, 16, 1280, 1280) == ("wino_pq256x320", 16 * 2048 * (1280 + 1280) * 2)
    assert label(32, 16, 2560, 1280, cin2=1280)[0] == "wino_pq256x320"
    assert label(32, 16, 1920, 1280, cin2=640)[0] == "wino_pq256x320"
    assert label(32, 8, 1280, 1280)[0] == "wino_pq256x320"
    assert label(32, 16, 1280, 1280, wino=False)[0] == "pp128x320"
    assert label(32, 16, 1280, 1280, workspace=False)[0] == "pp128x320"
    assert label(32, 32, 1280, 1280)[0] == "wino_pq256x320"             # 8192 tiles: 634 vs 800 us (tools/wino_check.py)
    assert label(32, 16, 1280, 1280, upsample=1)[0] == "wino_pq256x320"  # Upsample3D: the nearest x2 folded into the input transform
    assert not label(64, 64, 1280, 1280)[0].startswith("wino")          # 65536 tiles: beyond the window
    assert label(32, 16, 640, 1280)[0] == "wino_pq256x320"              # 640 input channels pay at <= 4096 tiles (100 vs 160 us) ...
    assert not label(32, 32, 640, 640)[0].startswith("wino")            # ... not at 8192 (254 vs 233-252: sixteen K = 640 GEMMs are epilogue-bound)
    assert not label(32, 16, 320, 640)[0].startswith("wino")
    assert not label(32, 16, 1280, 1280, stride=2)[0].startswith("wino")
    assert not label(32, 3, 1280, 1280, upsample=1)[0].startswith("wino")  # 6 x 6 logical: 288 tiles, not whole 256-row GEMM tiles
    assert label(32, 16, 1280, 1280, dtype=capi.CA_BF16)[0] == "wino_pq256x320"   # (bf16: the transforms in fp32 arithmetic)

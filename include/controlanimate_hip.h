/*
 * controlanimate_hip.h -- C ABI of the MI355X (gfx950) denoising-loop kernels.
 *
 * The reference (intellerce/controlanimate) has NO native/FFI boundary: its hot path is
 * Python calling torch/cuDNN/cuBLAS/xformers ops.  This header is therefore the boundary
 * a maintainer would bind UNDER the reference's Python modules (see INTEGRATION.md for the
 * ctypes stub).  Each entry point names the reference call site(s) it replaces.
 *
 * Conventions:
 *   - plain C, extern "C"; plain pointers and sizes; no torch / C++ types.
 *   - all pointers are DEVICE pointers on the current HIP device; the caller owns every
 *     buffer (kernels never allocate or free; scratch is passed in).
 *   - every call is asynchronous on the `stream` argument (a hipStream_t passed as void*).
 *   - returns CA_OK (0) or a negative CA_ERR_* code; never throws, never exits.
 *     ca_last_error() returns a thread-local message for the last failing call.
 *   - activations are channels-last: [images, H, W, C] (images = b*f in (b f) order),
 *     element type `dtype` = CA_BF16 or CA_F16; accumulation is always fp32.
 *   - "rows" means pixels/tokens: rows = images*H*W.
 */
#ifndef CONTROLANIMATE_HIP_H
#define CONTROLANIMATE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CA_ABI_VERSION 13

/* element types */
#define CA_BF16 0
#define CA_F16 1

/* error codes */
#define CA_OK 0
#define CA_ERR_INVALID_ARG (-1)
#define CA_ERR_UNSUPPORTED (-2)
#define CA_ERR_LAUNCH (-3)

/* epilogue activations */
#define CA_ACT_NONE 0
#define CA_ACT_SILU 1
#define CA_ACT_QUICK_GELU 2 /* x * sigmoid(1.702 x): CLIP ViT-L text/vision MLPs */
#define CA_ACT_GELU 3       /* erf GELU: CLIP ViT-H vision MLP (IP-Adapter image encoder) */

int ca_abi_version(void);
const char* ca_last_error(void);

/* ------------------------------------------------------------------------------------
 * ca_gemm: C[M,N] = epilogue( A[M,K] * W[N,K]^T )        (MFMA 16x16x32, fp32 accumulate)
 *
 * Replaces: every nn.Linear / 1x1 conv on the path --
 *   diffusers Attention.to_q/to_k/to_v/to_out (animatediff/models/attention.py:206-285 via
 *   modules/attention_processor.py:233-262), GEGLU.proj + ff.net[2]
 *   (animatediff/models/attention.py:303-357), Transformer3DModel.proj_in/proj_out
 *   (animatediff/models/attention.py:90,118), motion-module proj_in/proj_out
 *   (animatediff/models/motion_module.py:112,134), ResnetBlock3D.time_emb_proj and
 *   conv_shortcut (animatediff/models/resnet.py:163,186), TimestepEmbedding
 *   (animatediff/models/unet.py:526-534), ControlNet zero-convs.
 *
 * A may be the channel-concatenation of two sources (a2 != NULL): columns [0,k1) come from
 * `a` and [k1,k1+k2) from `a2` (replaces torch.cat at animatediff/models/unet_blocks.py:636,742
 * for the shortcut conv).  K = k1 + k2.
 *
 * epilogue, per element (m,n), v = acc:
 *   v += bias[n]                      (bias fp32 or NULL)
 *   v += rowbias[(m / rows_per_group) * ld_rowbias + n]   (fp32 or NULL; time-embedding add,
 *                                      animatediff/models/resnet.py:199-200)
 *   v *= alpha
 *   v += residual[m*ld_res + n]       (dtype or NULL; residual adds attention.py:273,285,288;
 *                                      motion_module.py:218-222; resnet.py:216)
 *   v *= post_scale                   (1/output_scale_factor, resnet.py:216)
 *   v = act(v)
 *   if geglu: weight rows are interleaved (h0,g0,h1,g1,...); out[m, n/2] = h * gelu_erf(g)
 *             (diffusers GEGLU); C then has N/2 columns.
 *   out_f32: store fp32 instead of dtype.
 * Requirements: K1, K2, N multiples of 8 (N multiple of 4 allowed when N < 8), lda/ldc/ld_res
 * multiples of 8 (dtype) so that 16-byte accesses are aligned.
 * ------------------------------------------------------------------------------------ */
typedef struct ca_gemm_args {
  const void* a;        /* [M, k1] rows at stride lda            */
  const void* a2;       /* [M, k2] rows at stride lda2, or NULL  */
  const void* w;        /* [N, K] row-major (PyTorch Linear.weight layout), dtype */
  void* c;              /* [M, N or N/2]                          */
  const float* bias;    /* [N] fp32 or NULL                       */
  const float* rowbias; /* [ceil(M/rows_per_group), ld_rowbias] fp32 or NULL */
  const void* residual; /* [M, N] dtype or NULL                   */
  int64_t lda, lda2, ldc, ld_res, ld_rowbias;
  int32_t m, n, k1, k2;
  int32_t rows_per_group;
  float alpha, post_scale;
  int32_t act;          /* CA_ACT_*      */
  int32_t geglu;        /* 0/1           */
  int32_t out_f32;      /* 0/1           */
  int32_t dtype;        /* CA_BF16/CA_F16 */
  /* LayerNorm folded into the GEMM (nn.LayerNorm + the Linear it feeds: animatediff/models/attention.py:
   * 214-237,263-288 norm1/2/3, motion_module.py:203-224 norms / ff_norm):  with W' = W diag(gamma) packed as
   * `w` and bias' = W beta + b as `bias`,
   *     LN(x) W^T + b = rstd * (x W'^T - mean * colsum(W')) + bias'
   * `ln_stats` [M][2] = (mean, rstd) per A row (ca_layernorm with `stats` set), `ln_colsum` [N] = the row sums
   * of W' as packed (both fp32, both or neither).  Applied to the accumulator before bias/rowbias.  A is then the
   * un-normalised tensor: the normalised copy is never written or re-read. */
  const float* ln_stats;
  const float* ln_colsum;
  /* ABI v6: `ln_colsum` set and `ln_stats` NULL = the kernel computes (mean, rstd = 1/sqrt(var + ln_eps)) of the A rows
   * itself while it streams them -- the separate statistics pass over A (ca_layernorm `stats`) disappears.  Available
   * only where ca_gemm_ln_inline_supported(args) returns 1 (the weight-resident K = 320 kernel of the 64x64-latent
   * level); elsewhere ca_gemm returns CA_ERR_ARG and the caller passes ln_stats. */
  float ln_eps;
  /* ABI v6: optional split-K scratch (device memory, caller-owned, may be shared by every call on one stream) of at
   * least ca_gemm_workspace_bytes(args) bytes, or NULL / too small: the GEMM then runs unsplit -- same results up to
   * fp32 summation order, only slower on grids that under-fill the chip (M = 2048: the 8x8-latent level). */
  void* workspace;
  int64_t workspace_bytes;
  /* ABI v6: LayerNorm statistics handed from the producing GEMM to the consuming one, without a pass over the tensor.
   *   producer: `row_sums_out` [M][P][2] fp32, P = ca_gemm_row_sums_parts(args) = N / 320 > 0: the epilogue also writes
   *             (sum, sum of squares) of every STORED output row per 320-column tile (the 128x320-tile kernel only;
   *             0 from the query = not available for this launch, do not set the pointer);
   *   consumer: `ln_parts` = P and `ln_stats` = that array instead of (mean, rstd): the epilogue adds the P partial
   *             sums in order and finishes mean / rstd = 1 / sqrt(var + ln_eps) itself (K of the consumer = the row width). */
  float* row_sums_out;
  int32_t ln_parts;
  /* ABI v9: `w` once more in MFMA-fragment order (written by ca_pack_w_frag(w, n, k, geglu, ...) with the SAME geglu flag as
   * this call), or NULL.  With it the K = 320 projections of the 64x64-latent level (M >= 16384, N % 320 == 0, N >= 960, one A source,
   * alpha = post_scale = 1, no activation: q|k|v, the GEGLU projection -- animatediff/models/attention.py:214-237,303-357,
   * motion_module.py:203-224) run on the activation-resident kernel, whose waves read W fragments straight from L2; `w` is
   * still what every other plan reads.  With ln_colsum set and ln_stats NULL that kernel leaves its (mean, rstd) in
   * `workspace` (ca_gemm_workspace_bytes = 8 M); without the workspace the launch keeps the weight-resident kernel. */
  const void* w_frag;
} ca_gemm_args;
int ca_gemm(const ca_gemm_args* args, void* stream);
/* ABI v9: dst[n * k] = the fragment-ordered copy of w[n, k] (k = 320, n % 64 == 0, 16-bit elements, 16-byte aligned) that
 * ca_gemm_args.w_frag takes; geglu = the flag of the ca_gemm calls that will use it (the weight-row interleave behind the
 * 16-byte stores differs).  Weights are packed once per model, so this runs at load time. */
int ca_pack_w_frag(const void* w, int32_t n, int32_t k, int32_t geglu, void* dst, void* stream);
/* ------------------------------------------------------------------------------------
 * ca_ff_fused (ABI v9): the whole feed-forward of a transformer block in one launch,
 *     y = GEGLU(LN(x) W1^T + b1) W2^T + b2 + residual
 * Replaces: BasicTransformerBlock / TemporalTransformerBlock feed-forward (animatediff/models/attention.py:288,303-357:
 *   `ff(norm3(hidden)) + hidden`; motion_module.py:203-224) at the 64x64-latent level -- LayerNorm fold, GEGLU projection and
 *   output projection as ca_gemm computes them (same rounding of the projection output and of h to the activation type), but
 *   the [M, 1280] intermediate stays in LDS.  Available for c = 320, inner = 1280, M >= 16384 (ca_ff_fused_supported; no
 *   launch, no device access); everything else runs as two ca_gemm calls.
 *   w1_frag: the LayerNorm-folded GEGLU weight [2560, 320] (rows value / gate interleaved) in the order of ca_pack_w_frag(geglu = 1);
 *   bias1 / colsum1 [2560] fp32 as ca_gemm's bias / ln_colsum; ln_stats [M][2] (mean, rstd) or NULL = computed in the kernel
 *   (ln_eps > 0); w2_frag: the output weight [320, 1280] in the order of ca_pack_w2_frag; bias2 [320] fp32 or NULL;
 *   residual [M, 320] or NULL. */
typedef struct ca_ff_args {
  const void* x;        /* [M, c] rows at stride lda */
  const void* w1_frag;
  const float* bias1;
  const float* colsum1;
  const float* ln_stats;
  const void* w2_frag;
  const float* bias2;
  const void* residual; /* [M, c] rows at stride ld_res, or NULL */
  void* y;              /* [M, c] rows at stride ldc */
  int64_t lda, ldc, ld_res;
  int32_t m, c, inner;
  float ln_eps;
  int32_t dtype;
  /* ABI v12: the transformer's output projection behind its last feed-forward, in the same launch
   * (animatediff/models/attention.py:163-175 `proj_out(hidden) + residual`; motion_module.py:158-163):  `y` then receives
   *   y_out = (feed-forward output) Wout^T + bias_out + residual_out.
   * w_out_frag: CA_ATTN_WOUT_FRAG_ELEMS elements written by ca_pack_w_out(proj_out.weight [c, c]); NULL = no output stage (as ABI v9).
   * Needs m % 128 == 0.  bias_out [c] fp32 or NULL; residual_out rows at stride ld_res_out (the transformer's input) or NULL. */
  const void* w_out_frag;
  const float* bias_out;
  const void* residual_out;
  int64_t ld_res_out;
} ca_ff_args;
int ca_ff_fused(const ca_ff_args* args, void* stream);
int ca_ff_fused_supported(const ca_ff_args* args);
/* dst[320 * 1280] = w[320, 1280] in the fragment order ca_ff_args.w2_frag takes (16-bit elements, 16-byte aligned). */
int ca_pack_w2_frag(const void* w, int32_t n, int32_t k, void* dst, void* stream);
/* ABI v10: the motion module's temporal self-attention of the 64x64-latent level up to (not including) its output projection,
 *   o = softmax(q k^T * scale) v per (pixel, head) over the frames,  q | k | v = (LayerNorm(x) + pe[frame]) Wqkv^T,
 * in one launch (animatediff/models/motion_module.py:251-331: norm, pos_encoder, to_q / to_k / to_v, attention with the frames
 * as the sequence).  Rows of x and o are in (batch, frame, token) order.  Takes c = 320, 8 heads, 16 frames (ABI v12: also 8 and 32
 * frames when the output stage is used), tokens a multiple of 128 / frames, >= 16384 rows (ca_tattn_fused_supported: no launch, no device access); everything else runs as ca_gemm + ca_attention.
 *   w_frag: CA_TATTN_W_FRAG_ELEMS 16-bit elements in the order of ca_pack_w_tattn (the UNFOLDED Wq | Wk | Wv);
 *   gamma [c] fp32: the LayerNorm weight; bias_pe [frames][ld_bias_pe] fp32: LayerNorm bias + positional encoding of the frame. */
#define CA_TATTN_W_FRAG_ELEMS 368640
typedef struct ca_tattn_args {
  const void* x;        /* [batch * frames * tokens, c] rows at stride lda */
  const void* w_frag;
  const float* gamma;
  const float* bias_pe;
  void* o;              /* [batch * frames * tokens, c] rows at stride ldo */
  int64_t lda, ldo, ld_bias_pe;
  int32_t batch, frames, tokens, heads, c;
  float ln_eps, scale;  /* scale = head_dim ** -0.5 */
  int32_t dtype;
  /* ABI v12: the attention's output projection, bias and residual in the SAME launch (motion_module.py:212-224,
   * `attention_block(norm(x)) + x`; modules/attention_processor.py:258-270 to_out[0]):  `o` then receives
   *   y = o Wout^T + bias_out + residual        instead of the attention output.
   * w_out_frag: CA_ATTN_WOUT_FRAG_ELEMS elements written by ca_pack_w_out(Wout [c, c]); NULL = no output stage (as ABI v10).
   * bias_out [c] fp32 or NULL; residual rows at stride ld_res (same row order as x / o) or NULL. */
  const void* w_out_frag;
  const float* bias_out;
  const void* residual;
  int64_t ld_res;
} ca_tattn_args;
int ca_tattn_fused(const ca_tattn_args* args, void* stream);
int ca_tattn_fused_supported(const ca_tattn_args* args);
/* ABI v12: dst[CA_ATTN_WOUT_FRAG_ELEMS] = w[320, 320] (an attention's to_out[0].weight) in the fragment order the output stage of
 * ca_tattn_fused / ca_xattn_fused streams (ca_tattn_args.w_out_frag); 16-bit elements, 16-byte aligned. */
#define CA_ATTN_WOUT_FRAG_ELEMS 102400
int ca_pack_w_out(const void* w, int32_t n, int32_t k, void* dst, void* stream);
/* dst[CA_TATTN_W_FRAG_ELEMS] = w[960, 320] (rows Wq, Wk, Wv) in the fragment order ca_tattn_args.w_frag takes (head dim 40
 * padded to 48 with zero rows; 16-bit elements, 16-byte aligned). */
int ca_pack_w_tattn(const void* w, int32_t n, int32_t k, void* dst, void* stream);
/* ABI v11: the text cross-attention of the 64x64-latent level up to (not including) its output projection,
 *   o = softmax(q K^T * scale) V per (row, head),  q = LayerNorm(x) Wq^T + bias,  K / V = the projected text tokens of the row's batch element,
 * in one launch (animatediff/models/attention.py:253-262: norm2, attn2.to_q, attention over encoder_hidden_states).  Row r belongs to
 * image r / tokens; image z uses text batch (z / frames_per_kv) % kv_mod.  Takes c = 320, 8 heads of 40, 65..80 keys, tokens % 128 == 0,
 * m >= 16384 and a multiple of tokens (ca_xattn_fused_supported: no launch, no device access); everything else runs as ca_gemm + ca_attention.
 *   wq_frag: CA_XATTN_W_FRAG_ELEMS 16-bit elements written by ca_xattn_pack_w from the LayerNorm-FOLDED Wq (Wq diag(gamma));
 *   bias [c] fp32 (Wq beta + b) or NULL;  kv_frag: kv_batches * 8 * CA_XATTN_KV_FRAG_ELEMS elements written by ca_xattn_pack_kv
 *   (K pre-multiplied by scale * log2 e, once per window and layer). */
#define CA_XATTN_W_FRAG_ELEMS 122880
#define CA_XATTN_KV_FRAG_ELEMS 7680
typedef struct ca_xattn_args {
  const void* x;        /* [m, c] rows at stride lda */
  const void* wq_frag;
  const float* bias;
  const void* kv_frag;
  void* o;              /* [m, c] rows at stride ldo */
  int64_t lda, ldo;
  int32_t m, tokens, frames_per_kv, kv_mod, kv_batches, nk, heads, c;
  float ln_eps;
  int32_t dtype;
  /* ABI v12: output projection + bias + residual in the same launch, as ca_tattn_args (animatediff/models/attention.py:253-262
   * `attn2(norm2(hidden)) + hidden`).  w_out_frag NULL = no output stage (as ABI v11). */
  const void* w_out_frag;
  const float* bias_out;
  const void* residual;
  int64_t ld_res;
  /* ABI v13: the IP-Adapter's image-prompt tokens in the same launch (modules/attention_processor.py:433-477, IPAttnProcessor2_0:
   * `hidden = attention(q, K, V) + scale * attention(q, K_ip, V_ip)` before to_out).  kv_frag_ip: a second ca_xattn_pack_kv buffer
   * (same kv_batches; packed from the to_k_ip | to_v_ip projection of the last nk_ip context rows, nk_ip 1..16); needs w_out_frag.
   * NULL / 0 = no image-prompt tokens (as ABI v12). */
  const void* kv_frag_ip;
  int32_t nk_ip;
  float ip_scale;
} ca_xattn_args;
int ca_xattn_fused(const ca_xattn_args* args, void* stream);
int ca_xattn_fused_supported(const ca_xattn_args* args);
/* dst[CA_XATTN_W_FRAG_ELEMS] = w[320, 320] in the per-head fragment order ca_xattn_args.wq_frag takes (head dim 40 padded to 48 with zero rows). */
int ca_xattn_pack_w(const void* w, int32_t n, int32_t k, void* dst, void* stream);
/* dst[kv_batches * 8 * CA_XATTN_KV_FRAG_ELEMS] = the K (columns 0..319, * scale * log2 e) and V (columns 320..639) rows
 * row_offset .. row_offset + nk of every batch of kv [kv_batches * rows_per_batch, ld] as MFMA fragments in lane order; nk 65..80 (the text keys) or, ABI v13,
 * 1..16 (the image-prompt set: ca_xattn_args.kv_frag_ip). */
int ca_xattn_pack_kv(const void* kv, int64_t ld, int32_t kv_batches, int32_t rows_per_batch, int32_t row_offset, int32_t nk, float scale, int32_t dtype,
                     void* dst, void* stream);
/* bytes of split-K scratch this launch can use (0: it would not split) */
int64_t ca_gemm_workspace_bytes(const ca_gemm_args* args);
/* partial sums per row this launch can leave in row_sums_out (0: it cannot).  N / 320 on the 128 x 320-tile kernels; ABI v8:
 * 4 * N / 320 on the 256 x 320 kernel (one per 80-column wave quarter) -- a consumer's ln_parts takes at most 4, so more are
 * finished with ca_ln_finish_sums first. */
int ca_gemm_row_sums_parts(const ca_gemm_args* args);
/* 1 if ca_gemm can take these args with ln_stats == NULL (fields other than the pointers' values are what matters;
 * no launch, no device access). */
int ca_gemm_ln_inline_supported(const ca_gemm_args* args);
/* ABI v8: 1 if a consumer launch with partial sums (ln_parts > 0) should rather get finished statistics: the kernel the plan
 * prefers for the shape (256 x 320 tiles) reads (mean, rstd) only.  The caller then runs ca_ln_finish_sums and passes
 * ln_stats = its output, ln_parts = 0.  No launch, no device access. */
int ca_gemm_wants_finished_stats(const ca_gemm_args* args);
/* ABI v8: (mean, rstd = 1 / sqrt(var + eps)) [rows][2] from `sums` [rows][parts][2] = (sum, sum of squares) per row and
 * 320-column tile as left by a producing GEMM (row_sums_out); k = the row width.  Same arithmetic and order as the consuming
 * epilogues that finish the sums themselves (reference: nn.LayerNorm statistics, animatediff/models/attention.py:214-237). */
int ca_ln_finish_sums(const float* sums, int parts, int64_t rows, int k, float eps, float* mean_rstd, void* stream);
/* ABI v7: the label of the kernel instantiation ca_gemm would launch for these arguments ("wres160", "pp128x320",
 * "ps128x320", "128x128", "128x160", "128x64_db", "128x128_splitk6", "reg_128x64", ...), written NUL-terminated into
 * buf[len].  No launch, no device access: the choice is a pure function of the sizes, strides, flags and which pointers
 * are set (the product library has no tuning environment variables).  Returns CA_OK or the error ca_gemm would return. */
int ca_gemm_plan_name(const ca_gemm_args* args, char* buf, int32_t len);

/* ------------------------------------------------------------------------------------
 * ca_conv3x3: NHWC 3x3 convolution, padding 1, stride 1 or 2, as an implicit GEMM
 *   (M = images*Hout*Wout, N = Cout, K = 9*Cin) on the same MFMA core as ca_gemm.
 *
 * Replaces: InflatedConv3d (animatediff/models/resnet.py:12-20) at conv_in/conv_out
 *   (unet.py:140,319), ResnetBlock3D.conv1/conv2 (resnet.py:153,173), Downsample3D
 *   (resnet.py:96, stride 2), Upsample3D (resnet.py:47,67: `upsample`=1 folds the nearest x2
 *   F.interpolate into the input gather), ControlNet convs and hint embedding.
 * Input may be the channel concat of two NHWC sources (x2 != NULL).
 * Weight layout: [Cout][kh][kw][Cin] (packed once at load from PyTorch's [Cout][Cin][kh][kw]).
 * Epilogue fields as in ca_gemm (rowbias = time embedding per batch element b:
 *   rows_per_group = f*Hout*Wout; residual = shortcut input).
 * Hin/Win are the dimensions of the STORED input; with upsample=1 the logical input is
 * 2*Hin x 2*Win.  Cin1, Cin2, Cout multiples of 8 (Cout multiple of 4 when < 8).
 * ------------------------------------------------------------------------------------ */
typedef struct ca_conv_args {
  const void* x;        /* [images, Hin, Win, cin1] */
  const void* x2;       /* [images, Hin, Win, cin2] or NULL */
  const void* w;        /* [cout, 3, 3, cin1+cin2] */
  void* y;              /* [images, Hout, Wout, cout] */
  const float* bias;
  const float* rowbias;
  const void* residual; /* [images*Hout*Wout, ld_res] */
  int64_t ld_res, ld_rowbias;
  int32_t images, hin, win, cin1, cin2, cout;
  int32_t stride;       /* 1 or 2 */
  int32_t upsample;     /* 0 or 1 (nearest x2 before the conv) */
  int32_t rows_per_group;
  float alpha, post_scale;
  int32_t act;
  int32_t out_f32;
  int32_t dtype;
  /* Scratch for the split-K schedule of small-M convolutions (8x8 / 16x16 latent levels, where
   * M/128 x Cout/128 tiles cannot fill 256 CUs).  Caller-owned device memory of at least
   * ca_conv3x3_workspace_bytes(args) bytes, or NULL / too small: the convolution then runs
   * unsplit.  Results are deterministic either way (slabs are added in a fixed order). */
  void* workspace;
  int64_t workspace_bytes;
  /* 0: zero padding 1 on every side (Conv2d padding=1).  1: no padding before, one row/column after
   * (diffusers Downsample2D(padding=0): F.pad(x, (0,1,0,1)) + Conv2d(stride=2), the VAE encoder's
   * downsamplers): Hout = (H + 1 - 3) / stride + 1. */
  int32_t pad_asym;
  /* ABI v12: the weight once more in Winograd form, U [16][cout][cin1 + cin2] = G g G^T per (cout, cin) (written by ca_pack_w_wino
   * from `w`), or NULL.  With it -- and a workspace of ca_conv3x3_workspace_bytes(args) bytes -- the deep convolutions of the small-latent
   * levels (stride 1, padding 1, with or without `upsample`, even output H and W, cin >= 1280 (>= 640 at up to 4096 tiles), cout % 320 == 0,
   * images * Hout * Wout / 4 a multiple of 256 and at most 16384 tiles) run as F(2x2, 3x3): an input transform, ONE launch of the 256 x 320 GEMM kernel over the sixteen transformed GEMMs,
   * an output transform that applies the epilogue.  2.25 x fewer multiply-adds; results differ from the direct form by fp16 rounding
   * of the transformed operands (tests/test_kernels_gpu.py::test_conv3x3_winograd).  NULL / no workspace / another shape: the direct form. */
  const void* w_wino;
  /* ABI v12: 1 = `x` IS the transformed input V [16][tiles][cin1] already (written by ca_groupnorm's wino_v; cin2 = 0): the Winograd
   * route starts at its GEMM.  The call fails unless the route is taken (w_wino, workspace, shape); the workspace then holds M only
   * (ca_conv3x3_workspace_bytes says how much). */
  int32_t x_is_wino_v;
} ca_conv_args;
/* ABI v12: dst[16 * cout * cin] = G g G^T of w [cout][3][3][cin] (the layout of ca_conv_args.w) in the element type `dtype`. */
int ca_pack_w_wino(const void* w, int32_t cout, int32_t cin, int32_t dtype, void* dst, void* stream);
int64_t ca_conv3x3_workspace_bytes(const ca_conv_args* args);
int ca_conv3x3(const ca_conv_args* args, void* stream);
/* ABI v7: as ca_gemm_plan_name, for ca_conv3x3 */
int ca_conv3x3_plan_name(const ca_conv_args* args, char* buf, int32_t len);

/* ------------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU), NHWC, two launches: statistics then apply.
 *
 * Replaces: InflatedGroupNorm / nn.GroupNorm + SiLU (animatediff/models/resnet.py:23-31,
 *   148-151,169-170,191-192,202,208; unet.py:316-317,614-615), GroupNorm eps=1e-6 without
 *   activation (animatediff/models/attention.py:86,130; motion_module.py:111,144).
 * `frames_per_stat` = 1 gives per-frame statistics (InflatedGroupNorm / v2); = f gives the
 * cross-frame statistics of plain nn.GroupNorm on the 5-D tensor (v1 motion module configs,
 * SURVEY App. C-5).  Input may be a channel concat of two sources.
 *
 * ca_groupnorm_stats writes partial (sum, sumsq) per (stat group, row chunk, norm group)
 * into `partials` (fp32, ca_groupnorm_partials_floats() elements); ca_groupnorm_apply
 * reduces the partials deterministically (fixed order, fp64 combine) and normalises.
 * ------------------------------------------------------------------------------------ */
typedef struct ca_groupnorm_args {
  const void* x;        /* [images, hw, c1] */
  const void* x2;       /* [images, hw, c2] or NULL */
  void* y;              /* [images, hw, c1+c2] */
  const float* gamma;   /* [c1+c2] */
  const float* beta;    /* [c1+c2] */
  float* partials;      /* scratch, see ca_groupnorm_partials_floats */
  int32_t images, hw, c1, c2;
  int32_t groups;       /* 32 */
  int32_t frames_per_stat;
  float eps;
  int32_t act;          /* CA_ACT_NONE / CA_ACT_SILU */
  int32_t dtype;
  /* ABI v12 (ca_groupnorm only): with wino_v the launch writes, INSTEAD of y, the Winograd F(2x2, 3x3) input transform of the
   * normalised (+ activated) tensor -- V [16][images * (wino_h / 2) * (wino_w / 2)][c1 + c2], what ca_conv3x3's Winograd route computes
   * from y as its first stage -- so that the convolution behind it (ca_conv_args.x_is_wino_v) starts at its GEMM and y never exists
   * (ResnetBlock3D: norm -> nonlinearity -> conv, animatediff/models/resnet.py:188-212).  hw = wino_h * wino_w.  Only where
   * ca_groupnorm_wino_supported(args) says 1; y and partials are unused then. */
  void* wino_v;
  int32_t wino_h, wino_w;
} ca_groupnorm_args;
int ca_groupnorm_wino_supported(const ca_groupnorm_args* args);
int64_t ca_groupnorm_partials_floats(int32_t images, int32_t hw, int32_t frames_per_stat, int32_t groups);
int ca_groupnorm_stats(const ca_groupnorm_args* args, void* stream);
int ca_groupnorm_apply(const ca_groupnorm_args* args, void* stream);
/* ABI v6: both passes in one call; small statistics groups (8x8 / 16x16 latents) run as ONE launch that keeps the
 * group in registers between the passes.  `partials` as above (unused by the fused kernel, still required). */
int ca_groupnorm(const ca_groupnorm_args* args, void* stream);

/* ------------------------------------------------------------------------------------
 * ca_layernorm: y[r,:] = LN(x[r,:]) * gamma + beta (+ pos[(r / rows_per_frame) % frames, :])
 *
 * Replaces: nn.LayerNorm in BasicTransformerBlock (animatediff/models/attention.py:214,231,237)
 *   and TemporalTransformerBlock (motion_module.py:203,209); the optional `pos` table fuses
 *   PositionalEncoding.forward (motion_module.py:243-245: x + pe[:, :f]) -- rows are in
 *   (b f n) order so the frame of row r is (r / rows_per_frame) % frames.
 * eps = 1e-5 (torch default). C multiple of 8, C <= 2048.
 * ------------------------------------------------------------------------------------ */
typedef struct ca_layernorm_args {
  const void* x; void* y;
  const float* gamma; const float* beta;
  const float* pos;     /* [frames, c] fp32 or NULL */
  int64_t rows;
  int32_t c;
  int32_t rows_per_frame, frames;
  float eps;
  int32_t dtype;
  float* stats;         /* non-NULL: write (mean, rstd) per row to stats[2*row ..] and nothing else
                           (y, gamma, beta, pos unused): input of ca_gemm_args.ln_stats */
} ca_layernorm_args;
int ca_layernorm(const ca_layernorm_args* args, void* stream);

/* ------------------------------------------------------------------------------------
 * ca_attention: O = softmax(Q K^T * scale) V per (batch z, head), flash-style online softmax,
 *   MFMA for QK^T and PV.  One kernel serves:
 *   - spatial self-attention  (AttnProcessor2_0, modules/attention_processor.py:200-272;
 *     called from animatediff/models/attention.py:271)
 *   - cross-attention with per-b text K/V shared by the f frames of b (kv_div = f; replaces the
 *     `repeat 'b n c -> (b f) n c'` at animatediff/models/attention.py:125) incl. the
 *     ControlNet variant that drops the last 4 tokens (CNAttnProcessor2_0, :608-609: pass nk-4)
 *   - IP-Adapter's second attention over the 4 image tokens, accumulated into O
 *     (IPAttnProcessor2_0, modules/attention_processor.py:461-477): accumulate=1, out_scale=scale
 *   - temporal self-attention over frames (VersatileAttention, animatediff/models/
 *     motion_module.py:272-329): the `(b f) d c -> (b d) f c` transposes are replaced by
 *     strided addressing (row_stride = tokens*ld, inner = tokens).
 *
 * Addressing (element offsets): for batch z (0 <= z < batches):
 *   zo = z / inner_count, zi = z % inner_count
 *   Q row i  at q + zo*q_outer + zi*q_inner + i*q_row + head*head_dim   (same scheme for O)
 *   z' = (z / kv_div) % kv_mod; K row j at k + (z'/kv_inner_count)*k_outer + (z'%kv_inner_count)*k_inner
 *                                   + j*k_row + head*head_dim   (same for V with v pointer)
 * kv_mod reproduces the reference's ControlNet prompt tiling torch.cat([embeds]*frame_count)
 * (modules/controlresiduals_pipeline.py:292: image z reads embeds[z % b]); pass kv_mod = batches when unused.
 * head_dim multiple of 8, <= 160.
 * ------------------------------------------------------------------------------------ */
typedef struct ca_attn_args {
  const void* q; const void* k; const void* v; void* o;
  int64_t q_outer, q_inner, q_row;
  int64_t o_outer, o_inner, o_row;
  int64_t k_outer, k_inner, k_row;   /* shared by K and V */
  int32_t inner_count, kv_inner_count, kv_div, kv_mod;
  int32_t batches, heads, head_dim;
  int32_t nq, nk;
  float scale;          /* softmax scale (head_dim^-0.5) */
  float out_scale;      /* multiplies the attention output */
  int32_t accumulate;   /* 1: O += out_scale * attn */
  int32_t dtype;
  int32_t causal;       /* 1: key j visible to query i only if j <= i (nq == nk).  CLIP text encoder
                           (transformers CLIPTextModel's causal mask; called through Compel at
                           modules/controlanimate_pipeline.py:133-135) */
  /* ABI v7: optional key mask, one byte per (batch z, key j) at key_mask[z * key_mask_stride + j]; 0 = the key is
   * invisible to every query of that batch element (its score is -inf before the softmax), combined with `causal`.
   * transformers' CLIPTextModel `attention_mask` ([B, L] -> additive [B,1,1,L]): what Compel 2.0.2's default
   * DownweightMode.MASK passes for a down-weighted fragment such as `(muscle body)0.2`
   * (modules/controlanimate_pipeline.py:133-135, configs/prompts/SampleConfig.yaml:16).  NULL = no mask.  A query whose
   * keys are ALL masked gets zeros. */
  const uint8_t* key_mask;
  int64_t key_mask_stride;
} ca_attn_args;
int ca_attention(const ca_attn_args* args, void* stream);
/* ABI v12: the label of the kernel ca_attention runs for these arguments ("attn_dma40" the LDS-DMA kernel at head_dim 40,
 * "attn_dma80", "attn_dma_fold" / "attn_dma_sr" / "attn_dma", "attn_short" the register-resident text cross-attention,
 * "attn_tiny16" / "attn_tiny32" the one-wave temporal forms, "attn_generic").  No launch, no device access: the parity tests
 * assert with it that the headline shapes ran on the kernels DESIGN.md names (tests/test_fullsize_gpu.py, test_dispatch_plan.py). */
int ca_attention_plan_name(const ca_attn_args* args, char* buf, int32_t len);

/* ------------------------------------------------------------------------------------
 * Small elementwise kernels of the loop.
 * ------------------------------------------------------------------------------------ */

/* out[i] = a[i] + b[i % b_period]  (dtype).  ControlNet residual adds incl. the broadcast of
 * b=1 residuals over the CFG batch (animatediff/models/unet.py:567-576,584-585). n multiple of 8. */
int ca_add_bcast(const void* a, const void* b, void* out, int64_t n, int64_t b_period,
                 int32_t dtype, void* stream);
/* ABI v8: dst = `times` copies of the `bytes` of src behind each other (one read of src): torch.cat([x] * times) along the
 * leading dimension -- the reference's torch.cat([latents] * 2) (animatediff/pipelines/controlanimation_pipeline.py:797) makes
 * the two CFG halves identical, the shared prefix is computed once and repeated here.  16-byte vector copies when bytes % 16 == 0
 * and both pointers are 16-byte aligned, a byte-granular kernel otherwise (any size, any alignment). */
int ca_repeat(const void* src, void* dst, int64_t bytes, int32_t times, void* stream);

/* y[r, :] = softmax(scale * x[r, :]) for an fp32 score matrix, written in `dtype`.
 * Replaces the softmax inside F.scaled_dot_product_attention for the VAE's mid-block attention
 * (third-party diffusers==0.23.0 AutoencoderKL: Attention(heads=1, dim_head=512) called at
 * animatediff/pipelines/controlanimation_pipeline.py:508 (decode) and :577,589 (encode)), whose
 * head_dim is beyond ca_attention's fused kernel: scores and P.V run through ca_gemm. cols % 4 == 0. */
int ca_softmax_rows(const float* x, void* y, int64_t rows, int32_t cols, int64_t ldx, int64_t ldy,
                    float scale, int32_t dtype, void* stream);

/* y = silu(x) on fp32 vectors (ResnetBlock3D: time_emb_proj(silu(temb)), resnet.py:196). */
int ca_silu_f32(const float* x, float* y, int64_t n, void* stream);

/* Timesteps(320, flip_sin_to_cos=True, freq_shift=0) (unet.py:526):
 * out[b, :] = [cos(t*f_i), sin(t*f_i)], f_i = exp(-ln(10000)*i/half); out dtype for the
 * following linear_1 GEMM. `t` per batch element in a device fp32 array or (t_dev NULL) the
 * scalar t_host. */
int ca_timestep_embedding(const float* t_dev, float t_host, void* out, int32_t batch,
                          int32_t dim, int32_t dtype, void* stream);

/* latents [b0, c, f, h, w] fp32 -> UNet input NHWC [rep*b0*f, h, w, cpad] dtype, multiplied by
 * in_scale (scheduler.scale_model_input) and duplicated `rep` times for CFG
 * (controlanimation_pipeline.py:797-800).  Channels c..cpad-1 are zero. */
int ca_latents_to_nhwc(const float* latents, void* out, int32_t b0, int32_t c, int32_t f,
                       int32_t h, int32_t w, int32_t cpad, int32_t rep, float in_scale,
                       int32_t dtype, void* stream);

/* UNet output NHWC [b*f, h, w, c] (dtype or fp32) -> [b, c, f, h, w] fp32 (API boundary). */
int ca_nhwc_to_ncfhw_f32(const void* x, float* out, int32_t b, int32_t c, int32_t f, int32_t h,
                         int32_t w, int32_t ldx, int32_t x_is_f32, int32_t dtype, void* stream);

/* generic layout converters at the module API boundary:
 * [b, c, f, h, w] (fp32 | fp16 | bf16, arbitrary strides given in elements) <-> NHWC dtype. */
int ca_ncfhw_to_nhwc(const void* x, int32_t x_kind /*0 f32,1 f16,2 bf16*/, const int64_t strides[5],
                     void* out, int32_t b, int32_t c, int32_t f, int32_t h, int32_t w, int32_t cpad,
                     int32_t dtype, void* stream);

/* Fused classifier-free-guidance combine + scheduler update
 * (controlanimation_pipeline.py:844-849 and the step functions :1520-1609 / diffusers
 * DDIM/LCM/Euler restated in oracle/schedulers.py).  eps comes NHWC fp32 [rep*f, h, w, ld_eps]
 * from conv_out; latents/noise/prev/denoised are [1, c, f, h, w] fp32.
 *   eps  = rep==2 ? e_u + g*(e_c - e_u) : e
 *   x0   = (x - coef[0]*eps) * coef[1];  if clip > 0: x0 = clamp(x0, -clip, clip)
 *   den  = coef[2]*x0 + coef[3]*x
 *   prev = coef[4]*den + coef[5]*eps + coef[6]*noise
 * `noise` / `denoised` may be NULL. */
int ca_cfg_scheduler_step(const float* eps, int32_t ld_eps, int32_t rep, float guidance,
                          const float* latents, const float* noise, float* prev, float* denoised,
                          int32_t c, int32_t f, int32_t h, int32_t w, const float coef[7],
                          float clip, void* stream);

/* out[i] = sum_k coef[k] * x[k][i], fp32, 1 <= n_terms <= 8 (out may alias any x[k]).
 * The update rule of the MULTISTEP / history-carrying samplers the reference's facade offers
 * (modules/controlanimate_pipeline.py:52-61: DPMSolverMultistep, LMSDiscrete, PNDM): each of their steps is a fixed
 * linear combination of the current sample, the current and earlier model outputs and (ancestral) noise, with
 * host-computed coefficients (controlanimate_amd/schedulers.py).  The CFG combine that precedes it is
 * ca_cfg_scheduler_step with coef = {0,1,0,0, 0,1,0} (prev := combined eps). */
int ca_lincomb(float* out, const float* const* x, const float* coef, int32_t n_terms, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CONTROLANIMATE_HIP_H */

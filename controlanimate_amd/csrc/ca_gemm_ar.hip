// Activation-resident K = 320 GEMM (ca_gemm_ar.h): separate translation unit (compile time).
#include "ca_gemm_core.h"
#include <atomic>

// CU count of the CURRENT device, cached per device id (a process may drive devices with different CU counts -- partition
// modes -- and first calls may race: the slots are written once each with the same value, atomically).
__attribute__((visibility("hidden"))) int ar_cu_count() {
  static std::atomic<int> cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int n = cache[dev].load(std::memory_order_relaxed);
  if (n <= 0) {
    hipDeviceProp_t prop;
    n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cache[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

namespace {
using namespace ca_gemm_detail;
#include "ca_gemm_ar.h"
#include "ca_attn_out.h"
#include "ca_ff_fused.h"
#include "ca_tattn_fused.h"
#include "ca_xattn_fused.h"

template <int DT>
int launch_ar(const GemmKParams& p, hipStream_t st) {
  const int tiles_m = (p.m + 127) / 128;
  const unsigned c_bytes = (unsigned)((((int64_t)p.m - 1) * p.ldc + (p.geglu ? p.n / 2 : p.n)) * 2);
  const unsigned res_bytes = p.res ? (unsigned)((((int64_t)p.m - 1) * p.ld_res + p.n) * 2) : 0u;
  const int slots = 2 * ar_cu_count();  // two blocks (80 KB of LDS, four waves each) per CU
  const unsigned grid = (unsigned)(tiles_m < slots ? tiles_m : slots);
  float* ws = p.ln_inline ? p.partial : nullptr;
  if (p.geglu) hipLaunchKernelGGL((k_gemm_ar<DT, 2>), dim3(grid), dim3(256), 0, st, p, p.wf, tiles_m, c_bytes, 0u, ws);
  else if (p.ln_colsum) hipLaunchKernelGGL((k_gemm_ar<DT, 1>), dim3(grid), dim3(256), 0, st, p, p.wf, tiles_m, c_bytes, 0u, ws);
  else hipLaunchKernelGGL((k_gemm_ar<DT, 0>), dim3(grid), dim3(256), 0, st, p, p.wf, tiles_m, c_bytes, res_bytes, ws);
  return CA_OK;
}
}  // namespace

int ca_launch_gemm_ar(const ca_gemm_detail::GemmKParams& p0, int dtype, hipStream_t st) {
  static const int dbg = CA_KNOB("CA_PP_DBG", 0);  // (timing experiments: 1 = no stores)
  ca_gemm_detail::GemmKParams p = p0;
  p.dbg = dbg;
  return dtype == CA_BF16 ? launch_ar<CA_BF16>(p, st) : launch_ar<CA_F16>(p, st);
}

extern "C" int ca_pack_w_frag(const void* w, int32_t n, int32_t k, int32_t geglu, void* dst, void* stream) {
  CA_REQUIRE(w && dst, "ca_pack_w_frag: null operand");
  CA_REQUIRE(k == 320 && n > 0 && n % 64 == 0, "ca_pack_w_frag: n=%d (multiple of 64) k=%d (320)", n, k);
  CA_REQUIRE((((uintptr_t)w | (uintptr_t)dst) & 15) == 0, "ca_pack_w_frag: operands must be 16-byte aligned");
  const int64_t pieces = (int64_t)n * 40;
  hipLaunchKernelGGL(k_pack_w_frag, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u16*)w, (u16*)dst, n, geglu);
  CA_CHECK_LAUNCH("ca_pack_w_frag");
  return CA_OK;
}

extern "C" int ca_pack_w2_frag(const void* w, int32_t n, int32_t k, void* dst, void* stream) {
  CA_REQUIRE(w && dst, "ca_pack_w2_frag: null operand");
  CA_REQUIRE(n == 320 && k == 1280, "ca_pack_w2_frag: n=%d k=%d (320 x 1280)", n, k);
  CA_REQUIRE((((uintptr_t)w | (uintptr_t)dst) & 15) == 0, "ca_pack_w2_frag: operands must be 16-byte aligned");
  hipLaunchKernelGGL(k_pack_w2_frag, dim3(200), dim3(256), 0, (hipStream_t)stream, (const u16*)w, (u16*)dst);
  CA_CHECK_LAUNCH("ca_pack_w2_frag");
  return CA_OK;
}

// the optional output stage of the one-launch attentions (ABI v12): 1 = absent or acceptable
static int attn_out_args_ok(const void* w_out_frag, const float* bias_out, const void* residual, int64_t ld_res, int64_t rows) {
  if (!w_out_frag) return (bias_out || residual) ? 0 : 1;  // bias / residual belong to the projection
  if ((((uintptr_t)w_out_frag | (uintptr_t)bias_out | (uintptr_t)residual) & 15) != 0) return 0;
  if (residual && (ld_res % 8 || ld_res < 320 || ((rows - 1) * ld_res + 320) * 2 >= 0x7FFFFF00ll)) return 0;
  return 1;
}

extern "C" int ca_ff_fused_supported(const ca_ff_args* a) {
  if (!a || !a->x || !a->w1_frag || !a->bias1 || !a->colsum1 || !a->w2_frag || !a->y) return 0;
  if (a->c != 320 || a->inner != 1280 || a->m < 16384) return 0;
  if (a->dtype != CA_BF16 && a->dtype != CA_F16) return 0;
  if (a->lda % 8 || a->ldc % 8 || (a->residual && a->ld_res % 8)) return 0;
  if (a->lda < 320 || a->ldc < 320 || (a->residual && a->ld_res < 320)) return 0;  // rows must not overlap
  if ((((uintptr_t)a->x | (uintptr_t)a->y | (uintptr_t)a->w1_frag | (uintptr_t)a->w2_frag | (uintptr_t)a->residual) & 15) != 0) return 0;
  // fp32 operands: the kernel reads ln_stats as float2 and the bias / column-sum tables as float4
  if (((uintptr_t)a->ln_stats & 7) != 0 || (((uintptr_t)a->bias1 | (uintptr_t)a->colsum1 | (uintptr_t)a->bias2) & 15) != 0) return 0;
  const int64_t lim = 0x7FFFFF00ll;
  if (((int64_t)(a->m - 1) * a->lda + 320) * 2 >= lim || ((int64_t)(a->m - 1) * a->ldc + 320) * 2 >= lim) return 0;
  if (a->residual && ((int64_t)(a->m - 1) * a->ld_res + 320) * 2 >= lim) return 0;
  if (!a->ln_stats && !(a->ln_eps > 0.f)) return 0;
  if (!attn_out_args_ok(a->w_out_frag, a->bias_out, a->residual_out, a->ld_res_out, a->m)) return 0;
  if (a->w_out_frag && a->m % 128) return 0;  // the output stage works on whole row tiles
  return 1;
}

extern "C" int ca_ff_fused(const ca_ff_args* a, void* stream) {
  CA_REQUIRE(a != nullptr, "ca_ff_fused: null args");
  CA_REQUIRE(ca_ff_fused_supported(a), "ca_ff_fused: arguments outside what the fused feed-forward takes (C = 320, inner 1280, M >= 16384, fragment-ordered weights, "
                                        "16-byte aligned operands, 32-bit byte offsets): ask ca_ff_fused_supported() first");
  FfParams p{};
  p.x = (const u16*)a->x;
  p.w1f = (const u16*)a->w1_frag;
  p.bias1 = a->bias1;
  p.cs1 = a->colsum1;
  p.w2f = (const u16*)a->w2_frag;
  p.bias2 = a->bias2;
  p.res = (const u16*)a->residual;
  p.y = (u16*)a->y;
  p.ln_stats = a->ln_stats;
  p.lda = a->lda, p.ldc = a->ldc, p.ld_res = a->ld_res;
  p.m = a->m;
  p.ln_eps = a->ln_eps;
  p.x_bytes = (unsigned)(((int64_t)(a->m - 1) * a->lda + 320) * 2);
  p.y_bytes = (unsigned)(((int64_t)(a->m - 1) * a->ldc + 320) * 2);
  p.res_bytes = a->residual ? (unsigned)(((int64_t)(a->m - 1) * a->ld_res + 320) * 2) : 0u;
  if (a->w_out_frag) {
    p.out.wof = (const u16*)a->w_out_frag;
    p.out.bias = a->bias_out;
    p.out.res = (const u16*)a->residual_out;
    p.out.ld_res = (int)a->ld_res_out;
    p.out.res_bytes = a->residual_out ? (unsigned)(((int64_t)(a->m - 1) * a->ld_res_out + 320) * 2) : 0u;
  }
  const int tiles_m = (a->m + 127) / 128;
  const unsigned grid = (unsigned)(tiles_m < ar_cu_count() ? tiles_m : ar_cu_count());
  if (a->w_out_frag) {
    if (a->dtype == CA_BF16) hipLaunchKernelGGL((k_ff_fused<CA_BF16, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, tiles_m);
    else hipLaunchKernelGGL((k_ff_fused<CA_F16, true>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, tiles_m);
  } else {
    if (a->dtype == CA_BF16) hipLaunchKernelGGL((k_ff_fused<CA_BF16, false>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, tiles_m);
    else hipLaunchKernelGGL((k_ff_fused<CA_F16, false>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p, tiles_m);
  }
  CA_CHECK_LAUNCH("ca_ff_fused");
  return CA_OK;
}

extern "C" int ca_pack_w_tattn(const void* w, int32_t n, int32_t k, void* dst, void* stream) {
  CA_REQUIRE(w && dst, "ca_pack_w_tattn: null operand");
  CA_REQUIRE(n == 960 && k == 320, "ca_pack_w_tattn: n=%d k=%d (Wq | Wk | Wv rows: 960 x 320)", n, k);
  CA_REQUIRE((((uintptr_t)w | (uintptr_t)dst) & 15) == 0, "ca_pack_w_tattn: operands must be 16-byte aligned");
  static_assert(CA_TATTN_WF_ELEMS == CA_TATTN_W_FRAG_ELEMS, "header constant");
  hipLaunchKernelGGL(k_pack_w_tattn, dim3((CA_TATTN_WF_ELEMS / 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const u16*)w, (u16*)dst);
  CA_CHECK_LAUNCH("ca_pack_w_tattn");
  return CA_OK;
}

extern "C" int ca_pack_w_out(const void* w, int32_t n, int32_t k, void* dst, void* stream) {
  CA_REQUIRE(w && dst, "ca_pack_w_out: null operand");
  CA_REQUIRE(n == 320 && k == 320, "ca_pack_w_out: n=%d k=%d (to_out[0].weight: 320 x 320)", n, k);
  CA_REQUIRE((((uintptr_t)w | (uintptr_t)dst) & 15) == 0, "ca_pack_w_out: operands must be 16-byte aligned");
  static_assert(CA_WOUT_ELEMS == CA_ATTN_WOUT_FRAG_ELEMS, "header constant");
  hipLaunchKernelGGL(k_pack_w_out, dim3((CA_WOUT_ELEMS / 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const u16*)w, (u16*)dst);
  CA_CHECK_LAUNCH("ca_pack_w_out");
  return CA_OK;
}

extern "C" int ca_tattn_fused_supported(const ca_tattn_args* a) {
  if (!a || !a->x || !a->w_frag || !a->gamma || !a->bias_pe || !a->o) return 0;
  if (a->c != 320 || a->heads != 8 || a->batch < 1) return 0;
  // 16 frames: both forms; 8 / 32 frames (ABI v12): the eight-wave kernel only, i.e. with the output stage
  if (a->frames != 16 && !((a->frames == 8 || a->frames == 32) && a->w_out_frag)) return 0;
  {
    const int px = 128 / a->frames;  // pixels per 128-row tile
    if (a->tokens < px || a->tokens % px) return 0;
  }
  if (a->dtype != CA_BF16 && a->dtype != CA_F16) return 0;
  const int64_t rows = (int64_t)a->batch * a->frames * a->tokens;
  if (rows < 16384) return 0;
  if (a->lda % 8 || a->ldo % 8 || a->lda < 320 || a->ldo < 320 || a->ld_bias_pe % 4 || a->ld_bias_pe < 320) return 0;
  if ((((uintptr_t)a->x | (uintptr_t)a->o | (uintptr_t)a->w_frag | (uintptr_t)a->gamma | (uintptr_t)a->bias_pe) & 15) != 0) return 0;
  const int64_t lim = 0x7FFFFF00ll;
  if (((rows - 1) * a->lda + 320) * 2 >= lim || ((rows - 1) * a->ldo + 320) * 2 >= lim) return 0;
  if (!(a->ln_eps > 0.f) || !(a->scale > 0.f)) return 0;
  if (!attn_out_args_ok(a->w_out_frag, a->bias_out, a->residual, a->ld_res, rows)) return 0;
  return 1;
}

extern "C" int ca_tattn_fused(const ca_tattn_args* a, void* stream) {
  CA_REQUIRE(a != nullptr, "ca_tattn_fused: null args");
  CA_REQUIRE(ca_tattn_fused_supported(a), "ca_tattn_fused: arguments outside what the fused temporal attention takes (C = 320, 8 heads, 16 frames -- or 8 / 32 with w_out_frag --, tokens %% (128 / frames) == 0, "
                                           ">= 16384 rows, fragment-ordered weights, 16-byte aligned operands, 32-bit byte offsets): ask ca_tattn_fused_supported() first");
  const int64_t rows = (int64_t)a->batch * a->frames * a->tokens;
  TattnParams p{};
  p.x = (const u16*)a->x;
  p.wf = (const u16*)a->w_frag;
  p.gamma = a->gamma;
  p.bp = a->bias_pe;
  p.o = (u16*)a->o;
  p.lda = (int)a->lda, p.ldo = (int)a->ldo, p.ld_bp = (int)a->ld_bias_pe;
  p.hw = a->tokens;
  p.ln_eps = a->ln_eps;
  p.scale_log2 = a->scale * 1.4426950408889634f;
  p.x_bytes = (unsigned)(((rows - 1) * a->lda + 320) * 2);
  p.o_bytes = (unsigned)(((rows - 1) * a->ldo + 320) * 2);
  const int tiles = (int)((int64_t)a->batch * a->tokens / (128 / a->frames));
  static const int dbg = CA_KNOB("CA_TATTN_DBG", 0);
  p.dbg = dbg;
  if (a->w_out_frag) {  // with the output projection: one block of eight waves per CU (k_tattn_out)
    p.out.wof = (const u16*)a->w_out_frag;
    p.out.bias = a->bias_out;
    p.out.res = (const u16*)a->residual;
    p.out.ld_res = (int)a->ld_res;
    p.out.res_bytes = a->residual ? (unsigned)(((rows - 1) * a->ld_res + 320) * 2) : 0u;
    const unsigned grid1 = (unsigned)(tiles < ar_cu_count() ? tiles : ar_cu_count());
#define CA_TATTN_OUT(FR)                                                                                                              \
  if (a->dtype == CA_BF16) hipLaunchKernelGGL((k_tattn_out<CA_BF16, FR>), dim3(grid1), dim3(512), 0, (hipStream_t)stream, p, tiles); \
  else hipLaunchKernelGGL((k_tattn_out<CA_F16, FR>), dim3(grid1), dim3(512), 0, (hipStream_t)stream, p, tiles);
    if (a->frames == 8) { CA_TATTN_OUT(8) }
    else if (a->frames == 32) { CA_TATTN_OUT(32) }
    else { CA_TATTN_OUT(16) }
#undef CA_TATTN_OUT
    CA_CHECK_LAUNCH("ca_tattn_fused(out)");
    return CA_OK;
  }
  const int slots = (dbg & 16) ? ar_cu_count() : 2 * ar_cu_count();  // (16: one block per CU -- stamps of a wave that has its SIMD to itself)
  const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
  if (a->dtype == CA_BF16) hipLaunchKernelGGL((k_tattn_fused<CA_BF16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, tiles);
  else hipLaunchKernelGGL((k_tattn_fused<CA_F16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, tiles);
  CA_CHECK_LAUNCH("ca_tattn_fused");
  return CA_OK;
}

extern "C" int ca_xattn_pack_w(const void* w, int32_t n, int32_t k, void* dst, void* stream) {
  CA_REQUIRE(w && dst, "ca_xattn_pack_w: null operand");
  CA_REQUIRE(n == 320 && k == 320, "ca_xattn_pack_w: n=%d k=%d (Wq: 320 x 320)", n, k);
  CA_REQUIRE((((uintptr_t)w | (uintptr_t)dst) & 15) == 0, "ca_xattn_pack_w: operands must be 16-byte aligned");
  static_assert(CA_XATTN_WF_ELEMS == CA_XATTN_W_FRAG_ELEMS && CA_XATTN_KVF_ELEMS == CA_XATTN_KV_FRAG_ELEMS, "header constants");
  hipLaunchKernelGGL(k_xattn_pack_w, dim3((CA_XATTN_WF_ELEMS / 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const u16*)w, (u16*)dst);
  CA_CHECK_LAUNCH("ca_xattn_pack_w");
  return CA_OK;
}

extern "C" int ca_xattn_pack_kv(const void* kv, int64_t ld, int32_t kv_batches, int32_t rows_per_batch, int32_t row_offset, int32_t nk, float scale, int32_t dtype,
                                void* dst, void* stream) {
  CA_REQUIRE(kv && dst, "ca_xattn_pack_kv: null operand");
  CA_REQUIRE(kv_batches > 0 && rows_per_batch > 0 && row_offset >= 0 && ((nk > 64 && nk <= 80) || (nk >= 1 && nk <= 16)) && row_offset + nk <= rows_per_batch && ld >= 640,
             "ca_xattn_pack_kv: kv_batches=%d rows_per_batch=%d row_offset=%d nk=%d (65..80 text keys, or 1..16 image-prompt tokens) ld=%lld (>= 640: K | V of 8 heads x 40)", kv_batches, rows_per_batch, row_offset, nk,
             (long long)ld);
  CA_REQUIRE(dtype == CA_BF16 || dtype == CA_F16, "ca_xattn_pack_kv: dtype %d", dtype);
  CA_REQUIRE(((uintptr_t)dst & 15) == 0 && scale > 0.f, "ca_xattn_pack_kv: dst must be 16-byte aligned, scale > 0");
  const float sl2 = scale * 1.4426950408889634f;
  if (dtype == CA_BF16) hipLaunchKernelGGL((k_xattn_pack_kv<CA_BF16>), dim3((unsigned)kv_batches * 8), dim3(64), 0, (hipStream_t)stream, (const u16*)kv, ld, rows_per_batch, row_offset, nk, sl2, (u16*)dst);
  else hipLaunchKernelGGL((k_xattn_pack_kv<CA_F16>), dim3((unsigned)kv_batches * 8), dim3(64), 0, (hipStream_t)stream, (const u16*)kv, ld, rows_per_batch, row_offset, nk, sl2, (u16*)dst);
  CA_CHECK_LAUNCH("ca_xattn_pack_kv");
  return CA_OK;
}

extern "C" int ca_xattn_fused_supported(const ca_xattn_args* a) {
  if (!a || !a->x || !a->wq_frag || !a->kv_frag || !a->o) return 0;
  if (a->c != 320 || a->heads != 8 || a->m < 16384 || a->tokens < 128 || a->tokens % 128 || a->m % a->tokens) return 0;
  if (a->nk <= 64 || a->nk > 80 || a->frames_per_kv < 1 || a->kv_mod < 1) return 0;
  {  // image z reads text batch (z / frames_per_kv) % kv_mod: the largest index used must exist among the packed batches
    const int images = a->m / a->tokens;
    const int groups = (images + a->frames_per_kv - 1) / a->frames_per_kv;
    const int needed = groups < a->kv_mod ? groups : a->kv_mod;
    if (a->kv_batches < needed) return 0;
  }
  if (a->dtype != CA_BF16 && a->dtype != CA_F16) return 0;
  if (a->lda % 8 || a->ldo % 8 || a->lda < 320 || a->ldo < 320) return 0;
  if ((((uintptr_t)a->x | (uintptr_t)a->o | (uintptr_t)a->wq_frag | (uintptr_t)a->kv_frag | (uintptr_t)a->bias) & 15) != 0) return 0;
  const int64_t lim = 0x7FFFFF00ll;
  if (((int64_t)(a->m - 1) * a->lda + 320) * 2 >= lim || ((int64_t)(a->m - 1) * a->ldo + 320) * 2 >= lim) return 0;
  if (!(a->ln_eps > 0.f)) return 0;
  if (!attn_out_args_ok(a->w_out_frag, a->bias_out, a->residual, a->ld_res, a->m)) return 0;
  if (a->kv_frag_ip) {  // ABI v13: the image-prompt tokens ride on the eight-wave kernel (with the output stage) only
    if (!a->w_out_frag || a->nk_ip < 1 || a->nk_ip > 16 || ((uintptr_t)a->kv_frag_ip & 15) != 0 || !(a->ip_scale == a->ip_scale) || a->ip_scale - a->ip_scale != 0.f) return 0;
  } else if (a->nk_ip != 0) {
    return 0;
  }
  return 1;
}

extern "C" int ca_xattn_fused(const ca_xattn_args* a, void* stream) {
  CA_REQUIRE(a != nullptr, "ca_xattn_fused: null args");
  CA_REQUIRE(ca_xattn_fused_supported(a), "ca_xattn_fused: arguments outside what the fused text cross-attention takes (C = 320, 8 heads, 65..80 keys, tokens %% 128 == 0, "
                                           ">= 16384 rows, packed weights and K / V, 16-byte aligned operands, 32-bit byte offsets): ask ca_xattn_fused_supported() first");
  XattnParams p{};
  p.x = (const u16*)a->x;
  p.wf = (const u16*)a->wq_frag;
  p.bias = a->bias;
  p.kvf = (const u16*)a->kv_frag;
  p.o = (u16*)a->o;
  p.lda = (int)a->lda, p.ldo = (int)a->ldo;
  p.m = a->m, p.tokens = a->tokens, p.frames_per_kv = a->frames_per_kv, p.kv_mod = a->kv_mod, p.nk = a->nk;
  p.ln_eps = a->ln_eps;
  p.x_bytes = (unsigned)(((int64_t)(a->m - 1) * a->lda + 320) * 2);
  p.o_bytes = (unsigned)(((int64_t)(a->m - 1) * a->ldo + 320) * 2);
  p.kvf_bytes = (unsigned)((int64_t)a->kv_batches * 8 * CA_XATTN_KVF_ELEMS * 2);
  const int tiles = a->m / 128;
  if (a->w_out_frag) {  // with the output projection: one block of eight waves per CU (k_xattn_out)
    p.out.wof = (const u16*)a->w_out_frag;
    p.out.bias = a->bias_out;
    p.out.res = (const u16*)a->residual;
    p.out.ld_res = (int)a->ld_res;
    p.out.res_bytes = a->residual ? (unsigned)(((int64_t)(a->m - 1) * a->ld_res + 320) * 2) : 0u;
    const unsigned grid1 = (unsigned)(tiles < ar_cu_count() ? tiles : ar_cu_count());
    if (a->kv_frag_ip) {
      p.kvf_ip = (const u16*)a->kv_frag_ip, p.nk_ip = a->nk_ip, p.ip_scale = a->ip_scale;
      if (a->dtype == CA_BF16) hipLaunchKernelGGL((k_xattn_out<CA_BF16, true>), dim3(grid1), dim3(512), 0, (hipStream_t)stream, p, tiles);
      else hipLaunchKernelGGL((k_xattn_out<CA_F16, true>), dim3(grid1), dim3(512), 0, (hipStream_t)stream, p, tiles);
    } else if (a->dtype == CA_BF16) hipLaunchKernelGGL((k_xattn_out<CA_BF16, false>), dim3(grid1), dim3(512), 0, (hipStream_t)stream, p, tiles);
    else hipLaunchKernelGGL((k_xattn_out<CA_F16, false>), dim3(grid1), dim3(512), 0, (hipStream_t)stream, p, tiles);
    CA_CHECK_LAUNCH("ca_xattn_fused(out)");
    return CA_OK;
  }
  const int slots = 2 * ar_cu_count();
  const unsigned grid = (unsigned)(tiles < slots ? tiles : slots);
  if (a->dtype == CA_BF16) hipLaunchKernelGGL((k_xattn_fused<CA_BF16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, tiles);
  else hipLaunchKernelGGL((k_xattn_fused<CA_F16>), dim3(grid), dim3(256), 0, (hipStream_t)stream, p, tiles);
  CA_CHECK_LAUNCH("ca_xattn_fused");
  return CA_OK;
}

#ifdef CA_EXPERIMENTS
extern "C" int ca_debug_ff_stamps(unsigned long long* out) {  // 2 x 256 words, host pointer
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ca_ff_stamps), sizeof(unsigned long long) * 512) == hipSuccess ? 0 : -1;
}
#endif

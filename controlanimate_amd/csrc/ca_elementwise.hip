// Small HBM-bound kernels of the denoising loop: residual adds, layout changes at the API
// boundary, timestep embedding, fused CFG + scheduler update.
#include "ca_common.h"
#include <string.h>

// ---- error plumbing (host) ----------------------------------------------------------------
static thread_local char g_ca_err[512] = "";
void ca_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_ca_err, sizeof(g_ca_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* ca_last_error(void) { return g_ca_err; }
extern "C" int ca_abi_version(void) { return CA_ABI_VERSION; }

namespace {

template <int DT>
__global__ __launch_bounds__(256) void k_add_bcast(const u16* a, const u16* b, u16* out, int64_t n8, int64_t period8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float fa[8], fb[8];
    unpack8<DT>(ld16(a + i * 8), fa);
    unpack8<DT>(ld16(b + (i % period8) * 8), fb);
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] += fb[j];
    st16(out + i * 8, pack8<DT>(fa));
  }
}

__global__ void k_silu_f32(const float* x, float* y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = silu_f(x[i]);
}

template <int DT>
__global__ void k_timestep_embedding(const float* t_dev, float t_host, u16* out, int batch, int dim) {
  const int half = dim >> 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * half) return;
  const int b = i / half, k = i - b * half;
  const float t = t_dev ? t_dev[b] : t_host;
  const float freq = expf(-9.210340371976184f * (float)k / (float)half);  // ln(10000)
  const float arg = t * freq;
  // flip_sin_to_cos=True: [cos | sin]
  out[(int64_t)b * dim + k] = Elem<DT>::from_f(cosf(arg));
  out[(int64_t)b * dim + half + k] = Elem<DT>::from_f(sinf(arg));
}

template <int DT>
__global__ void k_latents_to_nhwc(const float* lat, u16* out, int b0, int c, int f, int h, int w, int cpad, int rep, float in_scale) {
  // one thread per output pixel (of the un-replicated tensor)
  const int64_t npix = (int64_t)b0 * f * h * w;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  int64_t t = i;
  const int x = t % w; t /= w;
  const int y = t % h; t /= h;
  const int fr = t % f; t /= f;
  const int b = (int)t;
  for (int r = 0; r < rep; ++r) {
    u16* dst = out + ((((int64_t)(r * b0 + b) * f + fr) * h + y) * w + x) * cpad;
    for (int ch = 0; ch < cpad; ++ch) {
      float v = 0.f;
      if (ch < c) v = lat[((((int64_t)b * c + ch) * f + fr) * h + y) * w + x] * in_scale;
      dst[ch] = Elem<DT>::from_f(v);
    }
  }
}

template <int DT>
__global__ void k_nhwc_to_ncfhw_f32(const void* xin, float* out, int b, int c, int f, int h, int w, int ldx, int x_is_f32) {
  const int64_t n = (int64_t)b * c * f * h * w;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t t = i;
  const int x = t % w; t /= w;
  const int y = t % h; t /= h;
  const int fr = t % f; t /= f;
  const int ch = t % c; t /= c;
  const int bb = (int)t;
  const int64_t src = ((((int64_t)bb * f + fr) * h + y) * w + x) * ldx + ch;
  out[i] = x_is_f32 ? reinterpret_cast<const float*>(xin)[src] : Elem<DT>::to_f(reinterpret_cast<const u16*>(xin)[src]);
}

struct Strides5 {
  int64_t s[5];
};

template <int DT>
__global__ void k_ncfhw_to_nhwc(const void* xin, int kind, Strides5 st, u16* out, int b, int c, int f, int h, int w, int cpad) {
  const int64_t n = (int64_t)b * f * h * w * cpad;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t t = i;
  const int ch = t % cpad; t /= cpad;
  const int x = t % w; t /= w;
  const int y = t % h; t /= h;
  const int fr = t % f; t /= f;
  const int bb = (int)t;
  float v = 0.f;
  if (ch < c) {
    const int64_t src = bb * st.s[0] + ch * st.s[1] + fr * st.s[2] + y * st.s[3] + x * st.s[4];
    if (kind == 0) v = reinterpret_cast<const float*>(xin)[src];
    else if (kind == 1) v = Elem<CA_F16>::to_f(reinterpret_cast<const u16*>(xin)[src]);
    else v = Elem<CA_BF16>::to_f(reinterpret_cast<const u16*>(xin)[src]);
  }
  out[i] = Elem<DT>::from_f(v);
}

struct Coef7 {
  float c[7];
};

__global__ void k_cfg_scheduler_step(const float* eps, int ld_eps, int rep, float guidance, const float* lat,
                                     const float* noise, float* prev, float* den_out, int c, int f, int h, int w,
                                     Coef7 k, float clip) {
  const int64_t n = (int64_t)c * f * h * w;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t t = i;
  const int x = t % w; t /= w;
  const int y = t % h; t /= h;
  const int fr = t % f; t /= f;
  const int ch = (int)t;
  const int64_t pix = ((int64_t)fr * h + y) * w + x;
  float e = eps[pix * ld_eps + ch];
  if (rep == 2) {
    const float ec = eps[((int64_t)f * h * w + pix) * ld_eps + ch];
    e = e + guidance * (ec - e);
  }
  const float xs = lat[i];
  float x0 = (xs - k.c[0] * e) * k.c[1];
  if (clip > 0.f) x0 = fminf(fmaxf(x0, -clip), clip);
  const float den = k.c[2] * x0 + k.c[3] * xs;
  float pv = k.c[4] * den + k.c[5] * e;
  if (noise) pv += k.c[6] * noise[i];
  prev[i] = pv;
  if (den_out) den_out[i] = den;
}

struct LinComb {
  const float* x[8];
  float c[8];
  int n_terms;
};

__global__ void k_lincomb(float* out, LinComb a, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < a.n_terms) v = fmaf(a.c[k], a.x[k][i], v);
    out[i] = v;
  }
}

inline unsigned blocks_for(int64_t n, int bs, int64_t cap = 1 << 20) {
  int64_t b = (n + bs - 1) / bs;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace

extern "C" int ca_add_bcast(const void* a, const void* b, void* out, int64_t n, int64_t b_period, int32_t dtype, void* stream) {
  CA_REQUIRE(a && b && out, "ca_add_bcast: null operand");
  CA_REQUIRE(n > 0 && n % 8 == 0 && b_period > 0 && b_period % 8 == 0, "ca_add_bcast: n=%lld period=%lld must be multiples of 8", (long long)n, (long long)b_period);
  CA_REQUIRE(dtype == CA_BF16 || dtype == CA_F16, "ca_add_bcast: dtype %d", dtype);
  const unsigned blocks = blocks_for(n / 8, 256, 8192);
  if (dtype == CA_BF16) hipLaunchKernelGGL(k_add_bcast<CA_BF16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)a, (const u16*)b, (u16*)out, n / 8, b_period / 8);
  else hipLaunchKernelGGL(k_add_bcast<CA_F16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u16*)a, (const u16*)b, (u16*)out, n / 8, b_period / 8);
  CA_CHECK_LAUNCH("ca_add_bcast");
  return CA_OK;
}

// `times` copies of a buffer behind each other: torch.cat([x] * times) along the leading dimension with ONE read of x
// (the two CFG halves of the shared prefix, DESIGN.md section 3: the reference's torch.cat([latents] * 2) made them identical)
namespace {
__global__ __launch_bounds__(256) void k_repeat(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t n16, int times) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
    const u32x4 v = src[i];
    for (int t = 0; t < times; ++t) dst[(int64_t)t * n16 + i] = v;
  }
}
// any size / alignment (odd reduced-width shapes, views with a storage offset): byte granular
__global__ __launch_bounds__(256) void k_repeat_bytes(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int64_t n, int times) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const unsigned char v = src[i];
    for (int t = 0; t < times; ++t) dst[(int64_t)t * n + i] = v;
  }
}
}  // namespace

extern "C" int ca_repeat(const void* src, void* dst, int64_t bytes, int32_t times, void* stream) {
  CA_REQUIRE(src && dst, "ca_repeat: null operand");
  CA_REQUIRE(bytes > 0 && times >= 1 && times <= 64, "ca_repeat: bytes=%lld times=%d (1..64)", (long long)bytes, times);
  if (bytes % 16 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0)
    hipLaunchKernelGGL(k_repeat, dim3(blocks_for(bytes / 16, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, bytes / 16, times);
  else
    hipLaunchKernelGGL(k_repeat_bytes, dim3(blocks_for(bytes, 256, 8192)), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src,
                       (unsigned char*)dst, bytes, times);
  CA_CHECK_LAUNCH("ca_repeat");
  return CA_OK;
}

// Row softmax of an fp32 score matrix into the activation dtype: the VAE's single-head, head_dim 512
// attention (diffusers AutoencoderKL mid block) runs as GEMM (scores, fp32) -> this -> GEMM (P V);
// one block per row, three passes over a row that stays in L2.
namespace {
template <int DT>
__global__ __launch_bounds__(256) void k_softmax_rows(const float* x, u16* y, int cols, int64_t ldx, int64_t ldy, float scale_log2) {
  __shared__ float red[4];
  const float* xr = x + (int64_t)blockIdx.x * ldx;
  u16* yr = y + (int64_t)blockIdx.x * ldy;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float m = -INFINITY;
  for (int c = tid * 4; c < cols; c += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
    m = fmaxf(fmaxf(fmaxf(m, v[0]), fmaxf(v[1], v[2])), v[3]);
  }
  for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) red[wid] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  const float nm = -m * scale_log2;
  float sum = 0.f;
  for (int c = tid * 4; c < cols; c += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) sum += __builtin_amdgcn_exp2f(fmaf(v[k], scale_log2, nm));
  }
  for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) red[wid] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int c = tid * 4; c < cols; c += 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
    u32x2 o;
    o[0] = pack2<DT>(__builtin_amdgcn_exp2f(fmaf(v[0], scale_log2, nm)) * inv, __builtin_amdgcn_exp2f(fmaf(v[1], scale_log2, nm)) * inv);
    o[1] = pack2<DT>(__builtin_amdgcn_exp2f(fmaf(v[2], scale_log2, nm)) * inv, __builtin_amdgcn_exp2f(fmaf(v[3], scale_log2, nm)) * inv);
    *reinterpret_cast<u32x2*>(yr + c) = o;
  }
}
}  // namespace

extern "C" int ca_softmax_rows(const float* x, void* y, int64_t rows, int32_t cols, int64_t ldx, int64_t ldy, float scale,
                               int32_t dtype, void* stream) {
  CA_REQUIRE(x && y, "ca_softmax_rows: null operand");
  CA_REQUIRE(rows > 0 && rows < (1ll << 31) && cols > 0 && cols % 4 == 0, "ca_softmax_rows: rows=%lld cols=%d (cols %% 4)", (long long)rows, cols);
  CA_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldx >= cols && ldy >= cols, "ca_softmax_rows: leading dimensions");
  CA_REQUIRE(dtype == CA_BF16 || dtype == CA_F16, "ca_softmax_rows: dtype %d", dtype);
  const float sl2 = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == CA_BF16) hipLaunchKernelGGL(k_softmax_rows<CA_BF16>, dim3((unsigned)rows), dim3(256), 0, st, x, (u16*)y, cols, ldx, ldy, sl2);
  else hipLaunchKernelGGL(k_softmax_rows<CA_F16>, dim3((unsigned)rows), dim3(256), 0, st, x, (u16*)y, cols, ldx, ldy, sl2);
  CA_CHECK_LAUNCH("ca_softmax_rows");
  return CA_OK;
}

extern "C" int ca_silu_f32(const float* x, float* y, int64_t n, void* stream) {
  CA_REQUIRE(x && y && n > 0, "ca_silu_f32: bad args");
  hipLaunchKernelGGL(k_silu_f32, dim3(blocks_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  CA_CHECK_LAUNCH("ca_silu_f32");
  return CA_OK;
}

extern "C" int ca_timestep_embedding(const float* t_dev, float t_host, void* out, int32_t batch, int32_t dim, int32_t dtype, void* stream) {
  CA_REQUIRE(out && batch > 0 && dim > 0 && dim % 2 == 0, "ca_timestep_embedding: bad args");
  CA_REQUIRE(dtype == CA_BF16 || dtype == CA_F16, "ca_timestep_embedding: dtype %d", dtype);
  const int n = batch * (dim / 2);
  if (dtype == CA_BF16) hipLaunchKernelGGL(k_timestep_embedding<CA_BF16>, dim3(blocks_for(n, 128)), dim3(128), 0, (hipStream_t)stream, t_dev, t_host, (u16*)out, batch, dim);
  else hipLaunchKernelGGL(k_timestep_embedding<CA_F16>, dim3(blocks_for(n, 128)), dim3(128), 0, (hipStream_t)stream, t_dev, t_host, (u16*)out, batch, dim);
  CA_CHECK_LAUNCH("ca_timestep_embedding");
  return CA_OK;
}

extern "C" int ca_latents_to_nhwc(const float* latents, void* out, int32_t b0, int32_t c, int32_t f, int32_t h, int32_t w,
                                  int32_t cpad, int32_t rep, float in_scale, int32_t dtype, void* stream) {
  CA_REQUIRE(latents && out, "ca_latents_to_nhwc: null operand");
  CA_REQUIRE(b0 > 0 && c > 0 && f > 0 && h > 0 && w > 0 && cpad >= c && cpad % 8 == 0 && rep >= 1, "ca_latents_to_nhwc: bad sizes");
  CA_REQUIRE(dtype == CA_BF16 || dtype == CA_F16, "ca_latents_to_nhwc: dtype %d", dtype);
  const int64_t npix = (int64_t)b0 * f * h * w;
  if (dtype == CA_BF16) hipLaunchKernelGGL(k_latents_to_nhwc<CA_BF16>, dim3(blocks_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, latents, (u16*)out, b0, c, f, h, w, cpad, rep, in_scale);
  else hipLaunchKernelGGL(k_latents_to_nhwc<CA_F16>, dim3(blocks_for(npix, 256)), dim3(256), 0, (hipStream_t)stream, latents, (u16*)out, b0, c, f, h, w, cpad, rep, in_scale);
  CA_CHECK_LAUNCH("ca_latents_to_nhwc");
  return CA_OK;
}

extern "C" int ca_nhwc_to_ncfhw_f32(const void* x, float* out, int32_t b, int32_t c, int32_t f, int32_t h, int32_t w,
                                    int32_t ldx, int32_t x_is_f32, int32_t dtype, void* stream) {
  CA_REQUIRE(x && out, "ca_nhwc_to_ncfhw_f32: null operand");
  CA_REQUIRE(b > 0 && c > 0 && f > 0 && h > 0 && w > 0 && ldx >= c, "ca_nhwc_to_ncfhw_f32: bad sizes");
  CA_REQUIRE(dtype == CA_BF16 || dtype == CA_F16, "ca_nhwc_to_ncfhw_f32: dtype %d", dtype);
  const int64_t n = (int64_t)b * c * f * h * w;
  if (dtype == CA_BF16) hipLaunchKernelGGL(k_nhwc_to_ncfhw_f32<CA_BF16>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, b, c, f, h, w, ldx, x_is_f32);
  else hipLaunchKernelGGL(k_nhwc_to_ncfhw_f32<CA_F16>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, out, b, c, f, h, w, ldx, x_is_f32);
  CA_CHECK_LAUNCH("ca_nhwc_to_ncfhw_f32");
  return CA_OK;
}

extern "C" int ca_ncfhw_to_nhwc(const void* x, int32_t x_kind, const int64_t strides[5], void* out, int32_t b, int32_t c,
                                int32_t f, int32_t h, int32_t w, int32_t cpad, int32_t dtype, void* stream) {
  CA_REQUIRE(x && out && strides, "ca_ncfhw_to_nhwc: null operand");
  CA_REQUIRE(x_kind >= 0 && x_kind <= 2, "ca_ncfhw_to_nhwc: x_kind %d", x_kind);
  CA_REQUIRE(b > 0 && c > 0 && f > 0 && h > 0 && w > 0 && cpad >= c, "ca_ncfhw_to_nhwc: bad sizes");
  CA_REQUIRE(dtype == CA_BF16 || dtype == CA_F16, "ca_ncfhw_to_nhwc: dtype %d", dtype);
  Strides5 st;
  for (int i = 0; i < 5; ++i) st.s[i] = strides[i];
  const int64_t n = (int64_t)b * f * h * w * cpad;
  if (dtype == CA_BF16) hipLaunchKernelGGL(k_ncfhw_to_nhwc<CA_BF16>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, x_kind, st, (u16*)out, b, c, f, h, w, cpad);
  else hipLaunchKernelGGL(k_ncfhw_to_nhwc<CA_F16>, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, x_kind, st, (u16*)out, b, c, f, h, w, cpad);
  CA_CHECK_LAUNCH("ca_ncfhw_to_nhwc");
  return CA_OK;
}

extern "C" int ca_cfg_scheduler_step(const float* eps, int32_t ld_eps, int32_t rep, float guidance, const float* latents,
                                     const float* noise, float* prev, float* denoised, int32_t c, int32_t f, int32_t h,
                                     int32_t w, const float coef[7], float clip, void* stream) {
  CA_REQUIRE(eps && latents && prev && coef, "ca_cfg_scheduler_step: null operand");
  CA_REQUIRE(rep == 1 || rep == 2, "ca_cfg_scheduler_step: rep %d", rep);
  CA_REQUIRE(c > 0 && f > 0 && h > 0 && w > 0 && ld_eps >= c, "ca_cfg_scheduler_step: bad sizes");
  Coef7 k;
  for (int i = 0; i < 7; ++i) k.c[i] = coef[i];
  const int64_t n = (int64_t)c * f * h * w;
  hipLaunchKernelGGL(k_cfg_scheduler_step, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, eps, ld_eps, rep, guidance, latents, noise, prev, denoised, c, f, h, w, k, clip);
  CA_CHECK_LAUNCH("ca_cfg_scheduler_step");
  return CA_OK;
}

extern "C" int ca_lincomb(float* out, const float* const* x, const float* coef, int32_t n_terms, int64_t n, void* stream) {
  CA_REQUIRE(out && x && coef, "ca_lincomb: null operand");
  CA_REQUIRE(n_terms >= 1 && n_terms <= 8, "ca_lincomb: n_terms=%d must be 1..8", n_terms);
  CA_REQUIRE(n > 0, "ca_lincomb: n=%lld", (long long)n);
  LinComb a{};
  a.n_terms = n_terms;
  for (int k = 0; k < n_terms; ++k) {
    CA_REQUIRE(x[k] != nullptr, "ca_lincomb: x[%d] is null", k);
    a.x[k] = x[k];
    a.c[k] = coef[k];
  }
  hipLaunchKernelGGL(k_lincomb, dim3(blocks_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, out, a, n);
  CA_CHECK_LAUNCH("ca_lincomb");
  return CA_OK;
}

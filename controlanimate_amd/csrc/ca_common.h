// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes, MFMA 16x16x32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/controlanimate_hip.h"

// Tuning knobs.  The product build has none: CA_KNOB(name, default) IS its default, so which kernel a call runs is a
// pure function of its arguments (ca_gemm_plan_name; tests/test_dispatch_plan.py).  Experiment builds (-DCA_EXPERIMENTS:
// `python -m controlanimate_amd._build --experiments`, loaded through CA_HIP_LIB for same-box A/B timing) read the
// environment variable once per process.
#include <stdlib.h>
#ifdef CA_EXPERIMENTS
#define CA_KNOB(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define CA_KNOB(name, dflt) (dflt)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

// ---- host-side error plumbing -------------------------------------------------------
void ca_set_error(const char* fmt, ...);
#define CA_FAIL(code, ...)        \
  do {                            \
    ca_set_error(__VA_ARGS__);    \
    return (code);                \
  } while (0)
#define CA_REQUIRE(cond, ...)                         \
  do {                                                \
    if (!(cond)) CA_FAIL(CA_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)
#define CA_CHECK_LAUNCH(name)                                                    \
  do {                                                                           \
    hipError_t e_ = hipGetLastError();                                           \
    if (e_ != hipSuccess) CA_FAIL(CA_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
  } while (0)

// ---- element-type traits --------------------------------------------------------------
template <int DT>
struct Elem;

template <>
struct Elem<CA_BF16> {
  static __device__ __forceinline__ float to_f(u16 h) { return __uint_as_float(((unsigned)h) << 16); }
  static __device__ __forceinline__ u16 from_f(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 (RNE) on gfx950
    return __builtin_bit_cast(u16, b);
  }
  static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  // 16x16x16: a lane holds k = 4*(lane>>4) .. +3 (8 bytes) of row / column lane&15
  static __device__ __forceinline__ f32x4 mfma16(u32x2 a, u32x2 b, f32x4 c) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
  }
};

template <>
struct Elem<CA_F16> {
  static __device__ __forceinline__ float to_f(u16 h) { return (float)__builtin_bit_cast(_Float16, h); }
  static __device__ __forceinline__ u16 from_f(float f) {
    _Float16 b = (_Float16)f;
    return __builtin_bit_cast(u16, b);
  }
  static __device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(u32x2 a, u32x2 b, f32x4 c) {
    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(h16x4, a), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
  }
};

// Two fp32 values -> one register of two 16-bit elements, round to nearest even.  (Round 4: written as a vector conversion hipcc
// emits ONE v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32 instead of the 3-4 instructions below -- measured on the whole step: no gain
// (61.04 / 60.74 vs 60.89 / 60.36 ms), and the bf16 d = 40 attention kernel then returns wrong values (rel 0.1; cause not
// isolated -- hipcc's wait-state accounting around MFMA results is not airtight on this target, see k_attn_short in ca_attention.hip).
// Kept scalar.)
template <int DT>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  return (unsigned)Elem<DT>::from_f(lo) | ((unsigned)Elem<DT>::from_f(hi) << 16);
}
// Packing of softmax probabilities (values in [0,1]): fp16 uses the single-instruction
// round-toward-zero pack (v_cvt_pkrtz_f16_f32); bf16's v_cvt_pk_bf16_f32 already is one instruction.
template <int DT>
__device__ __forceinline__ unsigned pack2_prob(float lo, float hi) {
  if (DT == CA_F16) {
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
    h2 r = __builtin_amdgcn_cvt_pkrtz(lo, hi);
    return __builtin_bit_cast(unsigned, r);
  }
  return pack2<DT>(lo, hi);
}
template <int DT>
__device__ __forceinline__ void unpack8(u32x4 v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = Elem<DT>::to_f((u16)(v[i] & 0xffffu));
    f[2 * i + 1] = Elem<DT>::to_f((u16)(v[i] >> 16));
  }
}
template <int DT>
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack2<DT>(f[2 * i], f[2 * i + 1]);
  return v;
}

// x * sigmoid(x) with the hardware reciprocal (1 ulp): a plain `/` expands to the ~10-instruction IEEE
// division sequence (v_div_scale/v_div_fmas/v_div_fixup) per element, which made GroupNorm+SiLU VALU-bound.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf-GELU x * Phi(x) without transcendentals: Phi(x) - 1/2 = xc * P(xc^2), xc = clamp(x, +-4.5), P a
// degree-9 minimax-style fit (Chebyshev nodes, reweighted least squares; tools/fit_gelu.py).
// |gelu error| <= 6e-5 for all x (the fp16 output spacing at |gelu| ~ 0.25 is 2.4e-4).  14 full-rate
// VALU ops instead of ~16 + v_exp + v_rcp (quarter rate): the GEGLU epilogue was as long as the
// whole K loop of the K = 320 feed-forward GEMM.
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -4.5f, 4.5f);
  const float u = xc * xc;
  float p = -1.726317873e-12f;
  p = fmaf(p, u, 2.022429585e-10f);
  p = fmaf(p, u, -1.056706100e-08f);
  p = fmaf(p, u, 3.278913994e-07f);
  p = fmaf(p, u, -6.813716936e-06f);
  p = fmaf(p, u, 1.017339964e-04f);
  p = fmaf(p, u, -1.142714871e-03f);
  p = fmaf(p, u, 9.891773574e-03f);
  p = fmaf(p, u, -6.642068177e-02f);
  p = fmaf(p, u, 3.989246786e-01f);
  return fmaf(x, xc * p, 0.5f * x);
}

// Two at a time: gfx950 has packed fp32 FMA / MUL (v_pk_fma_f32: two lanes' worth of fp32 per instruction at full
// rate), so the pair of gates a GEGLU fragment holds costs 9 + 3 packed instructions instead of 2 x 14.  Same
// polynomial, same operation order per element: bit-identical to gelu_erf_f.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf_f2(f32x2 x) {
#ifndef CA_GELU_SCALAR  // v_pk_fma_f32 / v_pk_mul_f32: 9 + 3 packed instructions per pair (the product form)
  const f32x2 xc = {__builtin_amdgcn_fmed3f(x[0], -4.5f, 4.5f), __builtin_amdgcn_fmed3f(x[1], -4.5f, 4.5f)};
  const f32x2 u = xc * xc;
  f32x2 p = {-1.726317873e-12f, -1.726317873e-12f};
  p = __builtin_elementwise_fma(p, u, (f32x2){2.022429585e-10f, 2.022429585e-10f});
  p = __builtin_elementwise_fma(p, u, (f32x2){-1.056706100e-08f, -1.056706100e-08f});
  p = __builtin_elementwise_fma(p, u, (f32x2){3.278913994e-07f, 3.278913994e-07f});
  p = __builtin_elementwise_fma(p, u, (f32x2){-6.813716936e-06f, -6.813716936e-06f});
  p = __builtin_elementwise_fma(p, u, (f32x2){1.017339964e-04f, 1.017339964e-04f});
  p = __builtin_elementwise_fma(p, u, (f32x2){-1.142714871e-03f, -1.142714871e-03f});
  p = __builtin_elementwise_fma(p, u, (f32x2){9.891773574e-03f, 9.891773574e-03f});
  p = __builtin_elementwise_fma(p, u, (f32x2){-6.642068177e-02f, -6.642068177e-02f});
  p = __builtin_elementwise_fma(p, u, (f32x2){3.989246786e-01f, 3.989246786e-01f});
  return __builtin_elementwise_fma(x, xc * p, (f32x2){0.5f, 0.5f} * x);
#else
  // -DCA_GELU_SCALAR (A/B builds): two scalar evaluations.  Round 6 measured a packed fp32 instruction at 2.8 v_fma_f32 issue times at 2-4 waves
  // per SIMD (tools/probe_valu_rates.hip), i.e. 1.4 per element -- but the step does not notice (packed / scalar, same box, ms per step:
  // 49.62 / 49.79, 49.48 / 49.53) and the scalar form spills in the epilogues of k_gemm_ps: the packed form stays.  Bit-identical either way.
  float a = x[0], b = x[1];
  asm volatile("" : "+v"(a), "+v"(b));
  return (f32x2){gelu_erf_f(a), gelu_erf_f(b)};
#endif
}

// epilogue activation selected at run time (wave-uniform)
__device__ __forceinline__ float act_f(float x, int act) {
  if (act == CA_ACT_SILU) return silu_f(x);
  if (act == CA_ACT_QUICK_GELU) return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
  if (act == CA_ACT_GELU) return gelu_erf_f(x);
  return x;
}

// Three-/two-input max.  Written with fmaxf so the compiler sees the data dependence on MFMA results
// (its hazard recogniser does not look inside inline asm: a hand-written v_max3_f32 read the
// accumulators too early).  Files whose max chains consume MFMA output are compiled with
// -fno-honor-nans, which drops the `v_max_f32 x, x, x` sNaN-quieting hipcc otherwise emits per input.
__device__ __forceinline__ float vmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float vmax2(float a, float b) { return fmaxf(a, b); }
// max over the four 16-lane rows of a wave (lanes l, l^16, l^32, l^48), result in every lane: gfx950
// v_permlane{32,16}_swap exchange half-waves / odd-even rows in the VALU (no LDS round trip).
__device__ __forceinline__ float rowgroup_max(float m) {
  unsigned u = __float_as_uint(m);
  u32x2 a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = __float_as_uint(vmax2(__uint_as_float(a[0]), __uint_as_float(a[1])));
  u32x2 b = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return vmax2(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// sum over the four 16-lane rows, result in every lane (same two swaps)
__device__ __forceinline__ float rowgroup_sum(float m) {
  unsigned u = __float_as_uint(m);
  u32x2 a = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
  u32x2 b = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

// Bijective XCD-aware block remap (cdna_hip_programming.md T1): blocks are dispatched
// round-robin over the 8 XCDs; give each XCD a contiguous range of logical tile ids so that
// tiles sharing an operand panel hit the same L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned nx = 8;
  if (nwg < 2 * nx) return bid;
  unsigned xcd = bid % nx, idx = bid / nx;
  unsigned q = nwg / nx, r = nwg % nx;
  unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

static inline int ceil_div_i(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

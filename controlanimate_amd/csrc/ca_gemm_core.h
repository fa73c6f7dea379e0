// Shared pieces of the GEMM / implicit-GEMM convolution kernels (ca_gemm.hip, ca_gemm_pp.hip): kernel
// parameter block, tile order, LDS swizzle, epilogues.  Everything is `static`/inline: each translation
// unit gets its own copy.
#pragma once
#include "ca_common.h"
#include <stdlib.h>

namespace ca_gemm_detail {


struct GemmKParams {
  const u16* a;
  const u16* a2;
  const u16* w;
  void* c;
  const float* bias;
  const float* rowbias;
  const float* ln_stats;   // [M][2] (mean, rstd) of the A rows: LayerNorm folded into this GEMM (see ca_gemm_args)
  const float* ln_colsum;  // [N] sum_k W'[n][k]
  const u16* res;
  int64_t lda, lda2, ldc, ld_res, ld_rowbias;
  unsigned a_bytes, a2_bytes, w_bytes;  // buffer-descriptor sizes for the LDS-DMA variant
  int m, n;
  int c1, c2;      // channels (K per tap) from source 1 / 2
  int taps;        // 1 (dense) or 9 (3x3)
  int kc_tiles;    // ceil((c1+c2)/64)
  // conv geometry
  int hin, win, hout, wout, stride, ups, pad_lo;  // pad_lo: zero rows/cols before the image (after: implicit)
  int rows_per_group;
  float alpha, post;
  int act, geglu, out_f32;
  // split-K (small-M convolutions): `splits` blocks share one output tile, each reduces a contiguous
  // range of the K tiles into its own fp32 slab partial[split][m][n]; k_splitk_reduce adds the slabs
  // in a fixed order and applies the epilogue (deterministic, no atomics)
  int splits;
  float* partial;
  int dbg;  // timing experiments (CA_PP_DBG): 1 = no epilogue, 2 = no main loop
  int tap_inner;  // K-tile order of the implicit-GEMM convolution (see k_tile_split)
  float* row_sums;  // producer side: [M][N / 320] (sum, sum of squares) of the STORED output rows per 320-column tile (k_gemm_pp2 only)
  int ln_parts;     // consumer side: ln_stats holds [M][ln_parts] such partial sums, finished here with ln_eps; 0: (mean, rstd)
  int ln_inline;  // ln_colsum without ln_stats: the kernel computes (mean, rstd) of the A rows itself (k_gemm_wres, k_gemm_ar)
  float ln_eps;
  const u16* wf;  // ABI v9: W once more in MFMA-fragment order (ca_pack_w_frag), or NULL: the activation-resident kernel reads it
  // round 5: row-grouped weights (k_gemm_pq dense, EPI = 0): rows [g * w_group_rows, (g + 1) * w_group_rows) of A use the weight matrix
  // at w + g * w_group_stride BYTES -- the sixteen independent GEMMs of a Winograd convolution as one launch (ca_conv_wino.h).  0 = off.
  int w_group_rows;
  unsigned w_group_stride;
};

constexpr int BK = 64;
#ifndef CA_GEMM_ABLATE
#define CA_GEMM_ABLATE 0  // timing experiments only: 1 = no MFMA / fragment reads (DMA + barriers only), 2 = no DMA after tile 0,
                          // 3 = 2 + one barrier per tile, 4 = 2 + fragments read once (MFMA + barriers only), 5 = 2 + no barriers
#endif

// Tile order inside an XCD's contiguous id range: groups of GROUP_M row-tiles are swept column by
// column, so the ~128 blocks resident on an XCD share 8 A panels and ~16 W panels in its 4 MB L2.
// (With a plain row-major order every row-tile streamed the whole weight matrix again: measured
// FETCH_SIZE 1.6 GB for the 8192x10240x1280 GEGLU GEMM whose operands total 47 MB.)
constexpr int GROUP_M = 8;
__device__ __forceinline__ void tile_coords(unsigned bid, int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
  const unsigned gsz = GROUP_M * tiles_n;
  const unsigned group = bid / gsz, in = bid - group * gsz;
  const int first_m = group * GROUP_M;
  const int gm = tiles_m - first_m < GROUP_M ? tiles_m - first_m : GROUP_M;
  tile_m = first_m + in % gm;
  tile_n = in / gm;
}

// K tile t of an implicit-GEMM 3x3 convolution -> (tap, 64-channel tile).  tap_inner: the nine taps of one channel
// tile follow each other, so the ~4 input rows x 64 channels a block touches nine times stay in its XCD's L2 between
// the touches (32 KB per block); with the taps outermost the whole channel depth (164 KB per block at C = 320, more
// than the 4 MB L2 over the ~32 resident blocks of an XCD) passed between two touches and the counters showed the
// activations fetched ~4.8x past L2 (profiles/round1_pmc_traffic.json).  Only the fp32 summation order changes.
__device__ __forceinline__ void k_tile_split(const GemmKParams& p, int t, int kct, int& tap, int& cc) {
  if (p.taps == 1) {
    tap = 0;
    cc = t;
  } else if (p.tap_inner) {
    cc = t / p.taps;
    tap = t - cc * p.taps;
  } else {
    tap = t / kct;
    cc = t - tap * kct;
  }
}

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return row * BK + ((chunk ^ ((row >> 1) & 7)) << 3);
}

// ---- direct epilogue (N < 8 only): lane holds C[m = .. + l15][n = .. + 4g + (0..3)] -----------
template <int DT, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_direct(const GemmKParams& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                                     int l15, int g) {
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * TM * 16 + i * 16 + l15;
    if (m >= p.m) continue;
    const float* rbp = p.rowbias ? p.rowbias + (int64_t)(m / p.rows_per_group) * p.ld_rowbias : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * TN * 16 + j * 16 + g * 4;
      if (n >= p.n) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
      if (p.bias) {
        f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += b[r];
      }
      if (rbp) {
        f32x4 b = *reinterpret_cast<const f32x4*>(rbp + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += b[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
      if (p.res) {
        u32x2 rr = *reinterpret_cast<const u32x2*>(p.res + (int64_t)m * p.ld_res + n);
        v[0] += Elem<DT>::to_f((u16)(rr[0] & 0xffffu));
        v[1] += Elem<DT>::to_f((u16)(rr[0] >> 16));
        v[2] += Elem<DT>::to_f((u16)(rr[1] & 0xffffu));
        v[3] += Elem<DT>::to_f((u16)(rr[1] >> 16));
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= p.post;
      if (p.act != CA_ACT_NONE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_f(v[r], p.act);
      }
      if (p.geglu) {
        const f32x2 gg = gelu_erf_f2((f32x2){v[1], v[3]});
        float o0 = v[0] * gg[0];
        float o1 = v[2] * gg[1];
        const int64_t off = (int64_t)m * p.ldc + (n >> 1);
        if (p.out_f32) {
          float* cp = reinterpret_cast<float*>(p.c) + off;
          cp[0] = o0;
          cp[1] = o1;
        } else {
          *reinterpret_cast<unsigned*>(reinterpret_cast<u16*>(p.c) + off) = pack2<DT>(o0, o1);
        }
      } else {
        const int64_t off = (int64_t)m * p.ldc + n;
        if (p.out_f32) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.c) + off) = (f32x4){v[0], v[1], v[2], v[3]};
        } else {
          u32x2 o;
          o[0] = pack2<DT>(v[0], v[1]);
          o[1] = pack2<DT>(v[2], v[3]);
          *reinterpret_cast<u32x2*>(reinterpret_cast<u16*>(p.c) + off) = o;
        }
      }
    }
  }
}

// ---- LDS-staged epilogue: the accumulator fragments (8-byte pieces scattered over 16 rows per
// store) are transposed through LDS so that global traffic is 16-byte accesses covering whole
// 256-byte row segments: coalesced residual reads and output writes.  The K-loop buffers are free
// at this point (the loop ends with a barrier).  Staged value = (acc + bias + rowbias) * alpha
// rounded to the activation type; the residual is added in fp32 afterwards (the reference's fp16
// pipeline rounds at the same place: linear output, then `+ hidden_states`).
// ROWSUM (k_gemm_pp2): besides storing, leave (sum, sum of squares) of every stored row's BN columns in p.row_sums --
// the LayerNorm statistics of the NEXT layer without its pass over the tensor (ca_gemm_args.row_sums_out).  `rs` is LDS
// scratch of BM x BN/8 float2 behind the staging tile; fixed summation order (deterministic).
template <int DT, int BM, int BN, int TM, int TN, int NT = 256, bool ROWSUM = false>
__device__ __forceinline__ void gemm_epilogue(const GemmKParams& p, f32x4 (&acc)[TM][TN], u16* cs, int m0, int n0, int wm, int wn,
                                              int l15, int g, int tid, float2* rs = nullptr) {
  if (p.n < 8) {  // conv_out (Cout = 4)
    gemm_epilogue_direct<DT, TM, TN>(p, acc, m0, n0, wm, wn, l15, g);
    return;
  }
  constexpr int CLD = BN + 8;
  // folded LayerNorm: this lane's TM row statistics and TN column-sum quads, loaded once
  float2 ln_st[TM];
  f32x4 ln_cs[TN];
  if (p.ln_stats) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * TM * 16 + i * 16 + l15;
      if (p.ln_parts > 0) {  // partial sums left by the producing GEMM's epilogue: finish (mean, rstd) here, in order
        float a = 0.f, b = 0.f;
        if (m < p.m) {
          for (int q = 0; q < p.ln_parts; ++q) {
            const float2 v = *reinterpret_cast<const float2*>(p.ln_stats + ((int64_t)m * p.ln_parts + q) * 2);
            a += v.x;
            b += v.y;
          }
        }
        const float inv = 1.f / (float)(p.c1 + p.c2);
        const float mean = a * inv;
        ln_st[i] = make_float2(mean, rsqrtf(fmaxf(b * inv - mean * mean, 0.f) + p.ln_eps));
      } else {
        ln_st[i] = m < p.m ? *reinterpret_cast<const float2*>(p.ln_stats + (int64_t)m * 2) : make_float2(0.f, 0.f);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * TN * 16 + j * 16 + g * 4;
      ln_cs[j] = n < p.n ? *reinterpret_cast<const f32x4*>(p.ln_colsum + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  // per-row-group bias: when the wave's 16*TM rows fall into ONE group (time-embedding rows of a resnet, the
  // per-frame positional row bias of a temporal q|k|v projection) its TN quads are loaded once, not per row tile
  f32x4 rb_q[TN];
  bool rb_uniform = false;
  if (p.rowbias) {
    const int r_first = m0 + wm * TM * 16, r_last = r_first + TM * 16 - 1;
    rb_uniform = r_last < p.m && r_first / p.rows_per_group == r_last / p.rows_per_group;
    if (rb_uniform) {
      const float* base = p.rowbias + (int64_t)(r_first / p.rows_per_group) * p.ld_rowbias;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 16 + j * 16 + g * 4;
        rb_q[j] = n < p.n ? *reinterpret_cast<const f32x4*>(base + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * TM * 16 + i * 16 + l15;
    const int m = m0 + row;
    const float* rbp = (p.rowbias && !rb_uniform && m < p.m) ? p.rowbias + (int64_t)(m / p.rows_per_group) * p.ld_rowbias : nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = wn * TN * 16 + j * 16 + g * 4;
      const int n = n0 + col;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
      if (n < p.n) {
        if (p.ln_stats) {  // LN(x) W'^T = rstd * (x W'^T - mean * colsum(W'))
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ln_st[i].y * (v[r] - ln_st[i].x * ln_cs[j][r]);
        }
        if (p.bias) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += b[r];
        }
        if (rb_uniform) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rb_q[j][r];
        } else if (rbp) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(rbp + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += b[r];
        }
      }
      if (p.alpha != 1.f) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
      }
      u32x2 o;
      o[0] = pack2<DT>(v[0], v[1]);
      o[1] = pack2<DT>(v[2], v[3]);
      *reinterpret_cast<u32x2*>(cs + row * CLD + col) = o;
    }
  }
  __syncthreads();
  constexpr int CH = BN / 8;
  const bool simple = p.post == 1.f && p.act == CA_ACT_NONE && !p.geglu && !p.out_f32;
  const bool geglu_only = p.geglu && p.post == 1.f && p.act == CA_ACT_NONE && !p.res && !p.out_f32;
#pragma unroll
  for (int u = 0; u < BM * CH / NT; ++u) {
    const int id = tid + u * NT;
    const int row = id / CH, c8 = id - row * CH;
    const int m = m0 + row, n = n0 + c8 * 8;
    if (m >= p.m || n >= p.n) continue;
    if (simple) {
      // no post scale / activation / GEGLU / fp32 output: the staged value IS the result, or it only takes the residual --
      // fp16 + fp16 is exact in fp32, so a packed add rounds exactly like the unpack / fp32 add / round of the general
      // path below (28-40 fewer VALU instructions per 16-byte chunk)
      u32x4 pk = ld16(cs + row * CLD + c8 * 8);
      if (p.res) {
        const u32x4 rr = ld16(p.res + (int64_t)m * p.ld_res + n);
        if (DT == CA_F16) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            unsigned s_;
            asm("v_pk_add_f16 %0, %1, %2" : "=v"(s_) : "v"(pk[k]), "v"(rr[k]));  // (inline asm: see ca_gemm_wres.h on the ext-vector miscompile)
            pk[k] = s_;
          }
        } else {
          float a_[8], r_[8];
          unpack8<DT>(pk, a_);
          unpack8<DT>(rr, r_);
#pragma unroll
          for (int k = 0; k < 8; ++k) a_[k] += r_[k];
          pk = pack8<DT>(a_);
        }
      }
      st16(reinterpret_cast<u16*>(p.c) + (int64_t)m * p.ldc + n, pk);
      if (ROWSUM && p.row_sums) {
        float r[8];
        unpack8<DT>(pk, r);
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          a += r[k];
          b = fmaf(r[k], r[k], b);
        }
        rs[row * CH + c8] = make_float2(a, b);
      }
      continue;
    }
    if (geglu_only) {  // the feed-forward projection: h * gelu(g) of the staged (rounded) pairs, nothing else
      float v[8];
      unpack8<DT>(ld16(cs + row * CLD + c8 * 8), v);
      const f32x2 g0 = gelu_erf_f2((f32x2){v[1], v[3]}), g1 = gelu_erf_f2((f32x2){v[5], v[7]});
      u32x2 w;
      w[0] = pack2<DT>(v[0] * g0[0], v[2] * g0[1]);
      w[1] = pack2<DT>(v[4] * g1[0], v[6] * g1[1]);
      *reinterpret_cast<u32x2*>(reinterpret_cast<u16*>(p.c) + (int64_t)m * p.ldc + (n >> 1)) = w;
      continue;
    }
    float v[8];
    unpack8<DT>(ld16(cs + row * CLD + c8 * 8), v);
    if (p.res) {
      float r[8];
      unpack8<DT>(ld16(p.res + (int64_t)m * p.ld_res + n), r);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += r[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= p.post;
    if (p.act != CA_ACT_NONE) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = act_f(v[k], p.act);
    }
    if (p.geglu) {
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; k += 2) {
        const f32x2 gg = gelu_erf_f2((f32x2){v[2 * k + 1], v[2 * k + 3]});
        o[k] = v[2 * k] * gg[0];
        o[k + 1] = v[2 * k + 2] * gg[1];
      }
      const int64_t off = (int64_t)m * p.ldc + (n >> 1);
      if (p.out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.c) + off) = (f32x4){o[0], o[1], o[2], o[3]};
      } else {
        u32x2 w;
        w[0] = pack2<DT>(o[0], o[1]);
        w[1] = pack2<DT>(o[2], o[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<u16*>(p.c) + off) = w;
      }
    } else {
      const int64_t off = (int64_t)m * p.ldc + n;
      if (p.out_f32) {
        float* cp = reinterpret_cast<float*>(p.c) + off;
        *reinterpret_cast<f32x4*>(cp) = (f32x4){v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(cp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
      } else {
        const u32x4 pk = pack8<DT>(v);
        st16(reinterpret_cast<u16*>(p.c) + off, pk);
        if (ROWSUM && p.row_sums) {  // of the values as stored (rounded)
          float r[8];
          unpack8<DT>(pk, r);
          float a = 0.f, b = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            a += r[k];
            b = fmaf(r[k], r[k], b);
          }
          rs[row * CH + c8] = make_float2(a, b);
        }
      }
    }
  }
  if (ROWSUM && p.row_sums) {
    __syncthreads();
    if (tid < BM && m0 + tid < p.m) {
      float a = 0.f, b = 0.f;
      for (int c8 = 0; c8 < CH; ++c8) {
        const float2 v = rs[tid * CH + c8];
        a += v.x;
        b += v.y;
      }
      *reinterpret_cast<float2*>(p.row_sums + ((int64_t)(m0 + tid) * (p.n / BN) + n0 / BN) * 2) = make_float2(a, b);
    }
  }
}


constexpr unsigned DMA_OOB = 0xFFFFFFF0u;  // beyond any descriptor size we accept -> hardware writes zeros

}  // namespace ca_gemm_detail

// ping-pong 256 x BN kernel family (ca_gemm_pp.hip); bn = 256 | 128
int ca_launch_gemm_pp(const ca_gemm_detail::GemmKParams& p, int dtype, int mode, int bn, unsigned tiles, hipStream_t st);
int ca_launch_gemm_ar(const ca_gemm_detail::GemmKParams& p, int dtype, hipStream_t st);  // ca_gemm_ar.hip

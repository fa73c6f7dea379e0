// Winograd F(2x2, 3x3) for the deep 3x3 convolutions of the small-latent levels (round 5, ABI v12: ca_conv_args.w_wino).
//
// Why: at the 16x16-latent level a resnet convolution is M = 8192 output pixels x N = 1280 x K = 9 * 1280 .. 9 * 2560 -- 128 tiles
// of 256 x 320 for 256 CUs, so it ran on the 128 x 320 ping-pong tiles at 34 % of the matrix peak (profiles/round5_mfma_util.txt:
// k_gemm_pp2<1, 1>, 8 % of the step).  F(2x2, 3x3) computes every 2 x 2 block of outputs from 16 products instead of 36
// (reference arithmetic: animatediff/models/resnet.py:12-20 InflatedConv3d = nn.Conv2d per frame):
//     Y = A^T [ (G g G^T) (.) (B^T d B) ] A        d: 4 x 4 input tile (pad 1), g: 3 x 3 filter, Y: 2 x 2 outputs
// i.e. sixteen INDEPENDENT dense GEMMs  M_f[t][co] = sum_ci V_f[t][ci] U_f[co][ci]  (f = 4 xi + nu, t = tile) with K = Cin instead of
// 9 Cin and 2.25 x fewer MFMAs -- and sixteen times the tile count, which is what the 256 x 320 kernel wants: the sixteen GEMMs are
// ONE launch of k_gemm_pq on the 16 T-row matrix V with row-grouped weights (GemmKParams.w_group_rows: row tile -> U_f).
//
//   k_wino_in   x [images, H, W, C1] (| x2 [.., C2]) -> V [16][T][C1 + C2]        T = images (H / 2) (W / 2), packed fp16 arithmetic
//   k_gemm_pq   V [16 T, C] x U [16][Cout][C]        -> M [16][T][Cout]            (fp16, rounded once per product sum as any GEMM output)
//   k_wino_out  M                                    -> y [images, H, W, Cout]     fp32: A^T M A, bias, row bias, alpha, residual, post, activation
//
// V and M live in the caller's workspace (ca_conv3x3_workspace_bytes).  The input transform adds and subtracts pairs of activations:
// packed fp16 arithmetic (two roundings per V element) or, for bf16, fp32 arithmetic rounded per operation.
// Taken where it pays: stride 1, pad 1 (optionally behind a nearest x2 upsampling), even H and W, Cin >= 1280, Cout % 320 == 0, T % 256 == 0,
// at most 16384 tiles (wino_workspace_bytes in ca_gemm.hip).

// dst[f][co][ci] = sum_{kh, kw} G[xi][kh] G[nu][kw] w[co][kh][kw][ci],  f = 4 xi + nu,  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
template <int DT>
__global__ __launch_bounds__(256) void k_pack_w_wino(const u16* __restrict__ w, u16* __restrict__ dst, int cout, int cin) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one (co, ci) each
  if (idx >= (int64_t)cout * cin) return;
  const int ci = (int)(idx % cin);
  const int co = (int)(idx / cin);
  float g[3][3];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) g[kh][kw] = Elem<DT>::to_f(w[((int64_t)co * 9 + kh * 3 + kw) * cin + ci]);
  float t[4][3];  // G g
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    t[0][kw] = g[0][kw];
    t[1][kw] = 0.5f * ((g[0][kw] + g[1][kw]) + g[2][kw]);
    t[2][kw] = 0.5f * ((g[0][kw] - g[1][kw]) + g[2][kw]);
    t[3][kw] = g[2][kw];
  }
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) {
    const float u[4] = {t[xi][0], 0.5f * ((t[xi][0] + t[xi][1]) + t[xi][2]), 0.5f * ((t[xi][0] - t[xi][1]) + t[xi][2]), t[xi][2]};
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) dst[((int64_t)(xi * 4 + nu) * cout + co) * cin + ci] = Elem<DT>::from_f(u[nu]);
  }
}

struct WinoParams {
  const u16* x;
  const u16* x2;
  u16* v;          // [16][T][c]
  const u16* mm;   // [16][T][cout]
  u16* y;
  const float* bias;
  const float* rowbias;
  const u16* res;
  int64_t ld_res, ld_rowbias;
  int images, h, w, c1, c2, cout;  // h, w: the LOGICAL input = output size (twice the stored input with ups)
  int ups;                         // 1: nearest x2 folded into the gather (Upsample3D: F.interpolate + conv, resnet.py:67-81)
  int rows_per_group;
  float alpha, post;
  int act;
};

__device__ __forceinline__ unsigned wino_pk_add(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned wino_pk_sub(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// a +- b on a register of two 16-bit elements: fp16 packed (one rounding per operation); bf16 through fp32 (there is no packed bf16 add)
template <int DT>
__device__ __forceinline__ unsigned wino_add2(unsigned a, unsigned b) {
  if (DT == CA_F16) return wino_pk_add(a, b);
  return pack2<DT>(Elem<DT>::to_f((u16)(a & 0xffffu)) + Elem<DT>::to_f((u16)(b & 0xffffu)), Elem<DT>::to_f((u16)(a >> 16)) + Elem<DT>::to_f((u16)(b >> 16)));
}
template <int DT>
__device__ __forceinline__ unsigned wino_sub2(unsigned a, unsigned b) {
  if (DT == CA_F16) return wino_pk_sub(a, b);
  return pack2<DT>(Elem<DT>::to_f((u16)(a & 0xffffu)) - Elem<DT>::to_f((u16)(b & 0xffffu)), Elem<DT>::to_f((u16)(a >> 16)) - Elem<DT>::to_f((u16)(b >> 16)));
}

// one thread = one (tile, 8-channel chunk): 16 pieces in, 16 pieces out; chunk index fastest (coalesced rows of V)
template <int DT>
__global__ __launch_bounds__(256) void k_wino_in(WinoParams p) {
  const int c = p.c1 + p.c2;
  const int chunks = c >> 3;
  const int th = p.h >> 1, tw = p.w >> 1;
  const int64_t tiles = (int64_t)p.images * th * tw;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= tiles * chunks) return;
  const int ck = (int)(idx % chunks);
  const int64_t t = idx / chunks;
  const int tx = (int)(t % tw);
  const int ty = (int)((t / tw) % th);
  const int img = (int)(t / ((int64_t)tw * th));
  const bool second = ck * 8 >= p.c1;
  const u16* src = second ? p.x2 : p.x;
  const int cs = second ? p.c2 : p.c1;
  const int c0 = second ? ck * 8 - p.c1 : ck * 8;
  u32x4 d[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int hy = 2 * ty - 1 + i;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int wx = 2 * tx - 1 + j;
      const int sh = p.h >> p.ups, sw = p.w >> p.ups;  // stored size
      d[i][j] = (hy >= 0 && hy < p.h && wx >= 0 && wx < p.w) ? ld16(src + (((int64_t)img * sh + (hy >> p.ups)) * sw + (wx >> p.ups)) * cs + c0) : (u32x4){0u, 0u, 0u, 0u};
    }
  }
  // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3), then the same along the columns
  u32x4 r[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      r[0][j][e] = wino_sub2<DT>(d[0][j][e], d[2][j][e]);
      r[1][j][e] = wino_add2<DT>(d[1][j][e], d[2][j][e]);
      r[2][j][e] = wino_sub2<DT>(d[2][j][e], d[1][j][e]);
      r[3][j][e] = wino_sub2<DT>(d[1][j][e], d[3][j][e]);
    }
  const int64_t fstride = tiles * c;
  u16* dst = p.v + t * c + ck * 8;
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) {
    u32x4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[0][e] = wino_sub2<DT>(r[xi][0][e], r[xi][2][e]);
      o[1][e] = wino_add2<DT>(r[xi][1][e], r[xi][2][e]);
      o[2][e] = wino_sub2<DT>(r[xi][2][e], r[xi][1][e]);
      o[3][e] = wino_sub2<DT>(r[xi][1][e], r[xi][3][e]);
    }
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) st16(dst + (int64_t)(xi * 4 + nu) * fstride, o[nu]);
  }
}

// one thread = one (tile, 8-column chunk): 16 pieces of M in, the tile's 2 x 2 output pixels out (fp32 arithmetic, the epilogue of
// gemm_epilogue_direct: bias, row bias, alpha, residual, post scale, activation)
template <int DT>
__global__ __launch_bounds__(256) void k_wino_out(WinoParams p) {
  const int chunks = p.cout >> 3;
  const int th = p.h >> 1, tw = p.w >> 1;
  const int64_t tiles = (int64_t)p.images * th * tw;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= tiles * chunks) return;
  const int ck = (int)(idx % chunks);
  const int64_t t = idx / chunks;
  const int tx = (int)(t % tw);
  const int ty = (int)((t / tw) % th);
  const int img = (int)(t / ((int64_t)tw * th));
  const int64_t fstride = tiles * p.cout;
  const u16* src = p.mm + t * p.cout + ck * 8;
  float s[2][4][8];  // A^T M: (m0 + m1 + m2, m1 - m2 - m3) over xi, per nu
#pragma unroll
  for (int nu = 0; nu < 4; ++nu) {
    float m[4][8];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) unpack8<DT>(ld16(src + (int64_t)(xi * 4 + nu) * fstride), m[xi]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s[0][nu][e] = (m[0][e] + m[1][e]) + m[2][e];
      s[1][nu][e] = (m[1][e] - m[2][e]) - m[3][e];
    }
  }
  const int n0 = ck * 8;
  float bi[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bi[e] = p.bias ? p.bias[n0 + e] : 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t m = ((int64_t)img * p.h + 2 * ty + i) * p.w + 2 * tx + j;  // output row (pixel) index
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = j == 0 ? (s[i][0][e] + s[i][1][e]) + s[i][2][e] : (s[i][1][e] - s[i][2][e]) - s[i][3][e];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += bi[e];
      if (p.rowbias) {
        const float* rb = p.rowbias + (m / p.rows_per_group) * p.ld_rowbias + n0;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rb[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
      if (p.res) {
        float rr[8];
        unpack8<DT>(ld16(p.res + m * p.ld_res + n0), rr);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rr[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= p.post;
      if (p.act != CA_ACT_NONE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = act_f(v[e], p.act);
      }
      st16(p.y + m * p.cout + n0, pack8<DT>(v));
    }
}

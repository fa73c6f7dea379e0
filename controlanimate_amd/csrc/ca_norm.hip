// GroupNorm(+SiLU) and LayerNorm(+positional table) for channels-last activations (gfx950).
// Both are HBM-bound: 16-byte vector accesses, fp32 statistics, wavefront (64-lane) shuffles.
#include "ca_common.h"
#include <stdlib.h>

namespace {

constexpr int GN_MAX_SLOTS = 5;  // 64 lanes * 8 channels * 5 = 2560 channels
constexpr int GN_MAX_C = 64 * 8 * GN_MAX_SLOTS;

// Row chunks per statistics group.  The statistics pass uses ~256-row chunks (its per-block
// reduction is amortised over more rows), the apply pass ~64-row chunks; both go down to 4 rows per
// block for the 8x8 / 16x16 latent levels (a [32, 8, 8, 1280] tensor would otherwise run on 32 blocks).
inline int gn_chunks_rows(int64_t rows_per_stat, int rows_per_chunk) {
  int64_t n = (rows_per_stat + rows_per_chunk - 1) / rows_per_chunk;
  if (n < 16) n = (rows_per_stat + 3) / 4 < 16 ? (rows_per_stat + 3) / 4 : 16;
  if (n < 1) n = 1;
  if (n > 256) n = 256;
  return (int)n;
}
inline int gn_chunks_host(int64_t rows_per_stat) { return gn_chunks_rows(rows_per_stat, 256); }

struct GnParams {
  const u16* x;
  const u16* x2;
  u16* y;
  const float* gamma;
  const float* beta;
  float* partials;
  int c1, c2, groups;
  int64_t rows_per_stat;  // frames_per_stat * hw
  int nchunks;             // chunks of the statistics pass (= partial sums per group)
  int64_t rows_per_chunk;  // rows per block of the kernel being launched
  float eps;
  int act;
  // round 5 (ABI v12): the fused one-launch kernel writes the Winograd F(2x2, 3x3) INPUT TRANSFORM of the normalised activations
  // (V [16][T][C], csrc/ca_conv_wino.h) instead of y: the image is wino_h x wino_w pixels, T = images * (h / 2) * (w / 2)
  u16* wino_v;
  int wino_h, wino_w;
};

// Thread mapping: tx = lane over 16-byte channel chunks, ty = lane over rows; TX = 2^txlog in {8,16,32,64} is the
// choice with C/8 <= TX * slots (slots <= 5) that idles the fewest lanes, the smallest such TX on ties (more rows in
// flight per block): C = 320/640/1280 -> TX = 8/16/32 with 5 slots, C = 128/256/512 (VAE) -> TX = 8/16/32 with
// 2/2/2 slots (a 64-wide mapping idles 75% of the lanes at C = 128).  256/TX rows are in flight per block
// iteration, every thread issuing all its slot loads back to back.
__host__ __device__ inline int gn_txlog(int c8) {
  // (ties in idle lanes, round 6: the mapping with TWO slots if there is one -- C = 128 / 256 / 512: TX = 8 / 16 / 32 -- the kernels' per-thread
  //  arrays are instantiated per slot count, and two slots keep twice the waves per SIMD in flight that four do; else the smallest TX)
  int best = 6, best_waste = 1 << 30, best_slots = 0;
  for (int l = 3; l <= 6; ++l) {
    const int tx = 1 << l, slots = (c8 + tx - 1) / tx;
    if (slots > 5) continue;
    const int waste = slots * tx - c8;
    if (waste < best_waste || (waste == best_waste && best_slots != 2 && slots == 2)) {
      best_waste = waste;
      best = l;
      best_slots = slots;
    }
  }
  return best;
}

// grid: (nchunks, n_stat_groups)
template <int DT, int txlog, int SLOTS = GN_MAX_SLOTS>
__global__ __launch_bounds__(256) void k_gn_stats(GnParams p) {
  __shared__ float ch_s[GN_MAX_C];
  __shared__ float ch_ss[GN_MAX_C];
  const int C = p.c1 + p.c2, C8 = C >> 3;
  constexpr int TX = 1 << txlog, TY = 256 >> txlog;
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x >> txlog;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunk = blockIdx.x, sg = blockIdx.y;
  const int64_t r0 = (int64_t)chunk * p.rows_per_chunk;
  int64_t r1 = r0 + p.rows_per_chunk;
  if (r1 > p.rows_per_stat) r1 = p.rows_per_stat;
  const int64_t base_row = (int64_t)sg * p.rows_per_stat;

  float s[SLOTS][8], ss[SLOTS][8];
#pragma unroll
  for (int k = 0; k < SLOTS; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) s[k][j] = ss[k][j] = 0.f;

  // two row iterations per trip: all 2 x slots loads are issued before the first use (bytes in flight)
  // Rows are dealt to the blocks of a statistics group in TY-row pieces, round-robin: at any moment the
  // resident blocks read ONE moving window of memory (DRAM page locality) instead of gridDim.x distant
  // streams.  Two pieces per trip, all loads issued before the first use.
  for (int64_t piece = chunk;; piece += 2 * (int64_t)gridDim.x) {
    const int64_t ra = piece * TY + ty, rb = (piece + gridDim.x) * TY + ty;
    if (piece * TY >= p.rows_per_stat) break;
    const bool ok[2] = {ra < p.rows_per_stat, rb < p.rows_per_stat};
    const int64_t rows2[2] = {base_row + ra, base_row + rb};
    u32x4 raw[2][SLOTS];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        const int c8 = tx + (k << txlog);
        if (c8 < C8 && ok[h]) {
          const int ch = c8 << 3;
          const int64_t rr = rows2[h];
          raw[h][k] = ld16(ch < p.c1 ? p.x + rr * p.c1 + ch : p.x2 + rr * p.c2 + (ch - p.c1));
        }
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        if (tx + (k << txlog) < C8 && ok[h]) {
          float f[8];
          unpack8<DT>(raw[h][k], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            s[k][j] += f[j];
            ss[k][j] += f[j] * f[j];
          }
        }
      }
  }
  const int cpg = C / p.groups;
  if (cpg >= 8) {
    // A 16-byte chunk (8 channels) overlaps at most two groups: reduce (first part, second part) pairs
    // instead of 8 channels -- 4x fewer shuffles, one barrier.  Fixed order everywhere (deterministic).
    __shared__ float part[4][GN_MAX_C / 8][4];
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int c8 = tx + (k << txlog);
      if ((k << txlog) < C8) {
        const int g0 = (c8 << 3) / cpg;
        const int n0 = (g0 + 1) * cpg - (c8 << 3);  // channels of this chunk that belong to group g0 (>= 8: all)
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (j < n0) {
            v[0] += s[k][j];
            v[1] += ss[k][j];
          } else {
            v[2] += s[k][j];
            v[3] += ss[k][j];
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int off = TX; off < 64; off <<= 1) v[q] += __shfl_xor(v[q], off);
        if (lane < TX && c8 < C8) {
#pragma unroll
          for (int q = 0; q < 4; ++q) part[wave][c8][q] = v[q];
        }
      }
    }
    __syncthreads();
    if (threadIdx.x < p.groups) {
      const int g = threadIdx.x;
      float a = 0.f, b = 0.f;
      for (int c8 = (g * cpg) >> 3; c8 <= ((g + 1) * cpg - 1) >> 3; ++c8) {
        const int g0 = (c8 << 3) / cpg;
        const int q = g0 == g ? 0 : 2;
        for (int w = 0; w < 4; ++w) {
          a += part[w][c8][q];
          b += part[w][c8][q + 1];
        }
      }
      float* out = p.partials + (((int64_t)sg * p.nchunks + chunk) * p.groups + g) * 2;
      out[0] = a;
      out[1] = b;
    }
    return;
  }
  // deterministic reduction: row lanes of one wave by xor-shuffles (fixed tree), then the four waves
  // one after the other through LDS
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    if ((k << txlog) < C8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int off = TX; off < 64; off <<= 1) {
          s[k][j] += __shfl_xor(s[k][j], off);
          ss[k][j] += __shfl_xor(ss[k][j], off);
        }
      }
    }
  }
  for (int w = 0; w < 4; ++w) {
    if (wave == w && lane < TX) {
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        const int c8 = tx + (k << txlog);
        if (c8 < C8) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int ch = (c8 << 3) + j;
            if (w == 0) {
              ch_s[ch] = s[k][j];
              ch_ss[ch] = ss[k][j];
            } else {
              ch_s[ch] += s[k][j];
              ch_ss[ch] += ss[k][j];
            }
          }
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < p.groups) {
    float a = 0.f, b = 0.f;
    for (int j = 0; j < cpg; ++j) {
      a += ch_s[threadIdx.x * cpg + j];
      b += ch_ss[threadIdx.x * cpg + j];
    }
    float* out = p.partials + (((int64_t)sg * p.nchunks + chunk) * p.groups + threadIdx.x) * 2;
    out[0] = a;
    out[1] = b;
  }
}

template <int DT, int txlog, int SLOTS = GN_MAX_SLOTS>
__global__ __launch_bounds__(256) void k_gn_apply(GnParams p) {
  __shared__ double red[8][64][2];
  __shared__ float gm[64], gr[64];
  const int C = p.c1 + p.c2, C8 = C >> 3;
  constexpr int TX = 1 << txlog, TY = 256 >> txlog;
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x >> txlog;
  const int chunk = blockIdx.x, sg = blockIdx.y;
  const int cpg = C / p.groups;
  {
    // partial sums of this statistics group: 256 threads = (slot, group), each slot adds every S-th chunk
    // in fp64, then the slots are added in order (deterministic; 2 dependent loads instead of nchunks)
    const int S = p.groups <= 32 ? 8 : 4;
    const int g = threadIdx.x % (256 / S), slot = threadIdx.x / (256 / S);
    if (g < p.groups) {
      double a = 0.0, b = 0.0;
      const float* part = p.partials + ((int64_t)sg * p.nchunks * p.groups + g) * 2;
      for (int k = slot; k < p.nchunks; k += S) {
        const float2 v = *reinterpret_cast<const float2*>(part + (int64_t)k * p.groups * 2);
        a += (double)v.x;
        b += (double)v.y;
      }
      red[slot][g][0] = a;
      red[slot][g][1] = b;
    }
    __syncthreads();
    if (threadIdx.x < p.groups) {
      double a = 0.0, b = 0.0;
      for (int q = 0; q < S; ++q) {
        a += red[q][threadIdx.x][0];
        b += red[q][threadIdx.x][1];
      }
      const double cnt = (double)p.rows_per_stat * (double)cpg;
      const double mean = a / cnt;
      double var = b / cnt - mean * mean;
      if (var < 0.0) var = 0.0;
      gm[threadIdx.x] = (float)mean;
      gr[threadIdx.x] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
  }
  // this thread's channels are the same for every row: keep their scale/shift in registers
  float rs[SLOTS][8], rh[SLOTS][8];
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const int c8 = tx + (k << txlog);
    if (c8 < C8) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + (c8 << 3)), g1 = *reinterpret_cast<const f32x4*>(p.gamma + (c8 << 3) + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + (c8 << 3)), b1 = *reinterpret_cast<const f32x4*>(p.beta + (c8 << 3) + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int gi = ((c8 << 3) + j) / cpg;
        const float scale = gr[gi] * (j < 4 ? g0[j & 3] : g1[j & 3]);
        rs[k][j] = scale;
        rh[k][j] = (j < 4 ? b0[j & 3] : b1[j & 3]) - gm[gi] * scale;
      }
    }
  }
  const int64_t r0 = (int64_t)chunk * p.rows_per_chunk;
  int64_t r1 = r0 + p.rows_per_chunk;
  if (r1 > p.rows_per_stat) r1 = p.rows_per_stat;
  const int64_t base_row = (int64_t)sg * p.rows_per_stat;
  for (int64_t piece = chunk;; piece += 2 * (int64_t)gridDim.x) {  // same round-robin dealing as the statistics pass
    const int64_t ra = piece * TY + ty, rb = (piece + gridDim.x) * TY + ty;
    if (piece * TY >= p.rows_per_stat) break;
    const bool ok[2] = {ra < p.rows_per_stat, rb < p.rows_per_stat};
    const int64_t rows2[2] = {base_row + ra, base_row + rb};
    u32x4 raw[2][SLOTS];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        const int c8 = tx + (k << txlog);
        if (c8 < C8 && ok[h]) {
          const int ch = c8 << 3;
          const int64_t rr = rows2[h];
          raw[h][k] = ld16(ch < p.c1 ? p.x + rr * p.c1 + ch : p.x2 + rr * p.c2 + (ch - p.c1));
        }
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        const int c8 = tx + (k << txlog);
        if (c8 < C8 && ok[h]) {
          float f[8];
          unpack8<DT>(raw[h][k], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float v = f[j] * rs[k][j] + rh[k][j];
            f[j] = p.act == CA_ACT_SILU ? silu_f(v) : v;
          }
          st16(p.y + rows2[h] * C + (c8 << 3), pack8<DT>(f));
        }
      }
  }
}

// One launch for small statistics groups (the 8x8- and 16x16-latent levels: two launches of ~8 us each for 5-20 MB):
// one block per (statistics group, norm group) keeps that group's rows x channels in registers between the two passes
// -- read once, statistics, normalise, write.  Needs cpg % 8 == 0 (a 16-byte chunk inside one norm group: cpg = 40 / 80)
// and rows * cpg / 8 <= 256 * GNS_MAX chunks.  Deterministic (fixed shuffle tree, ordered LDS pass, fp64 finish).
constexpr int GNS_MAX = 12;
template <int DT, int NCH = GNS_MAX>  // NCH = chunks a thread holds: instantiated for 2 / 5 / 12 (round 6: a [32, 16, 16, 1280] launch needs 5, not the registers of 12)
__global__ __launch_bounds__(256) void k_gn_small(GnParams p) {
  __shared__ double red[4][2];
  __shared__ float mr[2];
  const int C = p.c1 + p.c2;
  const int cpg = C / p.groups, q = cpg >> 3;
  const int g = blockIdx.x, sg = blockIdx.y;
  const int total = (int)p.rows_per_stat * q;
  const int64_t base_row = (int64_t)sg * p.rows_per_stat;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 raw[NCH];
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int id = threadIdx.x + k * 256;
    if (id < total) {
      const int row = id / q, ch = g * cpg + (id - row * q) * 8;
      const int64_t rr = base_row + row;
      raw[k] = ld16(ch < p.c1 ? p.x + rr * p.c1 + ch : p.x2 + rr * p.c2 + (ch - p.c1));
    }
  }
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    if (threadIdx.x + k * 256 < total) {
      float f[8];
      unpack8<DT>(raw[k], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s += f[j];
        ss += f[j] * f[j];
      }
    }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    s += __shfl_xor(s, off);
    ss += __shfl_xor(ss, off);
  }
  if (lane == 0) {
    red[wave][0] = (double)s;
    red[wave][1] = (double)ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double a = red[0][0] + red[1][0] + red[2][0] + red[3][0], b = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    const double cnt = (double)p.rows_per_stat * (double)cpg;
    const double mean = a / cnt;
    double var = b / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[0] = (float)mean;
    mr[1] = (float)(1.0 / sqrt(var + (double)p.eps));
  }
  __syncthreads();
  const float mean = mr[0], rstd = mr[1];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int id = threadIdx.x + k * 256;
    if (id < total) {
      const int row = id / q, ch = g * cpg + (id - row * q) * 8;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + ch), g1 = *reinterpret_cast<const f32x4*>(p.gamma + ch + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + ch), b1 = *reinterpret_cast<const f32x4*>(p.beta + ch + 4);
      float f[8];
      unpack8<DT>(raw[k], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float scale = rstd * (j < 4 ? g0[j & 3] : g1[j & 3]);
        const float v = f[j] * scale + ((j < 4 ? b0[j & 3] : b1[j & 3]) - mean * scale);
        f[j] = p.act == CA_ACT_SILU ? silu_f(v) : v;
      }
      st16(p.y + (base_row + row) * C + ch, pack8<DT>(f));
    }
  }
}

// k_gn_small whose output is the Winograd input transform of GroupNorm(+SiLU)(x) (round 5): the block that owns (norm group g, image)
// has every pixel of its cpg channels on chip, so instead of writing y for a transform kernel to read back (csrc/ca_conv_wino.h:
// k_wino_in), it puts the normalised slab into LDS ([pixel][cpg] fp16, <= 40 KB) and writes B^T d B of every 4 x 4 neighbourhood
// straight into V [16][T][C] -- the convolution behind it (ca_conv_args.x_is_wino_v) starts at its GEMM.  fp16 only (packed adds).
__device__ __forceinline__ unsigned gnw_pk_add(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ unsigned gnw_pk_sub(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
constexpr int GNW_MAX_ROWS = 256, GNW_MAX_CPG = 80;
template <int NCH, int CPG_MAX>  // chunks a thread holds (2 / 5 / 12) and the widest norm group the LDS slab is sized for (40: 20 KB, 80: 40 KB)
__global__ __launch_bounds__(256) void k_gn_small_wino(GnParams p) {
  constexpr int DT = CA_F16;
  __shared__ double red[4][2];
  __shared__ float mr[2];
  __shared__ __attribute__((aligned(16))) u16 slab[GNW_MAX_ROWS * CPG_MAX];
  const int C = p.c1 + p.c2;
  const int cpg = C / p.groups, q = cpg >> 3;
  const int g = blockIdx.x, sg = blockIdx.y;  // (frames_per_stat == 1: statistics group = image)
  const int total = (int)p.rows_per_stat * q;
  const int64_t base_row = (int64_t)sg * p.rows_per_stat;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 raw[NCH];
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int id = threadIdx.x + k * 256;
    if (id < total) {
      const int row = id / q, ch = g * cpg + (id - row * q) * 8;
      const int64_t rr = base_row + row;
      raw[k] = ld16(ch < p.c1 ? p.x + rr * p.c1 + ch : p.x2 + rr * p.c2 + (ch - p.c1));
    }
  }
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    if (threadIdx.x + k * 256 < total) {
      float f[8];
      unpack8<DT>(raw[k], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s += f[j];
        ss += f[j] * f[j];
      }
    }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    s += __shfl_xor(s, off);
    ss += __shfl_xor(ss, off);
  }
  if (lane == 0) {
    red[wave][0] = (double)s;
    red[wave][1] = (double)ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double a = red[0][0] + red[1][0] + red[2][0] + red[3][0], b = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    const double cnt = (double)p.rows_per_stat * (double)cpg;
    const double mean = a / cnt;
    double var = b / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[0] = (float)mean;
    mr[1] = (float)(1.0 / sqrt(var + (double)p.eps));
  }
  __syncthreads();
  const float mean = mr[0], rstd = mr[1];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int id = threadIdx.x + k * 256;
    if (id < total) {
      const int row = id / q, cq = id - row * q, ch = g * cpg + cq * 8;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + ch), g1 = *reinterpret_cast<const f32x4*>(p.gamma + ch + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + ch), b1 = *reinterpret_cast<const f32x4*>(p.beta + ch + 4);
      float f[8];
      unpack8<DT>(raw[k], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float scale = rstd * (j < 4 ? g0[j & 3] : g1[j & 3]);
        const float v = f[j] * scale + ((j < 4 ? b0[j & 3] : b1[j & 3]) - mean * scale);
        f[j] = p.act == CA_ACT_SILU ? silu_f(v) : v;
      }
      st16(slab + (row * q + cq) * 8, pack8<DT>(f));  // (exactly the values k_gn_small stores to y)
    }
  }
  __syncthreads();
  // ---- B^T d B per (tile, chunk): as k_wino_in, reading the slab
  const int th = p.wino_h >> 1, tw = p.wino_w >> 1;
  const int items = th * tw * q;
  const int64_t tiles_total = (int64_t)gridDim.y * th * tw;
  const int64_t fstride = tiles_total * C;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int cq = it % q, t = it / q;
    const int tx = t % tw, ty = t / tw;
    u32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hy = 2 * ty - 1 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int wx = 2 * tx - 1 + j;
        d[i][j] = (hy >= 0 && hy < p.wino_h && wx >= 0 && wx < p.wino_w) ? ld16(slab + ((hy * p.wino_w + wx) * q + cq) * 8) : (u32x4){0u, 0u, 0u, 0u};
      }
    }
    u32x4 r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r[0][j][e] = gnw_pk_sub(d[0][j][e], d[2][j][e]);
        r[1][j][e] = gnw_pk_add(d[1][j][e], d[2][j][e]);
        r[2][j][e] = gnw_pk_sub(d[2][j][e], d[1][j][e]);
        r[3][j][e] = gnw_pk_sub(d[1][j][e], d[3][j][e]);
      }
    u16* dst = p.wino_v + ((int64_t)sg * th * tw + t) * C + g * cpg + cq * 8;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      u32x4 o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[0][e] = gnw_pk_sub(r[xi][0][e], r[xi][2][e]);
        o[1][e] = gnw_pk_add(r[xi][1][e], r[xi][2][e]);
        o[2][e] = gnw_pk_sub(r[xi][2][e], r[xi][1][e]);
        o[3][e] = gnw_pk_sub(r[xi][1][e], r[xi][3][e]);
      }
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) st16(dst + (int64_t)(xi * 4 + nu) * fstride, o[nu]);
    }
  }
}

// The same one-launch scheme for the 10- and 20-channel groups of the 64x64- and 32x32-latent levels, where a 16-byte
// chunk straddles norm groups: a block owns a UNIT of UC = lcm(cpg, 8) channels (40: four or two groups) of one statistics
// group, 1024 threads x up to 20 chunks in registers (64x64 latents: 4096 rows x 5 chunks exactly).  Read once, write
// once: 168 MB instead of 252 MB at [32, 64, 64, 320].  The units of one image sit on one XCD (their 80-byte pieces
// share 128-byte lines).  A chunk touches at most two groups: (first part, second part) sums as in k_gn_stats.
// NCH = chunks a thread holds: 20 (64x64 latents: one block per CU) or 5 (32x32 latents and below, <= 5120 chunks per unit: the block's
// registers then allow two blocks per CU, whose load, arithmetic and store phases overlap -- round 5).
constexpr int GNU_MAX = 20;
template <int DT, int NCH>
__global__ __launch_bounds__(1024, NCH <= 5 ? 2 : 1) void k_gn_unit(GnParams p, int uc, int units) {
  __shared__ float red[16][4][2];
  __shared__ float mr[4][2];
  __shared__ __attribute__((aligned(16))) float sc[128], sh[128];
  const int C = p.c1 + p.c2;
  const int cpg = C / p.groups, q = uc >> 3, ug = uc / cpg;  // chunks per row and unit, groups per unit
  // block -> (unit, statistics group): the `units` units of a statistics group on one XCD (round-robin dispatch)
  const unsigned bid = blockIdx.x;
  const int nstat = gridDim.x / units;
  int unit, sg;
  if (nstat % 8 == 0) {
    unit = (bid >> 3) % units;
    sg = (bid & 7) + 8 * (bid / (8 * units));
  } else {
    unit = bid % units;
    sg = bid / units;
  }
  const int total = (int)p.rows_per_stat * q;
  const int64_t base_row = (int64_t)sg * p.rows_per_stat;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 raw[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int id = threadIdx.x + k * 1024;
    if (id < total) {
      const int row = id / q, ch = unit * uc + (id - row * q) * 8;
      const int64_t rr = base_row + row;
      raw[k] = ld16(ch < p.c1 ? p.x + rr * p.c1 + ch : p.x2 + rr * p.c2 + (ch - p.c1));
    }
  }
  float acc[4][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int id = threadIdx.x + k * 1024;
    if (id < total) {
      const int cq = id % q;
      const int ga = (cq * 8) / cpg, n0 = (ga + 1) * cpg - cq * 8;  // channels of this chunk in group ga (>= 8: all)
      float f[8];
      unpack8<DT>(raw[k], f);
      float lo = 0.f, lo2 = 0.f, hi = 0.f, hi2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < n0) {
          lo += f[j];
          lo2 = fmaf(f[j], f[j], lo2);
        } else {
          hi += f[j];
          hi2 = fmaf(f[j], f[j], hi2);
        }
      }
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        acc[gi][0] += (ga == gi ? lo : 0.f) + (ga + 1 == gi ? hi : 0.f);
        acc[gi][1] += (ga == gi ? lo2 : 0.f) + (ga + 1 == gi ? hi2 : 0.f);
      }
    }
  }
#pragma unroll
  for (int gi = 0; gi < 4; ++gi)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float v = acc[gi][t];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off);
      if (lane == 0) red[wave][gi][t] = v;
    }
  __syncthreads();
  if (threadIdx.x < ug) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 16; ++w) {
      a += (double)red[w][threadIdx.x][0];
      b += (double)red[w][threadIdx.x][1];
    }
    const double cnt = (double)p.rows_per_stat * (double)cpg;
    const double mean = a / cnt;
    double var = b / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[threadIdx.x][0] = (float)mean;
    mr[threadIdx.x][1] = (float)(1.0 / sqrt(var + (double)p.eps));
  }
  __syncthreads();
  if (threadIdx.x < uc) {  // per-channel scale / shift of the unit (register pressure: 20 chunks stay live per thread)
    const int ch = unit * uc + threadIdx.x, gi = threadIdx.x / cpg;
    const float scale = mr[gi][1] * p.gamma[ch];
    sc[threadIdx.x] = scale;
    sh[threadIdx.x] = p.beta[ch] - mr[gi][0] * scale;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int id = threadIdx.x + k * 1024;
    if (id < total) {
      const int row = id / q, cq = id - row * q;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc + cq * 8), s1 = *reinterpret_cast<const f32x4*>(sc + cq * 8 + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh + cq * 8), h1 = *reinterpret_cast<const f32x4*>(sh + cq * 8 + 4);
      float f[8];
      unpack8<DT>(raw[k], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = fmaf(f[j], j < 4 ? s0[j & 3] : s1[j & 3], j < 4 ? h0[j & 3] : h1[j & 3]);
        f[j] = p.act == CA_ACT_SILU ? silu_f(v) : v;
      }
      st16(p.y + (base_row + row) * C + unit * uc + cq * 8, pack8<DT>(f));
    }
  }
}

// unit width for k_gn_unit: lcm(cpg, 8) if the launch qualifies, else 0
inline int gn_unit_channels(const GnParams& p) {
  static const int env = CA_KNOB("CA_GN_FUSED", 1);
  const int C = p.c1 + p.c2, cpg = C / p.groups;
  if (!env || !p.y || cpg % 8 == 0 || cpg < 8) return 0;  // (cpg >= 8: a 16-byte chunk then touches at most two groups)
  int uc = cpg;
  while (uc % 8) uc += cpg;
  if (uc / cpg > 4 || uc > 128 || C % uc != 0 || (p.c2 != 0 && p.c1 % uc != 0)) return 0;
  return p.rows_per_stat * (uc >> 3) <= 1024 * GNU_MAX ? uc : 0;
}

inline bool gn_small_ok(const GnParams& p) {
  static const int env = CA_KNOB("CA_GN_FUSED", 1);  // 0: always the two-kernel path
  const int C = p.c1 + p.c2, cpg = C / p.groups;
  return env && p.y && cpg % 8 == 0 && (p.c2 == 0 || p.c1 % cpg == 0) && p.rows_per_stat * (cpg >> 3) <= 256 * GNS_MAX;
}


int gn_fill(const ca_groupnorm_args* a, GnParams& p, const char* who) {
  CA_REQUIRE(a != nullptr, "%s: null args", who);
  CA_REQUIRE(a->x && a->partials, "%s: null operand", who);
  CA_REQUIRE(a->c1 > 0 && a->c1 % 8 == 0 && a->c2 >= 0 && a->c2 % 8 == 0, "%s: c1=%d c2=%d must be multiples of 8", who, a->c1, a->c2);
  CA_REQUIRE(a->c2 == 0 || a->x2, "%s: x2 missing", who);
  const int C = a->c1 + a->c2;
  CA_REQUIRE(C <= GN_MAX_C, "%s: C=%d exceeds %d", who, C, GN_MAX_C);
  CA_REQUIRE(a->groups > 0 && a->groups <= 64 && C % a->groups == 0, "%s: groups=%d does not divide C=%d", who, a->groups, C);
  CA_REQUIRE(a->frames_per_stat > 0 && a->images % a->frames_per_stat == 0, "%s: frames_per_stat=%d does not divide images=%d", who, a->frames_per_stat, a->images);
  CA_REQUIRE(a->hw > 0, "%s: hw", who);
  CA_REQUIRE(a->dtype == CA_BF16 || a->dtype == CA_F16, "%s: dtype %d", who, a->dtype);
  p.x = (const u16*)a->x;
  p.x2 = (const u16*)a->x2;
  p.y = (u16*)a->y;
  p.gamma = a->gamma;
  p.beta = a->beta;
  p.partials = a->partials;
  p.c1 = a->c1;
  p.c2 = a->c2;
  p.groups = a->groups;
  p.rows_per_stat = (int64_t)a->frames_per_stat * a->hw;
  p.nchunks = gn_chunks_host(p.rows_per_stat);
  p.rows_per_chunk = (p.rows_per_stat + p.nchunks - 1) / p.nchunks;
  p.eps = a->eps;
  p.act = a->act;
  return CA_OK;
}

// ---- LayerNorm ----------------------------------------------------------------------------
constexpr int LN_MAX_SLOTS = 4;  // C <= 2048

struct LnParams {
  const u16* x;
  u16* y;
  const float* gamma;
  const float* beta;
  const float* pos;
  float* stats;  // non-NULL: write (mean, rstd) per row instead of the normalised output
  int64_t rows;
  int c;
  int rows_per_frame, frames;
  float eps;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// One wave normalises LN_RPW rows per iteration: all their loads are issued before the first
// reduction (memory-level parallelism; a single 640-byte row per wave was latency-bound at
// 0.8 TB/s), rows stay packed in registers and are unpacked per pass (sum, variance, output).
constexpr int LN_RPW = 4;

template <int DT, int SLOTS>
__global__ __launch_bounds__(256) void k_layernorm(LnParams p) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int C8 = p.c >> 3;
  const float inv_c = 1.0f / (float)p.c;
  bool cok[SLOTS];
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) cok[k] = lane + 64 * k < C8;
  for (int64_t row0 = wave * LN_RPW; row0 < p.rows; row0 += nwaves * LN_RPW) {
    u32x4 raw[LN_RPW][SLOTS];
#pragma unroll
    for (int r = 0; r < LN_RPW; ++r) {
      const int64_t row = row0 + r;
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        raw[r][k] = (cok[k] && row < p.rows) ? ld16(p.x + row * p.c + ((lane + 64 * k) << 3)) : (u32x4){0u, 0u, 0u, 0u};
      }
    }
    float mean[LN_RPW], rstd[LN_RPW];
#pragma unroll
    for (int r = 0; r < LN_RPW; ++r) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        float f[8];
        unpack8<DT>(raw[r][k], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += f[j];
      }
      mean[r] = s;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
      for (int r = 0; r < LN_RPW; ++r) mean[r] += __shfl_xor(mean[r], off);
    }
#pragma unroll
    for (int r = 0; r < LN_RPW; ++r) {
      mean[r] *= inv_c;
      float q = 0.f;
#pragma unroll
      for (int k = 0; k < SLOTS; ++k) {
        float f[8];
        unpack8<DT>(raw[r][k], f);
        if (cok[k]) {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = f[j] - mean[r];
            q += d * d;
          }
        }
      }
      rstd[r] = q;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
      for (int r = 0; r < LN_RPW; ++r) rstd[r] += __shfl_xor(rstd[r], off);
    }
#pragma unroll
    for (int r = 0; r < LN_RPW; ++r) rstd[r] = rsqrtf(rstd[r] * inv_c + p.eps);
    if (p.stats) {  // statistics for a LayerNorm folded into the following GEMM (ca_gemm_args.ln_stats)
      if (lane == 0) {
#pragma unroll
        for (int r = 0; r < LN_RPW; ++r)
          if (row0 + r < p.rows) *reinterpret_cast<float2*>(p.stats + (row0 + r) * 2) = make_float2(mean[r], rstd[r]);
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      if (!cok[k]) continue;
      const int ch = (lane + 64 * k) << 3;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(p.gamma + ch), g1 = *reinterpret_cast<const f32x4*>(p.gamma + ch + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.beta + ch), b1 = *reinterpret_cast<const f32x4*>(p.beta + ch + 4);
#pragma unroll
      for (int r = 0; r < LN_RPW; ++r) {
        const int64_t row = row0 + r;
        if (row >= p.rows) continue;
        float f[8], o[8];
        unpack8<DT>(raw[r][k], f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = (f[j] - mean[r]) * rstd[r] * g0[j] + b0[j];
          o[j + 4] = (f[j + 4] - mean[r]) * rstd[r] * g1[j] + b1[j];
        }
        if (p.pos) {
          const float* pos = p.pos + ((row / p.rows_per_frame) % p.frames) * p.c + ch;
          const f32x4 p0 = *reinterpret_cast<const f32x4*>(pos), p1 = *reinterpret_cast<const f32x4*>(pos + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] += p0[j];
            o[j + 4] += p1[j];
          }
        }
        st16(p.y + row * p.c + ch, pack8<DT>(o));
      }
    }
  }
}

// Statistics-only pass (LayerNorm folded into the following GEMM).  Lane mapping as in the GroupNorm kernels:
// TX = 2^txlog lanes share a row (C/8 = TX * slots chunks), so a wave holds 64/TX rows at once and a row's sum /
// sum of squares is reduced with log2(TX) shuffle steps -- one pass, 2 rows per lane group per trip, against two
// 6-step reductions per 4 rows in k_layernorm.  var = E[x^2] - mean^2 in fp32 (C <= 2560 activations: the
// cancellation error ~1e-7 * mean^2 / var is far below the fp16 rounding of the GEMM operands).
template <int DT, int txlog>
__global__ __launch_bounds__(256) void k_ln_stats(LnParams p) {
  constexpr int TX = 1 << txlog, TY = 256 >> txlog;
  const int C8 = p.c >> 3;
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x >> txlog;
  const float inv_c = 1.0f / (float)p.c;
  for (int64_t row0 = (int64_t)blockIdx.x * 2 * TY; row0 < p.rows; row0 += (int64_t)gridDim.x * 2 * TY) {
    const int64_t rows2[2] = {row0 + ty, row0 + TY + ty};
    u32x4 raw[2][GN_MAX_SLOTS];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int k = 0; k < GN_MAX_SLOTS; ++k) {
        const int c8 = tx + (k << txlog);
        raw[h][k] = (c8 < C8 && rows2[h] < p.rows) ? ld16(p.x + rows2[h] * p.c + (c8 << 3)) : (u32x4){0u, 0u, 0u, 0u};
      }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int k = 0; k < GN_MAX_SLOTS; ++k) {
        if ((k << txlog) < C8) {
          float f[8];
          unpack8<DT>(raw[h][k], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            s += f[j];
            ss += f[j] * f[j];
          }
        }
      }
#pragma unroll
      for (int off = 1; off < TX; off <<= 1) {
        s += __shfl_xor(s, off);
        ss += __shfl_xor(ss, off);
      }
      if (tx == 0 && rows2[h] < p.rows) {
        const float mean = s * inv_c;
        const float var = fmaxf(ss * inv_c - mean * mean, 0.f);
        *reinterpret_cast<float2*>(p.stats + rows2[h] * 2) = make_float2(mean, rsqrtf(var + p.eps));
      }
    }
  }
}

template <int DT>
void launch_layernorm(const LnParams& p, hipStream_t st) {
  if (p.stats && p.c <= GN_MAX_C) {
    const int txlog = gn_txlog(p.c >> 3);
    const int ty = 256 >> txlog;
    int64_t blocks = (p.rows + 2 * ty - 1) / (2 * ty);
    if (blocks > 8192) blocks = 8192;
    dim3 grid((unsigned)blocks), block(256);
    switch (txlog) {
      case 3: hipLaunchKernelGGL((k_ln_stats<DT, 3>), grid, block, 0, st, p); break;
      case 4: hipLaunchKernelGGL((k_ln_stats<DT, 4>), grid, block, 0, st, p); break;
      case 5: hipLaunchKernelGGL((k_ln_stats<DT, 5>), grid, block, 0, st, p); break;
      default: hipLaunchKernelGGL((k_ln_stats<DT, 6>), grid, block, 0, st, p); break;
    }
    return;
  }
  int64_t waves = (p.rows + LN_RPW - 1) / LN_RPW;
  int64_t blocks = (waves + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  const int slots = (p.c / 8 + 63) / 64;
  dim3 grid((unsigned)blocks), block(256);
  switch (slots) {
    case 1: hipLaunchKernelGGL((k_layernorm<DT, 1>), grid, block, 0, st, p); break;
    case 2: hipLaunchKernelGGL((k_layernorm<DT, 2>), grid, block, 0, st, p); break;
    case 3: hipLaunchKernelGGL((k_layernorm<DT, 3>), grid, block, 0, st, p); break;
    default: hipLaunchKernelGGL((k_layernorm<DT, 4>), grid, block, 0, st, p); break;
  }
}

}  // namespace

extern "C" int64_t ca_groupnorm_partials_floats(int32_t images, int32_t hw, int32_t frames_per_stat, int32_t groups) {
  if (images <= 0 || hw <= 0 || frames_per_stat <= 0 || groups <= 0) return 0;
  const int64_t nstat = images / frames_per_stat;
  return nstat * gn_chunks_host((int64_t)frames_per_stat * hw) * groups * 2;
}

// The kernels keep per-thread arrays of SLOTS channel chunks (scale / shift, running sums, two rows of raw data): instantiated for the slot
// counts the mapping produces -- 2 (C = 128: the VAE's 512x512 level), 4 (256 / 512 / 960 / 1920) and 5 (320 / 640 / 1280 / 2560) -- so that a
// narrow tensor does not carry the registers of five slots (round 6: 140 -> ~70 VGPRs at C = 128, twice the waves per SIMD in flight).
template <int DT, int L>
void launch_gn_l(bool apply, int slots, const GnParams& p, dim3 grid, hipStream_t st) {
  if (slots <= 2) {
    if (apply) hipLaunchKernelGGL((k_gn_apply<DT, L, 2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_gn_stats<DT, L, 2>), grid, dim3(256), 0, st, p);
  } else if (slots <= 4) {
    if (apply) hipLaunchKernelGGL((k_gn_apply<DT, L, 4>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_gn_stats<DT, L, 4>), grid, dim3(256), 0, st, p);
  } else {
    if (apply) hipLaunchKernelGGL((k_gn_apply<DT, L, 5>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_gn_stats<DT, L, 5>), grid, dim3(256), 0, st, p);
  }
}

template <int DT>
void launch_gn(bool apply, const GnParams& p, dim3 grid, hipStream_t st) {
  const int c8 = (p.c1 + p.c2) >> 3;
  const int txlog = gn_txlog(c8);
  const int slots = (c8 + (1 << txlog) - 1) >> txlog;
  switch (txlog) {
    case 3: launch_gn_l<DT, 3>(apply, slots, p, grid, st); break;
    case 4: launch_gn_l<DT, 4>(apply, slots, p, grid, st); break;
    case 5: launch_gn_l<DT, 5>(apply, slots, p, grid, st); break;
    default: launch_gn_l<DT, 6>(apply, slots, p, grid, st);
  }
}

extern "C" int ca_groupnorm_stats(const ca_groupnorm_args* a, void* stream) {
  GnParams p{};
  int rc = gn_fill(a, p, "ca_groupnorm_stats");
  if (rc) return rc;
  dim3 grid(p.nchunks, a->images / a->frames_per_stat);
  if (a->dtype == CA_BF16) launch_gn<CA_BF16>(false, p, grid, (hipStream_t)stream);
  else launch_gn<CA_F16>(false, p, grid, (hipStream_t)stream);
  CA_CHECK_LAUNCH("ca_groupnorm_stats");
  return CA_OK;
}

extern "C" int ca_groupnorm_apply(const ca_groupnorm_args* a, void* stream) {
  GnParams p{};
  int rc = gn_fill(a, p, "ca_groupnorm_apply");
  if (rc) return rc;
  CA_REQUIRE(a->y && a->gamma && a->beta, "ca_groupnorm_apply: null operand");
  const int achunks = gn_chunks_rows(p.rows_per_stat, p.rows_per_stat >= 2048 ? 128 : 64);
  p.rows_per_chunk = (p.rows_per_stat + achunks - 1) / achunks;
  dim3 grid(achunks, a->images / a->frames_per_stat);
  if (a->dtype == CA_BF16) launch_gn<CA_BF16>(true, p, grid, (hipStream_t)stream);
  else launch_gn<CA_F16>(true, p, grid, (hipStream_t)stream);
  CA_CHECK_LAUNCH("ca_groupnorm_apply");
  return CA_OK;
}

// 1 if ca_groupnorm can write the Winograd input transform instead of y for these arguments (wino_v / wino_h / wino_w set): the
// one-launch kernel's shapes with per-image statistics, fp16, an even wino_h x wino_w = hw image of at most 256 pixels, <= 80 channels
// per norm group.  No launch, no device access.
extern "C" int ca_groupnorm_wino_supported(const ca_groupnorm_args* a) {
  if (!a || !a->x || !a->gamma || !a->beta || !a->wino_v || a->dtype != CA_F16 || a->frames_per_stat != 1) return 0;
  if (a->images <= 0 || a->c1 <= 0 || a->c1 % 8 || a->c2 < 0 || a->c2 % 8 || (a->c2 && !a->x2) || a->groups <= 0) return 0;
  const int C = a->c1 + a->c2;
  if (C % a->groups) return 0;
  const int cpg = C / a->groups;
  if (cpg % 8 || cpg > GNW_MAX_CPG || (a->c2 != 0 && a->c1 % cpg != 0)) return 0;
  if (a->wino_h < 2 || a->wino_w < 2 || (a->wino_h & 1) || (a->wino_w & 1) || a->wino_h * a->wino_w != a->hw || a->hw > GNW_MAX_ROWS) return 0;
  if ((int64_t)a->hw * (cpg >> 3) > 256 * GNS_MAX) return 0;
  if ((((uintptr_t)a->wino_v | (uintptr_t)a->x | (uintptr_t)a->x2) & 15) != 0) return 0;
  return 1;
}

// GroupNorm in one call: the fused single-launch kernel where a statistics group fits a block's registers, otherwise
// statistics + apply (needs `partials`).
extern "C" int ca_groupnorm(const ca_groupnorm_args* a, void* stream) {
  GnParams p{};
  if (a && a->wino_v) {  // ABI v12: the Winograd input transform of the normalised activations instead of y
    CA_REQUIRE(ca_groupnorm_wino_supported(a), "ca_groupnorm: wino_v with arguments outside the fused form (fp16, per-image statistics, even image of <= 256 pixels, "
                                                "<= 80 channels per group in whole 8-channel chunks): ask ca_groupnorm_wino_supported() first");
    p.x = (const u16*)a->x;
    p.x2 = (const u16*)a->x2;
    p.gamma = a->gamma;
    p.beta = a->beta;
    p.c1 = a->c1, p.c2 = a->c2, p.groups = a->groups;
    p.rows_per_stat = a->hw;
    p.eps = a->eps;
    p.act = a->act;
    p.wino_v = (u16*)a->wino_v;
    p.wino_h = a->wino_h, p.wino_w = a->wino_w;
    {
      const int cpg_w = (a->c1 + a->c2) / a->groups;
      const int64_t per_thread = ((int64_t)a->hw * (cpg_w >> 3) + 255) / 256;
      const dim3 grid_w(a->groups, a->images);
      hipStream_t st_w = (hipStream_t)stream;
#define CA_GNW_CASE(N)                                                                                  \
  if (cpg_w <= 40) hipLaunchKernelGGL((k_gn_small_wino<N, 40>), grid_w, dim3(256), 0, st_w, p);         \
  else hipLaunchKernelGGL((k_gn_small_wino<N, GNW_MAX_CPG>), grid_w, dim3(256), 0, st_w, p);
      if (per_thread <= 2) { CA_GNW_CASE(2) }
      else if (per_thread <= 5) { CA_GNW_CASE(5) }
      else { CA_GNW_CASE(GNS_MAX) }
#undef CA_GNW_CASE
    }
    CA_CHECK_LAUNCH("ca_groupnorm(wino)");
    return CA_OK;
  }
  int rc = gn_fill(a, p, "ca_groupnorm");
  if (rc) return rc;
  CA_REQUIRE(a->y && a->gamma && a->beta, "ca_groupnorm: null operand");
  if (gn_small_ok(p)) {
    const dim3 grid(p.groups, a->images / a->frames_per_stat);
    const int64_t per_thread = (p.rows_per_stat * ((p.c1 + p.c2) / p.groups >> 3) + 255) / 256;
#define CA_GNS_CASE(N)                                                                                                     \
  if (a->dtype == CA_BF16) hipLaunchKernelGGL((k_gn_small<CA_BF16, N>), grid, dim3(256), 0, (hipStream_t)stream, p);       \
  else hipLaunchKernelGGL((k_gn_small<CA_F16, N>), grid, dim3(256), 0, (hipStream_t)stream, p);
    if (per_thread <= 2) { CA_GNS_CASE(2) }
    else if (per_thread <= 5) { CA_GNS_CASE(5) }
    else { CA_GNS_CASE(GNS_MAX) }
#undef CA_GNS_CASE
    CA_CHECK_LAUNCH("ca_groupnorm");
    return CA_OK;
  }
  if (const int uc = gn_unit_channels(p)) {
    const int units = (p.c1 + p.c2) / uc;
    const dim3 grid(units * (a->images / a->frames_per_stat));
    static const int gnu_small = CA_KNOB("CA_GN_UNIT_SMALL", 1);  // (experiments: 0 = the 20-chunk instantiation for every size)
    const bool few = gnu_small && (int64_t)p.rows_per_stat * (uc >> 3) <= 1024 * 5;
    if (a->dtype == CA_BF16) {
      if (few) hipLaunchKernelGGL((k_gn_unit<CA_BF16, 5>), grid, dim3(1024), 0, (hipStream_t)stream, p, uc, units);
      else hipLaunchKernelGGL((k_gn_unit<CA_BF16, GNU_MAX>), grid, dim3(1024), 0, (hipStream_t)stream, p, uc, units);
    } else {
      if (few) hipLaunchKernelGGL((k_gn_unit<CA_F16, 5>), grid, dim3(1024), 0, (hipStream_t)stream, p, uc, units);
      else hipLaunchKernelGGL((k_gn_unit<CA_F16, GNU_MAX>), grid, dim3(1024), 0, (hipStream_t)stream, p, uc, units);
    }
    CA_CHECK_LAUNCH("ca_groupnorm");
    return CA_OK;
  }
  rc = ca_groupnorm_stats(a, stream);
  if (rc) return rc;
  return ca_groupnorm_apply(a, stream);
}

extern "C" int ca_layernorm(const ca_layernorm_args* a, void* stream) {
  CA_REQUIRE(a != nullptr, "ca_layernorm: null args");
  CA_REQUIRE(a->x && (a->stats || (a->y && a->gamma && a->beta)), "ca_layernorm: null operand");
  CA_REQUIRE(!a->stats || !a->pos, "ca_layernorm: stats mode has no positional table");
  CA_REQUIRE(a->rows > 0, "ca_layernorm: rows");
  CA_REQUIRE(a->c > 0 && a->c % 8 == 0 && a->c <= 64 * 8 * LN_MAX_SLOTS, "ca_layernorm: C=%d must be a multiple of 8 and <= %d", a->c, 64 * 8 * LN_MAX_SLOTS);
  CA_REQUIRE(!a->pos || (a->rows_per_frame > 0 && a->frames > 0), "ca_layernorm: pos needs rows_per_frame/frames");
  CA_REQUIRE(a->dtype == CA_BF16 || a->dtype == CA_F16, "ca_layernorm: dtype %d", a->dtype);
  LnParams p{};
  p.x = (const u16*)a->x;
  p.y = (u16*)a->y;
  p.gamma = a->gamma;
  p.beta = a->beta;
  p.pos = a->pos;
  p.stats = a->stats;
  p.rows = a->rows;
  p.c = a->c;
  p.rows_per_frame = a->rows_per_frame > 0 ? a->rows_per_frame : 1;
  p.frames = a->frames > 0 ? a->frames : 1;
  p.eps = a->eps;
  if (a->dtype == CA_BF16) launch_layernorm<CA_BF16>(p, (hipStream_t)stream);
  else launch_layernorm<CA_F16>(p, (hipStream_t)stream);
  CA_CHECK_LAUNCH("ca_layernorm");
  return CA_OK;
}

// ---- LayerNorm statistics from the partial sums a producing GEMM left (ca_gemm_args.row_sums_out): (mean, rstd) per row.
// Same arithmetic, in the same order, as the consuming epilogues that finish the sums themselves (ca_gemm_core.h / ca_gemm_ps.h).
namespace {
__global__ __launch_bounds__(256) void k_ln_finish_sums(const float* __restrict__ sums, int parts, int64_t rows, float inv_k, float eps, float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float a = 0.f, q = 0.f;
  for (int i = 0; i < parts; ++i) {
    const float2 v = *reinterpret_cast<const float2*>(sums + (r * parts + i) * 2);
    a = a + v.x;
    q = q + v.y;
  }
  const float mean = a * inv_k;
  *reinterpret_cast<float2*>(out + r * 2) = make_float2(mean, rsqrtf(fmaxf(q * inv_k - mean * mean, 0.f) + eps));
}
}  // namespace

extern "C" int ca_ln_finish_sums(const float* sums, int parts, int64_t rows, int k, float eps, float* mean_rstd, void* stream) {
  CA_REQUIRE(sums && mean_rstd, "ca_ln_finish_sums: null operand");
  CA_REQUIRE(parts >= 1 && parts <= 64 && rows > 0 && k > 0, "ca_ln_finish_sums: parts=%d rows=%lld k=%d", parts, (long long)rows, k);
  hipLaunchKernelGGL(k_ln_finish_sums, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sums, parts, rows, 1.f / (float)k, eps, mean_rstd);
  CA_CHECK_LAUNCH("ca_ln_finish_sums");
  return CA_OK;
}

// MFMA GEMM / implicit-GEMM 3x3 convolution core for gfx950.
//
//   C[M,N] = epilogue(A[M,K] * W[N,K]^T),  A dense (linear / 1x1 conv) or gathered on the fly
//   from an NHWC activation (3x3 conv, pad 1, stride 1/2, optional nearest-x2 upsample and
//   two-source channel concat).  Both operands are K-contiguous, so MFMA fragments are plain
//   16-byte reads.
//
// Tiling: 256 threads = 4 waves, block tile BM x BN x 64, double-buffered LDS, one barrier per K
// tile.  Two staging variants share the MFMA loop and the epilogue:
//   k_gemm_dma  global->LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): no VGPR round trip, no
//               ds_write pass; out-of-range lanes (conv halo, M/N tails) are given an offset beyond
//               the buffer descriptor's size and the hardware writes zeros (probed on MI355X,
//               tools/probe_glds.hip).  The DMA writes lane-linear, so the bank swizzle is applied
//               to the per-lane SOURCE chunk and again on the fragment read (same involution).
//               Needs channel counts that are multiples of 64 (every UNet/ControlNet layer but conv_in).
//   k_gemm      register-staged copies issued one tile ahead; handles any multiple-of-8 shape.
// LDS rows are 128 B (64 elements);
// the 16-byte chunk index is XOR-swizzled with (row>>1)&7 so that every ds_read_b128 lane group
// of a fragment read hits 16 distinct 16-B slots (MI355X_MICROARCH.md, LDS table).
// MFMA is issued with swapped operands (mfma(Wfrag, Afrag)) so each lane ends up holding 4
// consecutive output columns of one output row -> 8-byte epilogue stores.

#include "ca_gemm_core.h"

namespace {
using namespace ca_gemm_detail;
#include "ca_conv_wino.h"

template <int DT, int BM, int BN, int WAVES_M, int WAVES_N, int MODE>
__global__ __launch_bounds__(256) void k_gemm(GemmKParams p) {
  constexpr int TM = BM / WAVES_M / 16;
  constexpr int TN = BN / WAVES_N / 16;
  constexpr int AI = BM / 32;  // A rows per loader thread
  constexpr int BI = BN / 32;
  static_assert(2 * (BM + BN) * BK >= BM * (BN + 8), "epilogue staging must fit");
  __shared__ __attribute__((aligned(16))) u16 smem[2 * (BM + BN) * BK];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int g = lane >> 4, l15 = lane & 15;

  const int tiles_n = (p.n + BN - 1) / BN;
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  int tile_m, tile_n;
  tile_coords(bid, (p.m + BM - 1) / BM, tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- loader setup -------------------------------------------------------------------
  const int lr = tid >> 3;  // 0..31
  const int lc = tid & 7;   // 16-byte chunk within the 64-wide K tile
  const int kc = p.c1 + p.c2;
  const int64_t wld = (int64_t)p.taps * kc;

  // per-row state of the A loader
  int a_img[AI], a_ho[AI], a_wo[AI];
  bool a_ok[AI];
  int64_t a_rowoff[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    int m = m0 + lr + 32 * i;
    a_ok[i] = m < p.m;
    if (MODE == 1) {
      int mm = a_ok[i] ? m : 0;
      int hw = p.hout * p.wout;
      a_img[i] = mm / hw;
      int rem = mm - a_img[i] * hw;
      a_ho[i] = rem / p.wout;
      a_wo[i] = rem - a_ho[i] * p.wout;
      a_rowoff[i] = 0;
    } else {
      a_img[i] = a_ho[i] = a_wo[i] = 0;
      a_rowoff[i] = (int64_t)m;
    }
  }
  bool b_ok[BI];
  int64_t b_rowoff[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    int n = n0 + lr + 32 * i;
    b_ok[i] = n < p.n;
    b_rowoff[i] = (int64_t)n * wld;
  }

  u32x4 ra[AI], rb[BI];
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  auto load_tile = [&](int t) {
    int tap, cc;
    k_tile_split(p, t, p.kc_tiles, tap, cc);
    int ci = cc * BK + lc * 8;
    bool cok = ci < kc;
    // weights
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      rb[i] = (cok && b_ok[i]) ? ld16(p.w + b_rowoff[i] + (int64_t)tap * kc + ci) : zero4;
    }
    // activations
    const bool src2 = ci >= p.c1;
    const u16* base = src2 ? p.a2 : p.a;
    const int cs = src2 ? p.c2 : p.c1;
    const int cio = src2 ? ci - p.c1 : ci;
    if (MODE == 1) {
      int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        int hi = a_ho[i] * p.stride + kh - p.pad_lo;
        int wi = a_wo[i] * p.stride + kw - p.pad_lo;
        bool ok = cok && a_ok[i] && hi >= 0 && wi >= 0 && hi < (p.hin << p.ups) && wi < (p.win << p.ups);
        int hs = hi >> p.ups, ws = wi >> p.ups;
        int64_t pix = ((int64_t)a_img[i] * p.hin + hs) * p.win + ws;
        ra[i] = ok ? ld16(base + pix * cs + cio) : zero4;
      }
    } else {
      const int64_t ld = src2 ? p.lda2 : p.lda;
#pragma unroll
      for (int i = 0; i < AI; ++i) {
        ra[i] = (cok && a_ok[i]) ? ld16(base + a_rowoff[i] * ld + cio) : zero4;
      }
    }
  };
  auto store_tile = [&](int buf) {
    u16* sa = smem + buf * (BM + BN) * BK;
    u16* sb = sa + BM * BK;
#pragma unroll
    for (int i = 0; i < AI; ++i) st16(sa + lds_off(lr + 32 * i, lc), ra[i]);
#pragma unroll
    for (int i = 0; i < BI; ++i) st16(sb + lds_off(lr + 32 * i, lc), rb[i]);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = p.taps * p.kc_tiles;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) load_tile(t + 1);
    const u16* sa = smem + buf * (BM + BN) * BK;
    const u16* sb = sa + BM * BK;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = ld16(sa + lds_off(wm * TM * 16 + i * 16 + l15, s * 4 + g));
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = ld16(sb + lds_off(wn * TN * 16 + j * 16 + l15, s * 4 + g));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Elem<DT>::mfma(fb[j], fa[i], acc[i][j]);
    }
    if (t + 1 < nt) store_tile(buf ^ 1);
    __syncthreads();
  }

  gemm_epilogue<DT, BM, BN, TM, TN>(p, acc, smem, m0, n0, wm, wn, l15, g, tid);
}

// ---- LDS-DMA variant ------------------------------------------------------------------------

// NBUF = 2: tile t+1 is issued before the MFMA phase of tile t, `__syncthreads()` (which hipcc
//           precedes with vmcnt(0)) once per tile.
// NBUF >= 3: tiles are issued NBUF - 1 ahead into a ring; a wave waits with a COUNTED
//           `s_waitcnt vmcnt(per-tile DMA count x younger tiles)` (tile t landed, the younger ones may still be in
//           flight), then a raw s_barrier: DMA transfers stay in flight across barriers
//           (cdna_hip_programming.md "Pipelining across barriers").  All LDS is one array.
template <int DT, int BM, int BN, int WAVES_M, int WAVES_N, int MODE, int NBUF, int KT = 64>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64,
                             ((BM / WAVES_M) * (BN / WAVES_N) > 64 * 80 ? 2 : (NBUF * KT == 64 ? (BN > 128 ? 3 : 4) : (NBUF * (BM + BN) * KT * 2 > 80 * 1024 ? 1 : 2))) * 4 / (WAVES_M * WAVES_N))
void k_gemm_dma(GemmKParams p) {
  // KT = K elements per LDS stage: 64 (one 128-byte row per tile row) or 32 with NBUF = 2 -- the same 32 KB
  // as one 64-wide stage, so 4 blocks still share a CU, but each block also prefetches its own next stage
  static_assert(KT == 64 || KT == 32, "k tile");
  constexpr int CPR = KT / 8;        // 16-byte chunks per LDS row
  constexpr int RPI = 64 / CPR;      // tile rows one wave-wide DMA instruction covers
  constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;  // 4 waves (128-row tiles) or 8 waves (256x128 tiles)
  constexpr int TM = BM / WAVES_M / 16;
  constexpr int TN = BN / WAVES_N / 16;
  constexpr int AG = BM / RPI / NW;  // DMA instructions per wave per stage (A)
  constexpr int BG = BN / RPI / NW;  // (W)
  // the LDS-staged epilogue needs BM x (BN + 8) elements: more than ONE 128x128x64 stage
  constexpr int SMEM_ELEMS = NBUF * (BM + BN) * KT > BM * (BN + 8) ? NBUF * (BM + BN) * KT : BM * (BN + 8);
  // XOR swizzle of the chunk index: conflict-free ds_read_b128 for the hardware's 16-lane groups
  // (128-byte rows: (row>>1)&7; 64-byte rows: the map [0,3,2,1][(row>>2)&3] = (-(row>>2))&3)
  auto swz = [](int row) { return KT == 64 ? ((row >> 1) & 7) : ((-(row >> 2)) & 3); };
  auto lds_at = [&](int row, int chunk) { return row * KT + ((chunk ^ swz(row)) << 3); };
  __shared__ __attribute__((aligned(16))) u16 smem[SMEM_ELEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;
  const int g = lane >> 4, l15 = lane & 15;

  const int tiles_n = (p.n + BN - 1) / BN;
  const int tiles_m = (p.m + BM - 1) / BM;
  unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  int split = 0;
  if (p.splits > 1) {  // consecutive ids = the tiles of ONE K range (they share the weight slices)
    split = bid / (unsigned)(tiles_m * tiles_n);
    bid -= split * (unsigned)(tiles_m * tiles_n);
  }
  int tile_m, tile_n;
  tile_coords(bid, tiles_m, tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, p.a2 ? p.a2_bytes : p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  const int r8 = lane / CPR, cp = lane % CPR;  // row inside a DMA row group, LDS chunk position
  const int kc = p.c1 + p.c2;
  const int kct = kc / KT;  // K tiles per tap (the DMA path requires kc % 64 == 0)
  const unsigned wld = (unsigned)(p.taps * kc);

  // per staged row: swizzled source chunk, and either a byte offset (dense / weights) or pixel coords (conv)
  int a_chunk[AG], a_img[AG], a_ho[AG], a_wo[AG];
  unsigned a_off1[AG], a_off2[AG];
  bool a_ok[AG];
#pragma unroll
  for (int i = 0; i < AG; ++i) {
    const int row = (wid * AG + i) * RPI + r8;
    a_chunk[i] = cp ^ swz(row);
    const int m = m0 + row;
    a_ok[i] = m < p.m;
    const int mm = a_ok[i] ? m : p.m - 1;
    if (MODE == 1) {
      const int hw = p.hout * p.wout;
      a_img[i] = mm / hw;
      const int rem = mm - a_img[i] * hw;
      a_ho[i] = rem / p.wout;
      a_wo[i] = rem - a_ho[i] * p.wout;
      a_off1[i] = a_off2[i] = 0;
    } else {
      a_img[i] = a_ho[i] = a_wo[i] = 0;
      a_off1[i] = (unsigned)((int64_t)mm * p.lda * 2);
      a_off2[i] = (unsigned)((int64_t)mm * p.lda2 * 2);
    }
  }
  int b_chunk[BG];
  unsigned b_off[BG];
#pragma unroll
  for (int j = 0; j < BG; ++j) {
    const int row = (wid * BG + j) * RPI + r8;
    b_chunk[j] = cp ^ swz(row);
    int n = n0 + row;
    if (n >= p.n) n = p.n - 1;  // clamped rows feed accumulators that are never stored
    b_off[j] = (unsigned)n * wld * 2u;
  }

  auto stage = [&](int t, int buf) {
    u16* sa = smem + buf * (BM + BN) * KT;
    u16* sb = sa + BM * KT;
    int tap, cc;
    k_tile_split(p, t, kct, tap, cc);
    const int c0 = cc * KT;            // first channel of this K tile (tile-uniform)
    const bool src2 = c0 >= p.c1;      // c1 % 64 == 0 => a tile never straddles the two sources
    const int cs = src2 ? p.c2 : p.c1;
    const int cbase = src2 ? c0 - p.c1 : c0;
#pragma unroll
    for (int j = 0; j < BG; ++j) {
      const unsigned off = b_off[j] + (unsigned)(tap * kc + c0 + b_chunk[j] * 8) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(sb + (wid * BG + j) * RPI * KT), 16, off, 0, 0, 0);
    }
    if (MODE == 1) {
      const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
      for (int i = 0; i < AG; ++i) {
        const int hi = a_ho[i] * p.stride + kh - p.pad_lo;
        const int wi = a_wo[i] * p.stride + kw - p.pad_lo;
        const bool ok = a_ok[i] && hi >= 0 && wi >= 0 && hi < (p.hin << p.ups) && wi < (p.win << p.ups);
        const int pix = (a_img[i] * p.hin + (hi >> p.ups)) * p.win + (wi >> p.ups);
        const unsigned off = ok ? ((unsigned)pix * (unsigned)cs + (unsigned)(cbase + a_chunk[i] * 8)) * 2u : DMA_OOB;
        void* dst = sa + (wid * AG + i) * RPI * KT;
        if (src2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < AG; ++i) {
        const unsigned off = (src2 ? a_off2[i] : a_off1[i]) + (unsigned)(cbase + a_chunk[i] * 8) * 2u;
        void* dst = sa + (wid * AG + i) * RPI * KT;
        if (src2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt_all = p.taps * kct;
  const int t_first = p.splits > 1 ? (int)((int64_t)nt_all * split / p.splits) : 0;
  const int nt = (p.splits > 1 ? (int)((int64_t)nt_all * (split + 1) / p.splits) : nt_all) - t_first;
  auto compute = [&](int buf) {
    const u16* sa = smem + buf * (BM + BN) * KT;
    const u16* sb = sa + BM * KT;
    if (CA_GEMM_ABLATE == 1) return;
#pragma unroll
    for (int s = 0; s < KT / 32; ++s) {
      u32x4 fa[TM], fb[TN];
      if (CA_GEMM_ABLATE == 4) {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = (u32x4){(unsigned)tid, 1u, 2u, 3u};
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = (u32x4){(unsigned)lane, 5u, 6u, 7u};
      } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = ld16(sa + lds_at(wm * TM * 16 + i * 16 + l15, s * 4 + g));
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = ld16(sb + lds_at(wn * TN * 16 + j * 16 + l15, s * 4 + g));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Elem<DT>::mfma(fb[j], fa[i], acc[i][j]);
    }
  };
  if (NBUF == 1) {
    // single LDS buffer (32 KB for 128x128): two barriers per tile, but 3 blocks per CU -- the
    // other resident blocks' MFMA phases cover this block's transfer latency
    for (int t = 0; t < nt; ++t) {
      if (CA_GEMM_ABLATE < 2 || t == 0) stage(t_first + t, 0);
      if (CA_GEMM_ABLATE != 5 || t == 0) __syncthreads();
      compute(0);
      if (CA_GEMM_ABLATE != 3 && CA_GEMM_ABLATE != 5) __syncthreads();
    }
  } else if (NBUF == 2) {
    stage(t_first, 0);
    __syncthreads();  // hipcc drains the LDS-DMA queue (vmcnt(0)) ahead of the barrier
    for (int t = 0; t < nt; ++t) {
      const int buf = t & 1;
      if (t + 1 < nt) stage(t_first + t + 1, buf ^ 1);
      compute(buf);
      __syncthreads();
    }
  } else {
    // NBUF >= 3 (round 4): a ring with NBUF - 1 tiles in flight, for the launches whose K loop is a chain of DMA round trips with
    // almost nothing to compute per tile (M = 2048: the 8x8-latent level -- 8 MFMAs per wave and K tile against ~1.2 us per
    // round trip).  With two stages ONE tile is in flight while the previous one is computed; here tile t is awaited with a
    // COUNTED vmcnt (the pieces of the min(NBUF - 2, tiles left) younger tiles stay outstanding), one raw barrier per tile
    // publishes it, and tile t + NBUF - 1 goes into the slot whose reads the same barrier has just retired.
    constexpr int PER = AG + BG;  // DMA instructions per wave and stage
    static_assert((NBUF - 2) * PER < 64, "vmcnt");
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
      if (s < nt) stage(t_first + s, s);
    int slot = 0, slot_in = NBUF - 1;
    for (int t = 0; t < nt; ++t) {
      const int younger = nt - 1 - t < NBUF - 2 ? nt - 1 - t : NBUF - 2;
      if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * PER < 64 ? (NBUF - 2) * PER : 0) : "memory");
      __builtin_amdgcn_s_barrier();
      if (t + NBUF - 1 < nt) stage(t_first + t + NBUF - 1, slot_in);
      compute(slot);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (this tile's fragment reads have returned before the next barrier lets its slot go)
      slot = slot + 1 == NBUF ? 0 : slot + 1;
      slot_in = slot_in + 1 == NBUF ? 0 : slot_in + 1;
    }
    __syncthreads();  // the staged epilogue re-uses the ring's LDS
  }
  if (p.splits > 1) {  // raw fp32 slab; lane holds C[m = .. + l15][n = .. + 4g + (0..3)]
    float* slab = p.partial + (int64_t)split * p.m * p.n;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wm * TM * 16 + i * 16 + l15;
      if (m >= p.m) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 16 + j * 16 + g * 4;
        if (n < p.n) *reinterpret_cast<f32x4*>(slab + (int64_t)m * p.n + n) = acc[i][j];
      }
    }
    return;
  }
  gemm_epilogue<DT, BM, BN, TM, TN, NT>(p, acc, smem, m0, n0, wm, wn, l15, g, tid);
}

// Adds the split-K slabs in split order and applies the same epilogue as gemm_epilogue (including its
// rounding of (acc + bias + rowbias) * alpha to the activation type before the residual add).
template <int DT>
__global__ __launch_bounds__(256) void k_splitk_reduce(GemmKParams p) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c8n = p.n >> 3;
  const int64_t m = id / c8n;
  if (m >= p.m) return;
  const int n = (int)(id - m * c8n) * 8;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = 0.f;
  for (int sp = 0; sp < p.splits; ++sp) {
    const float* src = p.partial + ((int64_t)sp * p.m + m) * p.n + n;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] += a[k];
      v[4 + k] += b[k];
    }
  }
  if (p.ln_stats) {  // folded LayerNorm: rstd * (x W'^T - mean * colsum(W'))
    const float2 st = *reinterpret_cast<const float2*>(p.ln_stats + m * 2);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = st.y * (v[k] - st.x * p.ln_colsum[n + k]);
  }
  if (p.bias) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += p.bias[n + k];
  }
  if (p.rowbias) {
    const float* rbp = p.rowbias + (m / p.rows_per_group) * p.ld_rowbias + n;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += rbp[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] *= p.alpha;
  unpack8<DT>(pack8<DT>(v), v);
  if (p.res) {
    float r[8];
    unpack8<DT>(ld16(p.res + m * p.ld_res + n), r);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] += r[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] *= p.post;
  if (p.act != CA_ACT_NONE) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = act_f(v[k], p.act);
  }
  const int64_t off = m * p.ldc + n;
  if (p.out_f32) {
    float* cp = reinterpret_cast<float*>(p.c) + off;
    *reinterpret_cast<f32x4*>(cp) = (f32x4){v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(cp + 4) = (f32x4){v[4], v[5], v[6], v[7]};
  } else {
    st16(reinterpret_cast<u16*>(p.c) + off, pack8<DT>(v));
  }
}

// Split-K plan for a launch: 1 = none.  Only grids that leave the chip under-filled (8x8 / 16x16 latent
// levels) and have a long K loop are split; the tile shape used with a split is 128x128 (N % 128 == 0).
inline int splitk_plan(int m, int n, int nt, int geglu) {
  static const int env = CA_KNOB("CA_SPLITK", -1);
  if (env == 0 || geglu || n % 128 != 0) return 1;
  const int64_t blocks = (int64_t)ceil_div_i(m, 128) * (n / 128);
  if (blocks >= 384 || nt < 48) return 1;
  int s = env > 0 ? env : (int)((960 + blocks - 1) / blocks);  // measured best on 160 tiles: 6 (89 vs 181 us unsplit)
  if (s > 8) s = 8;
  while (s > 1 && nt / s < 12) --s;
  return s;
}

// Weight-resident streaming kernel (ca_gemm_wres.h) for the K = 320 GEMMs of the 64x64-latent level: does this dense
// launch take it?  M >= 16384 (measured: a tie at 32768 rows, ahead above).  (Also the condition under which ca_gemm can
// compute folded-LayerNorm statistics itself: ca_gemm_ln_inline_supported.)  The kernel reads ln_stats as (mean, rstd)
// per row: a launch that hands over partial sums (ln_parts) is not eligible.
inline bool wres_eligible(const GemmKParams& p) {
  static const int wres_env = CA_KNOB("CA_GEMM_WRES", -1);  // (experiment builds: 0 = never, 1 = whenever the shape qualifies)
  const int kc = p.c1 + p.c2;
  return wres_env != 0 && kc == 320 && p.taps == 1 && (p.c2 == 0 || p.c1 % 32 == 0) && p.n % 160 == 0 && p.n / 160 <= 32 && !p.out_f32 &&
         p.splits <= 1 && !p.ln_parts && p.a_bytes != 0 && p.w_bytes != 0 && (p.c2 == 0 || p.a2_bytes != 0) && p.a_bytes < 0x7FFFFF00u &&
         (!p.c2 || p.a2_bytes < 0x7FFFFF00u) && (!p.rowbias || p.rows_per_group % 32 == 0) &&
         (((int64_t)p.m - 1) * p.ldc + (p.geglu ? p.n / 2 : p.n)) * 2 < 0x7FFFFF00ll && (!p.res || (((int64_t)p.m - 1) * p.ld_res + p.n) * 2 < 0x7FFFFF00ll) &&
         (wres_env == 1 || p.m >= 16384);
}

// Activation-resident kernel (ca_gemm_ar.h, round 4): the same K = 320 launches when the caller also hands over W in fragment
// order (ca_gemm_args.w_frag).  A subset of what the weight-resident kernel takes: one A source, N a multiple of 64 (64-column
// panels dealt to four waves), alpha = post = 1, no activation, residual only without LayerNorm / GEGLU, row-bias groups of whole
// 128-row tiles; in-kernel LayerNorm statistics need the caller's scratch (p.partial: 8 M bytes).
// CA_GEMM_AR (experiment builds): 0 = never, 1 = every launch it can take (also N = 320), default: N >= 960.
inline bool ar_eligible(const GemmKParams& p) {
  static const int ar_env = CA_KNOB("CA_GEMM_AR", -1);
  if (ar_env == 0 || !p.wf || !wres_eligible(p)) return false;
  return p.c2 == 0 && p.n % 64 == 0 && p.alpha == 1.f && p.post == 1.f && p.act == CA_ACT_NONE && !p.row_sums && ((uintptr_t)p.wf & 15) == 0 &&
         ((uintptr_t)p.c & 15) == 0 && p.ldc % 8 == 0 && (!p.res || (((uintptr_t)p.res & 15) == 0 && p.ld_res % 8 == 0 && !p.geglu && !p.ln_colsum)) &&
         (!p.rowbias || (p.rows_per_group % 128 == 0 && !p.geglu)) && (!p.ln_inline || p.partial) &&
         (ar_env == 1 || p.n >= 960);
}

// Dense GEMMs of the 8x8-latent level (M = 2048: 160 tiles of 128x128 for 256 CUs, each walking its 20..80 K tiles
// alone at one DMA round trip per tile): the same slab schedule as the small convolutions.
inline int splitk_plan_dense(int m, int n, int nt, int geglu, int out_f32) {
  static const int env = CA_KNOB("CA_SPLITK_DENSE", -1);
  if (env == 0 || geglu || out_f32 || n % 128 != 0) return 1;
  const int64_t blocks = (int64_t)ceil_div_i(m, 128) * (n / 128);
  // (measured: 2048x1280x5120, 80 K tiles: 55 vs 61 us; 2048x1280x1280, 20 K tiles: 34 vs 19 us -- the fp32 slabs and the
  //  second launch cost more than a short K loop saves, hence the same threshold as the convolutions)
  if (blocks > 192 || nt < 48) return 1;
  int s = env > 0 ? env : (int)((960 + blocks - 1) / blocks);
  if (s > 8) s = 8;
  while (s > 1 && nt / s < 12) --s;
  return s;
}

// ---- the launch plan: WHICH kernel instantiation a set of arguments runs, as a pure function of the arguments (the
// product build has no environment knobs: CA_KNOB compiles to its default; experiment builds, -DCA_EXPERIMENTS, read
// them for same-box A/B runs).  ca_gemm_plan_name / ca_conv3x3_plan_name report it without a launch;
// tests/test_dispatch_plan.py pins every shape of the benchmark workload to its label.
enum PlanKind {
  PK_WRES = 0,    // weight-resident streaming kernel, 160-column panels (ca_gemm_wres.h)
  PK_PP2,         // 128 x 320 ping-pong tiles (ca_gemm_pp2.h)
  PK_PP2_SPLITK,  // the same with K ranges writing fp32 slabs + k_splitk_reduce
  PK_DMA,         // k_gemm_dma<bm, bn>: LDS-DMA staging, nbuf LDS stages
  PK_DMA_SPLITK,  // k_gemm_dma<128,128> K ranges + k_splitk_reduce
  PK_REG,         // k_gemm<bm, bn>: register-staged (channel counts the DMA path cannot take)
  PK_PS,          // persistent streaming kernel, 128 x 320 tiles (ca_gemm_ps.h)
  PK_PQ,          // persistent streaming kernel, 256 x 320 tiles / 128 x 80 wave tiles (ca_gemm_pq.h)
  PK_AR,          // activation-resident kernel, 128-row tiles / 128 x 80 wave tiles, W fragments from L2 (ca_gemm_ar.h)
  PK_EXP,         // experiment builds only: `exp` selects (see launch_gemm)
};
struct GemmPlan {
  int kind;
  int bm, bn, waves_m, waves_n, nbuf;
  int splits;       // K ranges (PK_*_SPLITK)
  unsigned tiles;   // output tiles (x splits = blocks) of the ping-pong kernels
  int exp;
};

inline bool dma_capable(const GemmKParams& p) {
  const int kc = p.c1 + p.c2;
  return kc % BK == 0 && (p.c2 == 0 || p.c1 % BK == 0) && p.a_bytes != 0 && p.w_bytes != 0 && (p.c2 == 0 || p.a2_bytes != 0);
}

// Persistent streaming kernel (ca_gemm_ps.h): can this launch run on it?
inline bool ps_capable(const GemmKParams& p) {
  const int nt = p.taps * p.kc_tiles;
  const int64_t ncols = p.geglu ? p.n / 2 : p.n;
  const bool fits32 = (((int64_t)p.m - 1) * p.ldc + ncols) * 2 < 0x7FFFFF00ll && (!p.res || (((int64_t)p.m - 1) * p.ld_res + p.n) * 2 < 0x7FFFFF00ll) &&
                      p.a_bytes < 0x7FFFFF00u && p.w_bytes < 0x7FFFFF00u && (!p.c2 || p.a2_bytes < 0x7FFFFF00u);
  const bool aligned = ((uintptr_t)p.c & 15) == 0 && (!p.res || ((uintptr_t)p.res & 15) == 0) && p.ldc % 8 == 0 && (!p.res || p.ld_res % 8 == 0);
  return dma_capable(p) && p.n % 320 == 0 && nt >= 2 && p.splits <= 1 && !p.out_f32 && !p.ln_inline && (p.ln_parts <= 2 || p.ln_parts == 4) && fits32 && aligned &&
         (!p.rowbias || p.rows_per_group % 64 == 0) && !(p.geglu && (p.res || p.row_sums)) && p.post == 1.f && p.act == CA_ACT_NONE;
}

// 256 x 320 streaming kernel (ca_gemm_pq.h): bias, row bias, alpha and residual only
inline bool pq_capable(const GemmKParams& p, int mode) {
  // (packed row state of the convolution gather: tap-0 pixel index in 24 signed bits, middle tap always inside the image)
  const bool conv_ok = mode != 1 || (p.pad_lo == 1 && p.ups == 0 && p.hin >= 2 && p.win >= 2 && (int64_t)(p.m / (p.hout * p.wout) + 1) * p.hin * p.win < (1 << 23) &&
                                     (p.hout - 1) * p.stride < p.hin && (p.wout - 1) * p.stride < p.win);
  const bool epi1 = p.geglu || p.ln_colsum || p.ln_stats;  // the LayerNorm / GEGLU epilogue variant: dense, no residual
  return ps_capable(p) && (!p.row_sums || (mode == 0 && !epi1)) && (mode == 1 || p.c2 == 0) && (!p.rowbias || p.rows_per_group % 128 == 0) && conv_ok &&
         (!epi1 || (mode == 0 && !p.res && !p.rowbias && p.ln_parts == 0 && (!p.ln_colsum || p.ln_stats)));
}

inline GemmPlan plan_gemm(const GemmKParams& p, int mode, bool allow_pq = true) {
  GemmPlan g{};
  g.splits = 1;
  const int kc = p.c1 + p.c2;
  const bool dma = dma_capable(p);
  const int nt = p.taps * p.kc_tiles;
  if (dma && p.splits > 1) {
    // dense only: 128x320 ping-pong tiles -- the pipelined K loop needs fewer blocks to cover the DMA latency, so fewer
    // (larger) K ranges and slabs: 2048x1280x5120 in 4 ranges x 64 tiles 45 vs 52 us.  (The 8x8-latent convolution
    // 2048x1280x11520 measured 88 vs 82 us this way and stays on the 128x128 schedule.)
    static const int pp_split_env = CA_KNOB("CA_SPLITK_PP", 1);
    if (pp_split_env && mode == 0 && p.n % 320 == 0) {
      const int tiles320 = ceil_div_i(p.m, 128) * (p.n / 320);
      int s_eff = 256 / tiles320;
      if (s_eff > p.splits) s_eff = p.splits;
      if (s_eff >= 2 && tiles320 * s_eff >= 128 && nt / s_eff >= 12) {
        g.kind = PK_PP2_SPLITK;
        g.bm = 128, g.bn = 320, g.splits = s_eff, g.tiles = (unsigned)tiles320;
        return g;
      }
    }
    g.kind = PK_DMA_SPLITK;
    g.bm = 128, g.bn = 128, g.waves_m = 2, g.waves_n = 2, g.nbuf = 1, g.splits = p.splits;
    g.tiles = (unsigned)(ceil_div_i(p.m, 128) * (p.n / 128));
    return g;
  }
  // Ping-pong kernels (8 waves, one block per CU, two wave groups alternating between an MFMA segment and a
  // fragment-read / DMA-issue segment, counted vmcnt).  Measured: the 128x320 tile divides every
  // channel count of the SD1.5 UNet exactly and wins where the 128x128 grid under-fills the chip (<= 2 rounds of tiles:
  // the 16x16- and 32x32-latent levels, +10..19%); with many rounds the exposed epilogue of a one-block-per-CU kernel
  // (35..45% of a K = 1280 GEMM) loses against 4 co-resident blocks of k_gemm_dma.
  static const int pp_env = CA_KNOB("CA_GEMM_PP", -1);  // (experiment builds: 0 = never, 2 = whenever N % 320 == 0, 1 / 3 / 4 = ca_gemm_pp.h / pp3.h)
  // Persistent streaming kernel (ca_gemm_ps.h): same main loop as the 128 x 320 ping-pong kernel, but no launch / prologue
  // bubble per tile and an epilogue whose stores nothing waits for.  Measured against the kernel each shape had before
  // (tools/ps_check.py --time, same box): 131072x320x1280 149 vs 180 us, 32768x640x640 50 vs 58, 8192x1280x1280 41.6 vs 43.3,
  // 8192x10240x1280 GEGLU 261 vs 270, 2048x10240x1280 GEGLU 67.6 vs 71.0, 32768x5120x640 GEGLU 321 vs 327; behind on long K
  // loops (its flag pieces cost ~5% of the main loop: 32768x640x2560 134 vs 122, 8192x1280x5120 114 vs 109), on wide plain
  // outputs where four co-resident 128x128 blocks already hide their epilogues (32768x1920x640 116 vs 107) and on every
  // convolution (-10..-25%).  Hence: dense, 2..20 K tiles, at least one tile per CU, GEGLU or at most four column tiles.
  // CA_GEMM_PS (experiment builds): 0 = never, 1 = every launch it can take, 2 = the same except the weight-resident kernel's.
  // 256 x 320 streaming kernel (ca_gemm_pq.h): 128 x 80 wave tiles take a quarter of the 128 x 320 kernels' LDS-port and
  // global -> LDS traffic per FLOP; it needs one tile per CU and a long K loop, and has no LayerNorm / GEGLU / row-sum epilogue.
  // Measured against the kernel each shape had before (tools/ps_check.py --time, one process, us): dense 32768x640x2560 106 vs
  // 124, 131072x320x1280 136 vs 157, 32768x640x640 42 vs 52, 32768x1920x640 (folded LayerNorm) 90 vs 102, GEGLU projections
  // 32768x5120x640 260 vs 331, 8192x10240x1280 210 vs 291 (1.02 PFLOP/s), 2048x10240x1280 58 vs 77; convolutions at 32x32
  // latents 640->640 228 vs 290, 1280->640 433 vs 556; behind where the 256-row tiles leave CUs idle (M = 8192 x N = 1280: 128
  // tiles, 154 vs 109; 8192x3840x1280: 384 tiles = 1.5 rounds, 107 vs 95) and on the 64x64-latent convolutions (320->320 296 vs
  // 265).  Step, one box, knobs build: off 66.04 / 65.82, convolutions only 65.84 / 65.62, + GEGLU 64.95 / 64.79, + plain dense
  // 64.37 / 64.34, + folded-LayerNorm projections 63.75 / 63.94.
  // CA_GEMM_PQ (experiment builds): 0 = never, 1 = every launch it can take
  static const int pq_env = CA_KNOB("CA_GEMM_PQ", -1);
  if (allow_pq && pq_env != 0 && pq_capable(p, mode)) {
    const int64_t tiles = (int64_t)ceil_div_i(p.m, 256) * (p.n / 320);
    // (whole rounds of 256 tiles, or many: 8192x3840x1280 = 384 tiles measured 107 vs 95 us on the 128x128 kernel)
    // (convolutions, clean build: 64x64 latents 640->320 468 vs 508, 640->640 908 vs 986, 32x32 1280->1280 820 vs 929; 320->320 at 64x64 -- N = 320, 45 K tiles -- 256 vs 251: not)
    const bool dflt = mode == 1 ? (tiles >= 256 && (p.n >= 640 || nt >= 64)) : (tiles >= 256 && (tiles % 256 == 0 || tiles >= 1024) && nt >= 8 && !wres_eligible(p));
    // (experiment builds, CA_GEMM_PQ: 2 = convolutions + GEGLU projections, 3 = convolutions only, 4 = 2 + plain dense, 5 = everything the rule allows)
    const bool epi1 = p.geglu || p.ln_colsum || p.ln_stats;
    const bool dflt2 = dflt && (mode == 1 || p.geglu), dflt3 = dflt && mode == 1, dflt4 = dflt && (mode == 1 || p.geglu || !epi1);
    if ((pq_env < 0 && dflt) || pq_env == 1 || (pq_env == 2 && dflt2) || (pq_env == 3 && dflt3) || (pq_env == 4 && dflt4) || (pq_env == 5 && dflt)) {
      g.kind = PK_PQ;
      g.bm = 256, g.bn = 320, g.tiles = (unsigned)tiles;
      return g;
    }
  }
  static const int ps_env = CA_KNOB("CA_GEMM_PS", -1);
  if (ps_env != 0 && ps_capable(p)) {
    const int64_t tiles = (int64_t)ceil_div_i(p.m, 128) * (p.n / 320);
    const bool dflt = mode == 0 && !wres_eligible(p) && nt <= 20 && tiles >= 256 && (p.geglu || p.n <= 1280);
    if ((ps_env < 0 && dflt) || ps_env == 1 || (ps_env == 2 && !(mode == 0 && wres_eligible(p)))) {
      g.kind = PK_PS;
      g.bm = 128, g.bn = 320, g.tiles = (unsigned)tiles;
      return g;
    }
  }
  if (mode == 0 && dma && ar_eligible(p)) {
    g.kind = PK_AR;
    g.bm = 128, g.bn = 64;
    return g;
  }
  if (mode == 0 && dma && wres_eligible(p)) {
    g.kind = PK_WRES;
    g.bm = 256, g.bn = 160;
    return g;
  }
  if (dma && pp_env != 0 && pp_env != 3 && nt >= 2 && p.n % 320 == 0 && p.splits <= 1) {  // 128 x 320 tiles
    const int64_t tiles = (int64_t)ceil_div_i(p.m, 128) * (p.n / 320);
#ifdef CA_EXPERIMENTS
    const int64_t ncols = p.geglu ? p.n / 2 : p.n;
    const bool fits32 = (((int64_t)p.m - 1) * p.ldc + ncols) * 2 < 0x7FFFFF00ll && (!p.res || (((int64_t)p.m - 1) * p.ld_res + p.n) * 2 < 0x7FFFFF00ll) &&
                        p.a_bytes < 0x7FFFFF00u && p.w_bytes < 0x7FFFFF00u && (!p.c2 || p.a2_bytes < 0x7FFFFF00u);
    const bool pp3_ok = nt >= 5 && !p.out_f32 && !p.ln_parts && fits32 && (!p.rowbias || p.rows_per_group % 64 == 0) && p.ldc % 8 == 0 && (!p.res || p.ld_res % 8 == 0);
    if (pp3_ok && pp_env == 4) {
      g.kind = PK_EXP, g.exp = 321, g.tiles = (unsigned)tiles;
      return g;
    }
#endif
    // (thresholds re-checked inside the step, same box, interleaved: dense 768 / 1024 tiles +0.25 ms, conv 512 +0.7, conv 128 +0.2)
    if (p.row_sums || pp_env == 1 || pp_env == 2 || (pp_env < 0 && tiles >= 128 && nt >= 10 && (tiles <= 256 || (tiles <= 512 && mode == 0)))) {
      g.kind = PK_PP2;
      g.bm = 128, g.bn = 320, g.tiles = (unsigned)tiles;
      return g;
    }
  }
#ifdef CA_EXPERIMENTS
  if (dma && (pp_env == 1 || pp_env == 3) && nt >= 2 && p.n % 128 == 0 && p.splits <= 1) {
    const bool bn256 = p.n % 256 == 0;
    g.kind = PK_EXP, g.exp = bn256 ? 256 : 128;
    g.tiles = (unsigned)((int64_t)ceil_div_i(p.m, 256) * (p.n / (bn256 ? 256 : 128)));
    return g;
  }
#endif
  // 128x128 tiles unless N is not a multiple of 128 or the grid would leave CUs idle
  // (8x8 / 16x16 latent levels: M = 2048 / 8192 rows -> < 2 blocks per CU with the big tile).
  static const int bn_env = CA_KNOB("CA_GEMM_BN", 0);
  bool wide = p.n % 128 == 0 && (int64_t)ceil_div_i(p.m, 128) * ceil_div_i(p.n, 128) >= 512;
  if (bn_env == 64) wide = false;
  if (bn_env == 128 && p.n % 128 == 0) wide = true;
  const int64_t blocks = (int64_t)ceil_div_i(p.m, 128) * ceil_div_i(p.n, wide ? 128 : 64);
  // LDS stages: ONE buffer (32 KB, two barriers per tile) lets 4 blocks share a CU, whose MFMA phases
  // cover each other's transfer latency: measured +10..25% over double buffering (2 blocks per CU)
  // and far better than 3-4 stage rings (1 block per CU).  Small grids (< 2 blocks per CU) have no
  // co-resident blocks to overlap with and keep the double buffer.
  static const int nbuf_env = CA_KNOB("CA_GEMM_NBUF", 0);
  // (round 4: three stages instead of two for the small grids -- their K loops are chains of DMA round trips with 8 MFMAs per
  //  wave and tile in between; two tiles in flight instead of one: -0.3 ms per step, 62.0 vs 62.3 interleaved three times.  The
  //  128 x 64 tile's ring is 74 KB: two blocks still share a CU.  Four stages (98 KB, one block per CU) lose.)
  int nbuf = nbuf_env ? nbuf_env : (blocks >= 512 ? 1 : 3);
  if (nbuf < 1 || nbuf > 4) nbuf = 3;
  static const int ring_env = CA_KNOB("CA_GEMM_RING", 0);  // (experiment builds: 2 = the round-3 double buffer, 4 = four stages, for the launches that take the ring)
  if (nbuf == 3 && (ring_env == 2 || ring_env == 4)) nbuf = ring_env;
  // N = 320 / 960 (every projection and conv of the 64x64-latent level): 128x160 tiles divide N
  // exactly and read the A panel 2 / 6 times instead of 5 / 15 times
  static const int t160_env = CA_KNOB("CA_GEMM_T160", 1);
  if (dma && !wide && t160_env && p.n % 160 == 0 && (int64_t)ceil_div_i(p.m, 128) * (p.n / 160) >= 512) {
    g.kind = PK_DMA;
    g.bm = 128, g.bn = 160, g.waves_m = 2, g.waves_n = 2, g.nbuf = 1;
    return g;
  }
#ifdef CA_EXPERIMENTS
  // CA_GEMM_BIG: 1 = 256x128 tiles whenever the grid allows, 3 = the wide feed-forward GEMMs only (the round-1 default:
  // +4..8% on 8192x10240x1280 and 32768x5120x640 measured in isolation; inside the step, with the ControlNet stream
  // beside it, the 128x128 tiles are 0.3 ms faster), 2 = 4 waves x (128 x 64) per wave, 2 blocks per CU
  static const int big_env = CA_KNOB("CA_GEMM_BIG", 0);
  const bool big = big_env == 1 || (big_env == 3 && mode == 0 && p.n >= 5120 && kc >= 640);
  if (dma && big_env == 2 && p.n % 128 == 0 && (int64_t)ceil_div_i(p.m, 256) * (p.n / 128) >= 256) {
    g.kind = PK_EXP, g.exp = 2562;
    return g;
  }
  if (dma && big && p.n % 128 == 0 && (int64_t)ceil_div_i(p.m, 256) * (p.n / 128) >= 512) {
    g.kind = PK_EXP, g.exp = 2561;
    return g;
  }
  static const int kt_env = CA_KNOB("CA_GEMM_KT", 64);
  if (wide && dma && kt_env == 32) {
    g.kind = PK_EXP, g.exp = 32;
    return g;
  }
#else
  (void)kc;
#endif
  g.kind = dma ? PK_DMA : PK_REG;
  g.bm = 128, g.bn = wide ? 128 : 64, g.waves_m = wide ? 2 : 4, g.waves_n = wide ? 2 : 1, g.nbuf = dma ? nbuf : 2;
  return g;
}

// can the epilogue of this (dense) launch leave per-row sums of its output (ca_gemm_args.row_sums_out)?  Only the 128 x 320
// tile kernels do; the answer is about the launch the arguments get WITHOUT the pointer.
inline int row_sums_parts_of(GemmKParams p) {  // partial sums per row the launch can leave (0: none)
  p.row_sums = nullptr;
  if (p.geglu || p.out_f32) return 0;
  const int k = plan_gemm(p, 0).kind;
  if (k == PK_PP2 || k == PK_PS) return p.n / 320;          // one (sum, sum of squares) per 320-column tile
  if (k == PK_PQ && !p.ln_colsum && !p.ln_stats) return 4 * (p.n / 320);  // the 256 x 320 kernel: one per 80-column wave quarter
  return 0;
}
inline bool row_sums_capable(const GemmKParams& p) { return row_sums_parts_of(p) > 0; }

inline void plan_label(const GemmPlan& g, char* buf, int len) {
  switch (g.kind) {
    case PK_WRES: snprintf(buf, len, "wres160"); break;
    case PK_AR: snprintf(buf, len, "ar128x64"); break;
    case PK_PP2: snprintf(buf, len, "pp128x320"); break;
    case PK_PS: snprintf(buf, len, "ps128x320"); break;
    case PK_PQ: snprintf(buf, len, "pq256x320"); break;
    case PK_PP2_SPLITK: snprintf(buf, len, "pp128x320_splitk%d", g.splits); break;
    case PK_DMA: snprintf(buf, len, "%dx%d%s", g.bm, g.bn, g.nbuf == 2 ? "_db" : g.nbuf == 3 ? "_r3" : g.nbuf == 4 ? "_r4" : ""); break;
    case PK_DMA_SPLITK: snprintf(buf, len, "128x128_splitk%d", g.splits); break;
    case PK_REG: snprintf(buf, len, "reg_%dx%d", g.bm, g.bn); break;
    default: snprintf(buf, len, "exp%d", g.exp); break;
  }
}

template <int DT, int MODE>
int launch_gemm(const GemmKParams& p, hipStream_t st) {
  const GemmPlan g = plan_gemm(p, MODE);
  const dim3 grid(ceil_div_i(p.m, g.bm ? g.bm : 128) * ceil_div_i(p.n, g.bn ? g.bn : 128));
  switch (g.kind) {
    case PK_WRES: return ca_launch_gemm_pp(p, DT, MODE, 160, 0u, st);
    case PK_AR: return ca_launch_gemm_ar(p, DT, st);
    case PK_PP2: return ca_launch_gemm_pp(p, DT, MODE, 320, g.tiles, st);
    case PK_PS: return ca_launch_gemm_pp(p, DT, MODE, 322, g.tiles, st);
    case PK_PQ: return ca_launch_gemm_pp(p, DT, MODE, 323, g.tiles, st);
    case PK_PP2_SPLITK: {
      GemmKParams q = p;
      q.splits = g.splits;
      const int rc = ca_launch_gemm_pp(q, DT, MODE, 320, g.tiles * (unsigned)g.splits, st);
      hipLaunchKernelGGL((k_splitk_reduce<DT>), dim3(ceil_div_i((int64_t)q.m * (q.n / 8), 256)), dim3(256), 0, st, q);
      return rc;
    }
    case PK_DMA_SPLITK:
      hipLaunchKernelGGL((k_gemm_dma<DT, 128, 128, 2, 2, MODE, 1>), dim3(g.tiles * (unsigned)p.splits), dim3(256), 0, st, p);
      hipLaunchKernelGGL((k_splitk_reduce<DT>), dim3(ceil_div_i((int64_t)p.m * (p.n / 8), 256)), dim3(256), 0, st, p);
      return CA_OK;
    case PK_DMA:
      if (g.bn == 160) hipLaunchKernelGGL((k_gemm_dma<DT, 128, 160, 2, 2, MODE, 1>), grid, dim3(256), 0, st, p);
      else if (g.bn == 128 && g.nbuf == 1) hipLaunchKernelGGL((k_gemm_dma<DT, 128, 128, 2, 2, MODE, 1>), grid, dim3(256), 0, st, p);
      else if (g.bn == 128 && g.nbuf == 4) hipLaunchKernelGGL((k_gemm_dma<DT, 128, 128, 2, 2, MODE, 4>), grid, dim3(256), 0, st, p);
      else if (g.bn == 128 && g.nbuf == 3) hipLaunchKernelGGL((k_gemm_dma<DT, 128, 128, 2, 2, MODE, 3>), grid, dim3(256), 0, st, p);
      else if (g.bn == 128) hipLaunchKernelGGL((k_gemm_dma<DT, 128, 128, 2, 2, MODE, 2>), grid, dim3(256), 0, st, p);
      else if (g.bn == 64 && g.nbuf == 4) hipLaunchKernelGGL((k_gemm_dma<DT, 128, 64, 4, 1, MODE, 4>), grid, dim3(256), 0, st, p);
      else if (g.bn == 64 && g.nbuf == 3) hipLaunchKernelGGL((k_gemm_dma<DT, 128, 64, 4, 1, MODE, 3>), grid, dim3(256), 0, st, p);
      else if (g.nbuf == 1) hipLaunchKernelGGL((k_gemm_dma<DT, 128, 64, 4, 1, MODE, 1>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((k_gemm_dma<DT, 128, 64, 4, 1, MODE, 2>), grid, dim3(256), 0, st, p);
      return CA_OK;
    case PK_REG:
      if (g.bn == 128) hipLaunchKernelGGL((k_gemm<DT, 128, 128, 2, 2, MODE>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((k_gemm<DT, 128, 64, 4, 1, MODE>), grid, dim3(256), 0, st, p);
      return CA_OK;
    default: break;
  }
#ifdef CA_EXPERIMENTS
  if (g.exp == 321 || g.exp == 256 || g.exp == 128) return ca_launch_gemm_pp(p, DT, MODE, g.exp, g.tiles, st);
  if (g.exp == 2562) hipLaunchKernelGGL((k_gemm_dma<DT, 256, 128, 2, 2, MODE, 1>), dim3(ceil_div_i(p.m, 256) * (p.n / 128)), dim3(256), 0, st, p);
  if (g.exp == 2561) hipLaunchKernelGGL((k_gemm_dma<DT, 256, 128, 4, 2, MODE, 1>), dim3(ceil_div_i(p.m, 256) * (p.n / 128)), dim3(512), 0, st, p);
  if (g.exp == 32) hipLaunchKernelGGL((k_gemm_dma<DT, 128, 128, 2, 2, MODE, 2, 32>), dim3(ceil_div_i(p.m, 128) * ceil_div_i(p.n, 128)), dim3(256), 0, st, p);
  return CA_OK;
#else
  return CA_ERR_LAUNCH;
#endif
}

// descriptor size in bytes, or 0 when the buffer is too large for 32-bit offsets (-> register variant)
inline unsigned desc_bytes(int64_t elems) {
  const int64_t b = elems * 2;
  return (b > 0 && b < (int64_t)0xFFFFFF00ll) ? (unsigned)b : 0u;
}

int check_epilogue(const char* who, int n, int geglu, int out_f32, int64_t ldc, int64_t ld_res, const void* res) {
  // N = 4 (conv_out) takes the direct 4-column epilogue; everything wider goes through the LDS-staged
  // epilogue, which moves 8-column (16-byte) chunks: N, ldc and ld_res must then be multiples of 8
  // (N = 12, 20, ... would store 8 values at column N-4: past the row end)
  CA_REQUIRE(n == 4 || (n >= 8 && n % 8 == 0), "%s: N=%d must be 4 or a multiple of 8", who, n);
  CA_REQUIRE(!geglu || n % 8 == 0, "%s: geglu needs N %% 8 == 0", who);
  if (n >= 8) {
    CA_REQUIRE(ldc % (geglu ? 4 : 8) == 0, "%s: ldc=%lld must be a multiple of %d", who, (long long)ldc, geglu ? 4 : 8);
    CA_REQUIRE(!res || ld_res % 8 == 0, "%s: ld_res=%lld must be a multiple of 8", who, (long long)ld_res);
  } else {
    CA_REQUIRE(ldc % 4 == 0, "%s: ldc=%lld misaligned", who, (long long)ldc);
    CA_REQUIRE(!res || ld_res % 4 == 0, "%s: ld_res=%lld misaligned", who, (long long)ld_res);
  }
  (void)out_f32;
  return CA_OK;
}

}  // namespace

static int gemm_fill(const ca_gemm_args* a, GemmKParams& p) {
  CA_REQUIRE(a != nullptr, "ca_gemm: null args");
  CA_REQUIRE(a->a && a->w && a->c, "ca_gemm: null operand");
  CA_REQUIRE(a->m > 0 && a->k1 > 0 && a->k2 >= 0, "ca_gemm: bad sizes m=%d k1=%d k2=%d", a->m, a->k1, a->k2);
  CA_REQUIRE(a->k1 % 8 == 0 && a->k2 % 8 == 0, "ca_gemm: k1=%d k2=%d must be multiples of 8", a->k1, a->k2);
  CA_REQUIRE(a->lda % 8 == 0 && (a->k2 == 0 || (a->a2 && a->lda2 % 8 == 0)), "ca_gemm: lda/lda2 misaligned or a2 missing");
  CA_REQUIRE(a->dtype == CA_BF16 || a->dtype == CA_F16, "ca_gemm: dtype %d", a->dtype);
  CA_REQUIRE(!a->rowbias || a->rows_per_group > 0, "ca_gemm: rows_per_group");
  CA_REQUIRE(!a->rowbias || a->ld_rowbias % 4 == 0, "ca_gemm: ld_rowbias misaligned");
  int rc = check_epilogue("ca_gemm", a->n, a->geglu, a->out_f32, a->ldc, a->ld_res, a->residual);
  if (rc) return rc;
  p.a = (const u16*)a->a;
  p.a2 = (const u16*)a->a2;
  p.w = (const u16*)a->w;
  p.c = a->c;
  p.bias = a->bias;
  p.rowbias = a->rowbias;
  p.ln_stats = a->ln_stats;
  p.ln_colsum = a->ln_colsum;
  p.row_sums = a->row_sums_out;
  p.ln_parts = a->ln_parts;
  CA_REQUIRE(a->ln_parts >= 0 && a->ln_parts <= 16, "ca_gemm: ln_parts=%d", a->ln_parts);
  CA_REQUIRE(a->ln_parts == 0 || (a->ln_stats && a->ln_colsum && a->ln_eps > 0.f), "ca_gemm: ln_parts needs ln_stats (the partial sums), ln_colsum and ln_eps > 0");
  p.ln_inline = (a->ln_colsum && !a->ln_stats) ? 1 : 0;
  p.ln_eps = a->ln_eps;
  p.wf = (const u16*)a->w_frag;
  CA_REQUIRE(!a->ln_stats || a->ln_colsum, "ca_gemm: ln_stats without ln_colsum");
  CA_REQUIRE(!p.ln_inline || a->ln_eps > 0.f, "ca_gemm: ln_colsum without ln_stats asks for in-kernel statistics and needs ln_eps > 0");
  CA_REQUIRE(!a->ln_colsum || (a->n >= 8 && a->k2 == 0), "ca_gemm: the folded LayerNorm needs N >= 8 and a single A source");
  p.res = (const u16*)a->residual;
  p.lda = a->lda;
  p.lda2 = a->lda2;
  p.ldc = a->ldc;
  p.ld_res = a->ld_res;
  p.ld_rowbias = a->ld_rowbias;
  p.a_bytes = desc_bytes((int64_t)(a->m - 1) * a->lda + a->k1);
  p.a2_bytes = a->k2 ? desc_bytes((int64_t)(a->m - 1) * a->lda2 + a->k2) : 0u;
  p.w_bytes = desc_bytes((int64_t)a->n * (a->k1 + a->k2));
  p.m = a->m;
  p.n = a->n;
  p.c1 = a->k1;
  p.c2 = a->k2;
  p.taps = 1;
  p.kc_tiles = ceil_div_i(a->k1 + a->k2, BK);
  p.rows_per_group = a->rows_per_group > 0 ? a->rows_per_group : 1;
  p.alpha = a->alpha;
  p.post = a->post_scale;
  p.act = a->act;
  p.geglu = a->geglu;
  p.out_f32 = a->out_f32;
  p.splits = 1;
  return CA_OK;
}

// gemm_fill + the split-K decision: everything the plan depends on
static int gemm_prepare(const ca_gemm_args* a, GemmKParams& p) {
  int rc = gemm_fill(a, p);
  if (rc) return rc;
  {
    const bool dma_ok = (a->k1 + a->k2) % BK == 0 && (a->k2 == 0 || a->k1 % BK == 0);
    // (k_splitk_reduce and the weight-resident kernel read ln_stats as (mean, rstd): partial sums never take those plans)
    const int s = dma_ok && !p.ln_inline && !p.row_sums && !p.ln_parts ? splitk_plan_dense(p.m, p.n, p.kc_tiles, p.geglu, p.out_f32) : 1;
    if (s > 1 && a->workspace && a->workspace_bytes >= (int64_t)s * p.m * p.n * 4) {
      p.splits = s;
      p.partial = reinterpret_cast<float*>(a->workspace);
    }
  }
  if (p.ln_inline && p.wf && a->workspace && a->workspace_bytes >= (int64_t)p.m * 8) p.partial = reinterpret_cast<float*>(a->workspace);  // (mean, rstd) scratch of k_gemm_ar
  CA_REQUIRE(!p.row_sums || row_sums_capable(p), "ca_gemm: row_sums_out is not available for this launch: ask ca_gemm_row_sums_parts() first");
  CA_REQUIRE(!p.ln_inline || wres_eligible(p), "ca_gemm: in-kernel LayerNorm statistics (ln_stats NULL) are not available for this launch: "
             "ask ca_gemm_ln_inline_supported() first and pass ln_stats otherwise");
  return CA_OK;
}

extern "C" int ca_gemm(const ca_gemm_args* a, void* stream) {
  GemmKParams p{};
  int rc = gemm_prepare(a, p);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == CA_BF16) launch_gemm<CA_BF16, 0>(p, st);
  else launch_gemm<CA_F16, 0>(p, st);
  CA_CHECK_LAUNCH("ca_gemm");
  return CA_OK;
}

extern "C" int64_t ca_gemm_workspace_bytes(const ca_gemm_args* a) {
  if (!a || a->m <= 0 || a->n <= 0 || a->k1 <= 0 || a->k2 < 0) return 0;
  const int kc = a->k1 + a->k2;
  if (a->ln_colsum && !a->ln_stats && a->w_frag && !a->ln_parts && !a->row_sums_out) {
    // in-kernel LayerNorm statistics of the activation-resident kernel: (mean, rstd) per row
    GemmKParams p{};
    if (gemm_fill(a, p) != CA_OK) return 0;
    p.partial = reinterpret_cast<float*>(uintptr_t(16));  // "a scratch is there": would the launch take that kernel?
    return ar_eligible(p) ? (int64_t)a->m * 8 : 0;
  }
  if (kc % BK != 0 || (a->k2 != 0 && a->k1 % BK != 0) || (a->ln_colsum && !a->ln_stats) || a->ln_parts || a->row_sums_out) return 0;
  const int s = splitk_plan_dense(a->m, a->n, ceil_div_i(kc, BK), a->geglu, a->out_f32);
  return s > 1 ? (int64_t)s * a->m * a->n * 4 : 0;
}

extern "C" int ca_gemm_row_sums_parts(const ca_gemm_args* a) {
  GemmKParams p{};
  if (!a || a->geglu || a->out_f32) return 0;
  ca_gemm_args b = *a;
  b.row_sums_out = nullptr;  // (the question is about the launch, whatever the pointer)
  if (gemm_fill(&b, p) != CA_OK) return 0;
  const int kc = a->k1 + a->k2;
  const bool dma_ok = kc % BK == 0 && (a->k2 == 0 || a->k1 % BK == 0);
  if (dma_ok && !p.ln_inline && !p.ln_parts && a->workspace && splitk_plan_dense(p.m, p.n, p.kc_tiles, p.geglu, p.out_f32) > 1) return 0;
  return row_sums_parts_of(p);
}

extern "C" int ca_gemm_ln_inline_supported(const ca_gemm_args* a) {
  GemmKParams p{};
  if (!a || !a->ln_colsum || gemm_fill(a, p) != CA_OK) return 0;
  return wres_eligible(p) ? 1 : 0;
}

// 1 if this launch, which hands over partial sums (ln_parts > 0), would run on a kernel that takes finished (mean, rstd) only
// and is the faster one for the shape (the 256 x 320 streaming kernel): the caller then finishes the sums (ca_ln_finish_sums)
// and passes ln_parts = 0.
extern "C" int ca_gemm_wants_finished_stats(const ca_gemm_args* a) {
  GemmKParams p{};
  if (!a || !a->ln_colsum || !a->ln_stats || a->ln_parts <= 0 || a->row_sums_out) return 0;
  ca_gemm_args b = *a;
  b.ln_parts = 0;
  if (gemm_fill(&b, p) != CA_OK) return 0;
  return plan_gemm(p, 0).kind == PK_PQ ? 1 : 0;
}

// The Winograd route of ca_conv3x3 (ca_conv_wino.h): 0 = not taken, else the workspace it needs (V [16][T][cin] + M [16][T][cout]).
static int64_t wino_workspace_bytes(const ca_conv_args* a) {
  if (!a || !a->w_wino || (a->dtype != CA_F16 && a->dtype != CA_BF16) || a->stride != 1 || a->pad_asym || a->out_f32) return 0;
  if (a->upsample != 0 && a->upsample != 1) return 0;
  if (a->x_is_wino_v && (a->cin2 != 0 || a->upsample)) return 0;
  const int h = a->hin << a->upsample, w = a->win << a->upsample;  // logical input = output size
  if (a->images <= 0 || h < 2 || w < 2 || (h & 1) || (w & 1)) return 0;
  const int kc = a->cin1 + a->cin2;
  static const int min_cin = CA_KNOB("CA_WINO_MIN_CIN", 1280);  // (experiments: where the route stops paying)
  if (kc < 640 || kc % BK != 0 || a->cin1 % 8 != 0 || a->cin2 % 8 != 0 || a->cout % 320 != 0) return 0;
  const int64_t tiles = (int64_t)a->images * (h / 2) * (w / 2);
  // input channels: >= 1280 everywhere in the window; 640 .. 1279 only at <= 4096 tiles, where the direct form is short of tiles
  // (32 x 16x16 640->1280: 100 vs 160 us; at 8192 tiles 640->640 254 vs 233-252, 960->640 318 vs 331: no / marginal gain)
  if (kc < min_cin && !(min_cin == 1280 && tiles <= 4096)) return 0;
  // whole 256-row tiles per transformed GEMM.  Measured (tools/wino_check.py, us, Winograd vs direct): 32 x 16x16 1280->1280 170 vs 276,
  // 2560->1280 285 vs 529, 32 x 8x8 1280->1280 66 vs 87, 32 x 32x32 1920->640 510 vs 584, 1280->1280 634 vs 800; with 640 input channels
  // the sixteen K = 640 GEMMs are epilogue-bound and the 4 x larger V / M tensors cost more than the saved MFMAs (no gain): >= 1280 only
  static const int max_tiles = CA_KNOB("CA_WINO_MAX_TILES", 16384);
  if (tiles % 256 != 0 || tiles > max_tiles) return 0;
  if (16 * tiles * (int64_t)(kc > a->cout ? kc : a->cout) * 2 >= 0x7FFFFF00ll) return 0;  // 32-bit byte offsets in the GEMM
  return 16 * tiles * (int64_t)((a->x_is_wino_v ? 0 : kc) + a->cout) * 2;  // V (unless the caller hands it over as x) + M
}

extern "C" int64_t ca_conv3x3_workspace_bytes(const ca_conv_args* a) {
  if (!a || a->images <= 0 || a->hin <= 0 || a->win <= 0 || (a->stride != 1 && a->stride != 2)) return 0;
  {
    const int64_t wb = wino_workspace_bytes(a);
    if (wb > 0) return wb;
  }
  const int hl = a->hin << a->upsample, wl = a->win << a->upsample;
  const int pad = a->pad_asym ? 1 : 2;
  const int64_t m = (int64_t)a->images * ((hl + pad - 3) / a->stride + 1) * ((wl + pad - 3) / a->stride + 1);
  const int kc = a->cin1 + a->cin2;
  if (m >= (1ll << 31) || kc % BK != 0 || (a->cin2 != 0 && a->cin1 % BK != 0)) return 0;
  const int s = splitk_plan((int)m, a->cout, 9 * ceil_div_i(kc, BK), 0);
  return s > 1 ? (int64_t)s * m * a->cout * 4 : 0;
}

static int conv_prepare(const ca_conv_args* a, GemmKParams& p) {
  CA_REQUIRE(a != nullptr, "ca_conv3x3: null args");
  CA_REQUIRE(a->x && a->w && a->y, "ca_conv3x3: null operand");
  CA_REQUIRE(a->images > 0 && a->hin > 0 && a->win > 0, "ca_conv3x3: bad geometry");
  CA_REQUIRE(a->cin1 > 0 && a->cin1 % 8 == 0 && a->cin2 >= 0 && a->cin2 % 8 == 0,
             "ca_conv3x3: cin1=%d cin2=%d must be multiples of 8", a->cin1, a->cin2);
  CA_REQUIRE(a->cin2 == 0 || a->x2, "ca_conv3x3: x2 missing");
  CA_REQUIRE(a->stride == 1 || a->stride == 2, "ca_conv3x3: stride %d", a->stride);
  CA_REQUIRE(a->upsample == 0 || a->upsample == 1, "ca_conv3x3: upsample %d", a->upsample);
  CA_REQUIRE(a->dtype == CA_BF16 || a->dtype == CA_F16, "ca_conv3x3: dtype %d", a->dtype);
  CA_REQUIRE(!a->rowbias || a->rows_per_group > 0, "ca_conv3x3: rows_per_group");
  int rc = check_epilogue("ca_conv3x3", a->cout, 0, a->out_f32, a->cout, a->ld_res, a->residual);
  if (rc) return rc;
  const int hl = a->hin << a->upsample, wl = a->win << a->upsample;
  CA_REQUIRE(a->pad_asym == 0 || a->pad_asym == 1, "ca_conv3x3: pad_asym %d", a->pad_asym);
  const int pad = a->pad_asym ? 1 : 2;  // total padding per axis: 1+1, or 0 before / 1 after
  const int hout = (hl + pad - 3) / a->stride + 1;
  const int wout = (wl + pad - 3) / a->stride + 1;
  const int64_t m64 = (int64_t)a->images * hout * wout;
  CA_REQUIRE(m64 < (1ll << 31), "ca_conv3x3: too many output pixels");
  p.a = (const u16*)a->x;
  p.a2 = (const u16*)a->x2;
  p.w = (const u16*)a->w;
  p.c = a->y;
  p.bias = a->bias;
  p.rowbias = a->rowbias;
  p.res = (const u16*)a->residual;
  p.ldc = a->cout;
  p.ld_res = a->ld_res;
  p.ld_rowbias = a->ld_rowbias;
  p.a_bytes = desc_bytes((int64_t)a->images * a->hin * a->win * a->cin1);
  p.a2_bytes = a->cin2 ? desc_bytes((int64_t)a->images * a->hin * a->win * a->cin2) : 0u;
  p.w_bytes = desc_bytes((int64_t)a->cout * 9 * (a->cin1 + a->cin2));
  p.m = (int)m64;
  p.n = a->cout;
  p.c1 = a->cin1;
  p.c2 = a->cin2;
  p.taps = 9;
  static const int tap_inner_env = CA_KNOB("CA_CONV_TAP_INNER", 1);
  p.tap_inner = tap_inner_env;  // (0: taps outermost, the round-1 order -- A/B experiments)
  p.kc_tiles = ceil_div_i(a->cin1 + a->cin2, BK);
  p.hin = a->hin;
  p.win = a->win;
  p.hout = hout;
  p.wout = wout;
  p.stride = a->stride;
  p.ups = a->upsample;
  p.pad_lo = a->pad_asym ? 0 : 1;
  p.rows_per_group = a->rows_per_group > 0 ? a->rows_per_group : 1;
  p.alpha = a->alpha;
  p.post = a->post_scale;
  p.act = a->act;
  p.geglu = 0;
  p.out_f32 = a->out_f32;
  p.splits = 1;
  {
    const int kc = a->cin1 + a->cin2;
    const bool dma_ok = kc % BK == 0 && (a->cin2 == 0 || a->cin1 % BK == 0) && p.a_bytes && p.w_bytes && (a->cin2 == 0 || p.a2_bytes);
    const int s = dma_ok ? splitk_plan(p.m, p.n, 9 * p.kc_tiles, 0) : 1;
    if (s > 1 && a->workspace && a->workspace_bytes >= (int64_t)s * p.m * p.n * 4) {
      p.splits = s;
      p.partial = (float*)a->workspace;
    }
  }
  return CA_OK;
}

static bool wino_taken(const ca_conv_args* a) {
  const int64_t wb = wino_workspace_bytes(a);
  return wb > 0 && a->workspace && a->workspace_bytes >= wb && (((uintptr_t)a->workspace | (uintptr_t)a->w_wino) & 15) == 0;
}

static int launch_conv_wino(const ca_conv_args* a, const GemmKParams& cp, hipStream_t st) {
  const int kc = a->cin1 + a->cin2;
  const int hl = a->hin << a->upsample, wl = a->win << a->upsample;
  const int64_t tiles = (int64_t)a->images * (hl / 2) * (wl / 2);
  WinoParams w{};
  w.x = (const u16*)a->x;
  w.x2 = (const u16*)a->x2;
  w.v = (u16*)a->workspace;
  u16* mm = (u16*)a->workspace + (a->x_is_wino_v ? 0 : 16 * tiles * kc);
  w.mm = mm;
  w.y = (u16*)a->y;
  w.bias = a->bias;
  w.rowbias = a->rowbias;
  w.res = (const u16*)a->residual;
  w.ld_res = a->ld_res;
  w.ld_rowbias = a->ld_rowbias;
  w.images = a->images, w.h = hl, w.w = wl, w.c1 = a->cin1, w.c2 = a->cin2, w.cout = a->cout;
  w.ups = a->upsample;
  w.rows_per_group = cp.rows_per_group;
  w.alpha = a->alpha, w.post = a->post_scale, w.act = a->act;
  const int64_t in_threads = tiles * (kc / 8), out_threads = tiles * (a->cout / 8);
  if (a->x_is_wino_v) w.v = (u16*)a->x;  // V was written by the GroupNorm in front (ca_groupnorm_args.wino_v)
  else if (a->dtype == CA_BF16) hipLaunchKernelGGL((k_wino_in<CA_BF16>), dim3((unsigned)((in_threads + 255) / 256)), dim3(256), 0, st, w);
  else hipLaunchKernelGGL((k_wino_in<CA_F16>), dim3((unsigned)((in_threads + 255) / 256)), dim3(256), 0, st, w);
  // the sixteen transformed GEMMs as ONE launch of the 256 x 320 kernel: A = V [16 T, kc], weights U_f for the rows of group f
  GemmKParams q{};
  q.a = w.v;
  q.w = (const u16*)a->w_wino;
  q.c = mm;
  q.lda = kc;
  q.ldc = a->cout;
  q.a_bytes = desc_bytes(16 * tiles * kc);
  q.w_bytes = desc_bytes((int64_t)16 * a->cout * kc);
  q.m = (int)(16 * tiles);
  q.n = a->cout;
  q.c1 = kc;
  q.taps = 1;
  q.kc_tiles = kc / BK;
  q.rows_per_group = 1;
  q.alpha = 1.f, q.post = 1.f;
  q.splits = 1;
  q.w_group_rows = (int)tiles;
  q.w_group_stride = (unsigned)((int64_t)a->cout * kc * 2);
  const unsigned gemm_tiles = (unsigned)((q.m / 256) * (q.n / 320));
  int rc = ca_launch_gemm_pp(q, a->dtype, 0, 323, gemm_tiles, st);
  if (rc) return rc;
  if (a->dtype == CA_BF16) hipLaunchKernelGGL((k_wino_out<CA_BF16>), dim3((unsigned)((out_threads + 255) / 256)), dim3(256), 0, st, w);
  else hipLaunchKernelGGL((k_wino_out<CA_F16>), dim3((unsigned)((out_threads + 255) / 256)), dim3(256), 0, st, w);
  return CA_OK;
}

extern "C" int ca_pack_w_wino(const void* w, int32_t cout, int32_t cin, int32_t dtype, void* dst, void* stream) {
  CA_REQUIRE(w && dst, "ca_pack_w_wino: null operand");
  CA_REQUIRE(cout > 0 && cin > 0 && (dtype == CA_BF16 || dtype == CA_F16), "ca_pack_w_wino: cout=%d cin=%d dtype=%d", cout, cin, dtype);
  const int64_t n = (int64_t)cout * cin;
  if (dtype == CA_BF16) hipLaunchKernelGGL((k_pack_w_wino<CA_BF16>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u16*)w, (u16*)dst, cout, cin);
  else hipLaunchKernelGGL((k_pack_w_wino<CA_F16>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u16*)w, (u16*)dst, cout, cin);
  CA_CHECK_LAUNCH("ca_pack_w_wino");
  return CA_OK;
}

extern "C" int ca_conv3x3(const ca_conv_args* a, void* stream) {
  GemmKParams p{};
  int rc = conv_prepare(a, p);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (wino_taken(a)) {
    rc = launch_conv_wino(a, p, st);
    if (rc) return rc;
    CA_CHECK_LAUNCH("ca_conv3x3(winograd)");
    return CA_OK;
  }
  CA_REQUIRE(!a->x_is_wino_v, "ca_conv3x3: x_is_wino_v, but the Winograd route does not take these arguments (images=%d %dx%d cin=%d+%d cout=%d dtype=%d w_wino=%p "
             "workspace=%p of %lld bytes, needs %lld)", a->images, a->hin, a->win, a->cin1, a->cin2, a->cout, a->dtype, a->w_wino, a->workspace,
             (long long)a->workspace_bytes, (long long)wino_workspace_bytes(a));
  if (a->dtype == CA_BF16) launch_gemm<CA_BF16, 1>(p, st);
  else launch_gemm<CA_F16, 1>(p, st);
  CA_CHECK_LAUNCH("ca_conv3x3");
  return CA_OK;
}

// ---- which kernel would these arguments run?  (no launch, no device access: the pointers only have to be non-NULL where
// the launch requires them)
extern "C" int ca_gemm_plan_name(const ca_gemm_args* a, char* buf, int32_t len) {
  CA_REQUIRE(buf && len > 0, "ca_gemm_plan_name: buffer");
  GemmKParams p{};
  int rc = gemm_prepare(a, p);
  if (rc) return rc;
  plan_label(plan_gemm(p, 0), buf, len);
  return CA_OK;
}

extern "C" int ca_conv3x3_plan_name(const ca_conv_args* a, char* buf, int32_t len) {
  CA_REQUIRE(buf && len > 0, "ca_conv3x3_plan_name: buffer");
  GemmKParams p{};
  int rc = conv_prepare(a, p);
  if (rc) return rc;
  if (wino_taken(a)) {
    snprintf(buf, (size_t)len, "wino_pq256x320");
    return CA_OK;
  }
  plan_label(plan_gemm(p, 1), buf, len);
  return CA_OK;
}

// Flash-style attention for gfx950: O = softmax(Q K^T * scale) V with online softmax.
//
// Formulation (everything "transposed" so the softmax state is lane-local):
//   S^T tile  = mfma(A = K fragment [16 keys x 32 d], B = Q fragment [16 queries x 32 d])
//               -> lane (g = lane>>4, c = lane&15) holds S^T[key = 4g + r][query = c], r = 0..3.
//   A query is one lane COLUMN: its running max / sum are per-lane scalars (combined over the
//   four 16-lane groups with two xor-shuffles), and P^T is already in MFMA B-operand layout.
//   O^T tile  = mfma(A = V^T fragment [16 dv x 32 keys], B = P^T fragment [32 keys x 16 queries])
//               -> lane holds O^T[dv = 4g + r][query = c] -> 8-byte stores of 4 consecutive dv.
// K is staged row-major in LDS; V is staged TRANSPOSED (dv-major) with the key order of every
// 32-key chunk permuted so that the 8 keys an A-fragment lane needs are one 16-byte read.
// One kernel serves spatial self-attention, text/IP cross-attention (K/V shared by the frames of
// a batch element) and temporal attention (strided rows) via the batch/row strides.
#include "ca_common.h"
#include <stdlib.h>
#ifndef CA_ATTN_SETPRIO
#define CA_ATTN_SETPRIO 0  // measured: no gain on this kernel (1.67 vs 1.61 ms)
#endif

__attribute__((visibility("hidden"))) int ar_cu_count();  // (library-internal: not part of the C ABI)

namespace {

template <bool B>
struct BoolC {
  static constexpr bool value = B;
};

struct AttnKParams {
  const u16* q;
  const u16* k;
  const u16* v;
  u16* o;
  int64_t q_outer, q_inner, q_row;
  int64_t o_outer, o_inner, o_row;
  int64_t k_outer, k_inner, k_row;
  int inner_count, kv_inner_count, kv_div, kv_mod;
  int batches, heads, head_dim, nq, nk;
  int qblocks;
  float scale_log2;
  float out_scale;
  int accumulate;
  int sum_row;  // 1: V^T row `head_dim` is all ones, so the PV MFMA also produces the softmax row sums
  int causal;   // 1: key j is visible to query i only if j <= i (CLIP text encoder); generic kernel only
  const unsigned char* key_mask;  // per (batch, key): 0 = invisible (ca_attn_args.key_mask); generic kernel only
  int64_t key_mask_stride;
};

template <int DT, int DK32, int DV16, int QT, int NW, int KB, bool PF>
__global__ __launch_bounds__(NW * 64) void k_attn(AttnKParams p) {
  constexpr int NT = NW * 64;
  constexpr bool SETPRIO = CA_ATTN_SETPRIO;  // raise wave priority around the MFMA clusters (T5)
  constexpr int DKP = DK32 * 32;
  constexpr int DVP = DV16 * 16;
  constexpr int KLD = DKP + 8;  // K tile row stride (elements)
  constexpr int VLD = KB + 8;   // V^T tile row stride (elements)
  constexpr int KT = KB / 16;
  constexpr int KC = KB / 32;
  // PF: the next K/V tile's global loads are issued before the MFMA phase of the current one and
  // written to the OTHER LDS buffer after it (one barrier per tile, no exposed load latency).
  constexpr int TILE = KB * KLD + DVP * VLD;
  constexpr int KI = (KB * (DKP / 8) + NT - 1) / NT;         // K chunks staged per thread
  constexpr int VI = ((KB / 4) * (DVP / 8) + NT - 1) / NT;   // V key-quads staged per thread
  __shared__ __attribute__((aligned(16))) u16 smem[(PF ? 2 : 1) * TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;

  // XCD-aware order: the q-blocks of one (image, head) -- which all stream the same K/V -- run on ONE XCD and
  // share its L2 (round-robin dispatch spread them over 8 L2s: 1.39 GB fetched for 252 MB of q|k|v)
  unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = bid % p.qblocks;
  bid /= p.qblocks;
  const int head = bid % p.heads;
  const int z = bid / p.heads;

  const int zo = z / p.inner_count, zi = z - zo * p.inner_count;
  const int zk = (z / p.kv_div) % p.kv_mod;
  const int zko = zk / p.kv_inner_count, zki = zk - zko * p.kv_inner_count;
  const u16* qp = p.q + zo * p.q_outer + zi * p.q_inner + (int64_t)head * p.head_dim;
  u16* op = p.o + zo * p.o_outer + zi * p.o_inner + (int64_t)head * p.head_dim;
  const int64_t kvoff = zko * p.k_outer + zki * p.k_inner + (int64_t)head * p.head_dim;
  const u16* kp = p.k + kvoff;
  const u16* vp = p.v + kvoff;

  const int q0 = qb * (NW * QT * 16) + wid * (QT * 16);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // Q fragments (B operand): Q[q0 + t*16 + l15][kc*32 + g*8 .. +8]
  u32x4 qf[QT][DK32];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qi = q0 + t * 16 + l15;
#pragma unroll
    for (int kc = 0; kc < DK32; ++kc) {
      const int d = kc * 32 + g * 8;
      qf[t][kc] = (qi < p.nq && d < p.head_dim) ? ld16(qp + (int64_t)qi * p.q_row + d) : zero4;
    }
  }

  f32x4 oacc[QT][DV16];
  float mref[QT], lrun[QT];  // mref: reference maximum, already multiplied by scale*log2(e)
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    mref[t] = -INFINITY;
    lrun[t] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DV16; ++dt) oacc[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // Per-thread staging assignments are loop-invariant: source pointers, validity and LDS offsets are
  // computed once; per tile only `kv0 * k_row` is added (the softmax leaves little VALU headroom).
  u32x4 rk[KI], rv[VI][4];
  const u16* ksrc[KI];
  const u16* vsrc[VI];
  int kkey[KI], klds[KI], vkey[VI], vlds[VI], vone[VI];
  bool kok[KI], vok[VI], vitem[VI], kitem[KI];
#pragma unroll
  for (int u = 0; u < KI; ++u) {
    const int it = tid + u * NT;
    const int key = it / (DKP / 8), dc = it - key * (DKP / 8);
    kitem[u] = it < KB * (DKP / 8);
    kok[u] = kitem[u] && dc * 8 < p.head_dim;
    kkey[u] = key;
    klds[u] = key * KLD + dc * 8;
    ksrc[u] = kp + (int64_t)key * p.k_row + dc * 8;
  }
#pragma unroll
  for (int u = 0; u < VI; ++u) {
    const int it = tid + u * NT;
    const int quad = it / (DVP / 8), dc = it - quad * (DVP / 8);
    vitem[u] = it < (KB / 4) * (DVP / 8);
    vok[u] = vitem[u] && dc * 8 < p.head_dim;
    vkey[u] = quad * 4;
    vlds[u] = dc * 8 * VLD + (quad >> 3) * 32 + (quad & 3) * 8 + ((quad >> 2) & 1) * 4;
    vone[u] = p.sum_row ? p.head_dim - dc * 8 : -1;  // the all-ones row, if it falls in this chunk
    vsrc[u] = vp + (int64_t)quad * 4 * p.k_row + dc * 8;
  }
  auto load_tile = [&](int kv0) {
    const int64_t adv = (int64_t)kv0 * p.k_row;
#pragma unroll
    for (int u = 0; u < KI; ++u) rk[u] = (kok[u] && kv0 + kkey[u] < p.nk) ? ld16(ksrc[u] + adv) : zero4;
#pragma unroll
    for (int u = 0; u < VI; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        rv[u][j] = (vok[u] && kv0 + vkey[u] + j < p.nk) ? ld16(vsrc[u] + adv + (int64_t)j * p.k_row) : zero4;
    }
  };
  auto store_tile = [&](int buf) {
    u16* Ks = smem + buf * TILE;
    u16* Vts = Ks + KB * KLD;
    // K tile: [KB keys][DKP] row-major, zero padded
#pragma unroll
    for (int u = 0; u < KI; ++u)
      if (kitem[u]) st16(Ks + klds[u], rk[u]);
    // V^T tile: [DVP dv][KB keys (permuted inside every 32-key chunk)]
    // (a quad-major lane order would make these 8-byte writes bank-conflict-free, but it scatters
    // the global loads over 64 rows per instruction and measured slower: 1.71 vs 1.61 ms)
    const unsigned ones = DT == CA_F16 ? 0x3C003C00u : 0x3F803F80u;
#pragma unroll
    for (int u = 0; u < VI; ++u) {
      if (vitem[u]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int w = i >> 1;
          u32x2 o;
          if (i & 1) {
            o[0] = (rv[u][0][w] >> 16) | (rv[u][1][w] & 0xffff0000u);
            o[1] = (rv[u][2][w] >> 16) | (rv[u][3][w] & 0xffff0000u);
          } else {
            o[0] = (rv[u][0][w] & 0xffffu) | (rv[u][1][w] << 16);
            o[1] = (rv[u][2][w] & 0xffffu) | (rv[u][3][w] << 16);
          }
          if (i == vone[u]) o = (u32x2){ones, ones};
          *reinterpret_cast<u32x2*>(Vts + vlds[u] + i * VLD) = o;
        }
      }
    }
  };

  if (PF) {
    load_tile(0);
    store_tile(0);
    __syncthreads();
    if (KB < p.nk) load_tile(KB);
  }
  // One tile of the KV loop.  `tail_c` is a compile-time flag: full tiles carry no key masking at all
  // (hipcc if-converts a runtime `if (tail)` into ~115 compare/select instructions per tile).
  auto tile_body = [&](int kv0, int iter, auto tail_c) {
    constexpr bool TAIL = decltype(tail_c)::value;
    const int buf = PF ? (iter & 1) : 0;
    if (!PF) {
      __syncthreads();
      load_tile(kv0);
      store_tile(0);
      __syncthreads();
    }
    const u16* Ks = smem + buf * TILE;
    const u16* Vts = Ks + KB * KLD;

    // ---- S^T = K Q^T ----------------------------------------------------------------------
    f32x4 sacc[QT][KT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) sacc[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);
    // kc outer / kt inner: MFMAs that accumulate into the same S^T tile are KT*QT instructions apart
    // (back-to-back dependent 16x16x32 MFMAs stall ~2x: measured 29 instead of 16 cycles each)
#pragma unroll
    for (int kc = 0; kc < DK32; ++kc) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const u32x4 kf = ld16(Ks + (kt * 16 + l15) * KLD + kc * 32 + g * 8);
#pragma unroll
        for (int t = 0; t < QT; ++t) sacc[t][kt] = Elem<DT>::mfma(kf, qf[t][kc], sacc[t][kt]);
      }
    }
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);

    // ---- online softmax (per lane = per query column) and P^T fragments ---------------------
    // VALU budget matters as much as MFMA at head_dim 40: per score one fma + one v_exp + a share
    // of a max and of a pack; masking only in the tail tile; row sums come out of the PV MFMA via
    // the all-ones V^T row when there is a spare padded row (sum_row).
    // Deferred running maximum: the reference point mref (already in the exp2 domain) only moves when
    // some score exceeds it by more than 2^8, so rescaling O is a rare wave-uniform branch, the common
    // path has no cross-lane traffic, and p = 2^(s - mref) <= 256; the final division by the row sum
    // makes the result independent of the reference point.
    u32x4 pf[QT][KC];
    float mloc[QT];
    bool grow = false;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (TAIL) {  // (also every tile of a causal / key-masked launch)
        const int qlim = p.causal ? q0 + t * 16 + l15 : 0x7fffffff;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kv0 + kt * 16 + g * 4 + r;
            bool hide = key >= p.nk || key > qlim;
            if (p.key_mask && key < p.nk) hide |= p.key_mask[(int64_t)z * p.key_mask_stride + key] == 0;
            if (hide) sacc[t][kt][r] = -INFINITY;
          }
      }
      float m = vmax3(sacc[t][0][0], sacc[t][0][1], sacc[t][0][2]);
      m = vmax2(m, sacc[t][0][3]);
#pragma unroll
      for (int kt = 1; kt < KT; ++kt) {
        m = vmax3(m, sacc[t][kt][0], sacc[t][kt][1]);
        m = vmax3(m, sacc[t][kt][2], sacc[t][kt][3]);
      }
      mloc[t] = m;
      grow |= m * p.scale_log2 > mref[t] + 8.0f;
    }
    if (__builtin_amdgcn_ballot_w64(grow) != 0ull) {
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        const float mnew = fmaxf(mref[t], rowgroup_max(mloc[t]) * p.scale_log2);
        const float alpha = __builtin_amdgcn_exp2f(mref[t] - mnew);
        mref[t] = mnew;
        lrun[t] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DV16; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[t][dt][r] *= alpha;
      }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const float nmc = -mref[t];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sacc[t][kt][r] = __builtin_amdgcn_exp2f(fmaf(sacc[t][kt][r], p.scale_log2, nmc));
      if (!p.sum_row) {
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) psum += sacc[t][kt][r];
        lrun[t] += psum;  // per-lane partial; the four row groups are added in the epilogue
      }
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        pf[t][c][0] = pack2_prob<DT>(sacc[t][2 * c][0], sacc[t][2 * c][1]);
        pf[t][c][1] = pack2_prob<DT>(sacc[t][2 * c][2], sacc[t][2 * c][3]);
        pf[t][c][2] = pack2_prob<DT>(sacc[t][2 * c + 1][0], sacc[t][2 * c + 1][1]);
        pf[t][c][3] = pack2_prob<DT>(sacc[t][2 * c + 1][2], sacc[t][2 * c + 1][3]);
      }
    }

    // ---- O^T += V^T P^T -------------------------------------------------------------------
    if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int c = 0; c < KC; ++c) {
#pragma unroll
      for (int dt = 0; dt < DV16; ++dt) {
        const u32x4 vf = ld16(Vts + (dt * 16 + l15) * VLD + c * 32 + g * 8);
#pragma unroll
        for (int t = 0; t < QT; ++t) oacc[t][dt] = Elem<DT>::mfma(vf, pf[t][c], oacc[t][dt]);
      }
    }
    if (SETPRIO) __builtin_amdgcn_s_setprio(0);
    if (PF) {
      if (kv0 + KB < p.nk) store_tile(buf ^ 1);
      __syncthreads();
      if (kv0 + 2 * KB < p.nk) load_tile(kv0 + 2 * KB);
    }
  };
  {
    const int nfull = p.nk / KB;
    int iter = 0;
    if (p.causal || p.key_mask) {
      for (; iter < nfull; ++iter) tile_body(iter * KB, iter, BoolC<true>{});
    } else {
      for (; iter < nfull; ++iter) tile_body(iter * KB, iter, BoolC<false>{});
    }
    if (nfull * KB < p.nk) tile_body(nfull * KB, iter, BoolC<true>{});
  }

  // ---- epilogue: lane holds O[q = .. + l15][dv = dt*16 + 4g + r] ------------------------------
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qi = q0 + t * 16 + l15;
    float lsum = lrun[t];
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    if (p.sum_row) {  // row sums live in O^T[head_dim][q]: lanes of group (head_dim % 16) / 4, register 0
      float lv = 0.f;
#pragma unroll
      for (int dt = 0; dt < DV16; ++dt)
        if (dt == (p.head_dim >> 4)) lv = oacc[t][dt][0];
      lsum = __shfl(lv, (((p.head_dim & 15) >> 2) << 4) + l15);
    }
    if (qi >= p.nq) continue;
    const float inv = p.out_scale / lsum;
    u16* orow = op + (int64_t)qi * p.o_row;
#pragma unroll
    for (int dt = 0; dt < DV16; ++dt) {
      const int dv = dt * 16 + g * 4;
      if (dv >= p.head_dim) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = oacc[t][dt][r] * inv;
      if (p.accumulate) {
        u32x2 old = *reinterpret_cast<const u32x2*>(orow + dv);
        v[0] += Elem<DT>::to_f((u16)(old[0] & 0xffffu));
        v[1] += Elem<DT>::to_f((u16)(old[0] >> 16));
        v[2] += Elem<DT>::to_f((u16)(old[1] & 0xffffu));
        v[3] += Elem<DT>::to_f((u16)(old[1] >> 16));
      }
      u32x2 o;
      o[0] = pack2<DT>(v[0], v[1]);
      o[1] = pack2<DT>(v[2], v[3]);
      *reinterpret_cast<u32x2*>(orow + dv) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant for the long-sequence case (spatial self-attention, N >= 1024; head_dim <= 64).
// K and V tiles are copied global -> LDS row-major by `buffer_load_dwordx4 ... lds` (no VGPR round
// trip, no ds_write pass, no register transposition: staging was 30% of the register-staged kernel).
//   * rows are 128 B (64 elements); the 16-byte chunk is XOR-swizzled with (row>>1)&7 on the DMA
//     source side and on every read, as in ca_gemm.hip;
//   * chunks beyond head_dim are never transferred (exec-masked) -- the pad region is zeroed once;
//     keys beyond nk get an out-of-range offset -> the hardware writes zeros;
//   * V^T MFMA A-fragments come from the row-major V tile with the gfx950 transpose read
//     ds_read_b64_tr_b16: inside a 16-lane group lane p supplies the address of piece p of a
//     [4 keys][16 dv] block (row p/4, 4 columns at 4*(p%4)) and receives column p, i.e. 4 keys of one
//     dv (semantics probed on MI355X: tools/probe_tr.hip).  Two reads (keys 4g.. and 16+4g..) give the 8
//     keys of the fragment in exactly the order the P^T fragment holds them.
//   * two LDS buffers, the next tile's DMA is issued before the MFMA phase; one barrier per tile.
// FOLD (needs a free 16-byte pad chunk in the K tile too, i.e. head_dim + 8 <= 32*DK32): the softmax argument comes out
// of the QK^T MFMA ready-made.  Q is pre-multiplied by scale*log2(e) (fp32 multiply, one rounding), column head_dim of
// every K row holds 1 and the matching k-slot of the Q fragment holds -mref (the quantised reference maximum of that
// query), so S' = s*scale*log2(e) - mref and p = 2^S' with no per-score VALU besides the v_exp itself.
// K16 (round 3): head dims of 32 .. 44 (SD1.5's 40) needed a second, mostly empty 32-deep chunk for d = 32 .. 39 and the FOLD
// column -- 37% of the QK^T MFMA work multiplied zeros.  With K16 the chunk behind the DK32 full ones is 16 deep and runs
// on v_mfma_f32_16x16x16 (a lane holds k = 4g .. 4g+3: 8-byte fragment reads): QK^T costs 48 instead of 64 columns.
template <int DT, int DK32, int DV16, int QT, int NW, int KB, bool SR, bool FOLD, bool K16 = false>
__global__ __launch_bounds__(NW * 64, (DK32 == 1 && DV16 == 3 && SR && FOLD && K16) ? 4 : 1) void k_attn_dma(AttnKParams p) {  // (d = 40: four waves per SIMD, <= 128 VGPRs)
  // LDS row length (elements): 64 for head dims up to 64; 128 for 65..80 (two 32-deep chunks + the 16-deep one: SD1.5's d = 80 with
  // no padding at all -- round 4: the 1024-token self-attention of the 32x32-latent level ran on the generic kernel at 170 us)
  constexpr int ROW = (DK32 * 32 + (K16 ? 16 : 0)) > 64 ? 128 : 64;
  constexpr int CPR = ROW / 8;            // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;           // rows one wave-wide DMA instruction covers (8 / 4)
  // chunk swizzle, conflict-free for ds_read_b128's 16-lane groups: 128-byte rows (row >> 1) & 7, 256-byte rows row & 15
  auto swz = [](int row) { return ROW == 64 ? ((row >> 1) & 7) : (row & 15); };
  constexpr int KT = KB / 16, KC = KB / 32;
  constexpr int TILE = 2 * KB * ROW;      // K tile + V tile
  constexpr int GRP = KB / RPI / NW;      // DMA instructions per wave per operand and tile
  __shared__ __attribute__((aligned(16))) u16 smem[2 * TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;

  // XCD-aware order: the q-blocks of one (image, head) -- which all stream the same K/V -- run on ONE XCD and
  // share its L2 (round-robin dispatch spread them over 8 L2s: 1.39 GB fetched for 252 MB of q|k|v)
  unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = bid % p.qblocks;
  bid /= p.qblocks;
  const int head = bid % p.heads;
  const int z = bid / p.heads;
  const int zo = z / p.inner_count, zi = z - zo * p.inner_count;
  const int zk = (z / p.kv_div) % p.kv_mod;
  const int zko = zk / p.kv_inner_count, zki = zk - zko * p.kv_inner_count;
  const u16* qp = p.q + zo * p.q_outer + zi * p.q_inner + (int64_t)head * p.head_dim;
  u16* op = p.o + zo * p.o_outer + zi * p.o_inner + (int64_t)head * p.head_dim;
  const int64_t kvoff = zko * p.k_outer + zki * p.k_inner + (int64_t)head * p.head_dim;
  const u16* kp = p.k + kvoff;
  const u16* vp = p.v + kvoff;
  const unsigned kv_bytes = (unsigned)(((int64_t)(p.nk - 1) * p.k_row + p.head_dim) * 2);
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kp, 0, kv_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)vp, 0, kv_bytes, 0x00020000);

  const int q0 = qb * (NW * QT * 16) + wid * (QT * 16);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // zero both buffers once: the pad chunks (d >= head_dim) are never written again.  SR (head_dim <
  // 16*DV16): the first pad chunk of every V row holds ones instead, so row head_dim.. of O^T
  // accumulates the softmax denominator in the MFMA (sum of the ROUNDED probabilities) for free.
  for (int i = tid; i < 2 * TILE / 8; i += NW * 64) st16(smem + i * 8, zero4);
  if (SR) {
    __syncthreads();
    const unsigned one2 = DT == CA_F16 ? 0x3C003C00u : 0x3F803F80u;
    const u32x4 ones4 = {one2, one2, one2, one2};
    const int cs = p.head_dim >> 3;
    for (int i = tid; i < 2 * KB; i += NW * 64) {
      const int buf = i / KB, r = i - buf * KB;
      st16(smem + buf * TILE + KB * ROW + r * ROW + ((cs ^ swz(r)) << 3), ones4);
      if (FOLD) smem[buf * TILE + r * ROW + ((cs ^ swz(r)) << 3)] = (u16)(one2 & 0xffffu);  // K[r][head_dim] = 1
    }
  }

  u32x4 qf[QT][DK32];
  u32x2 qf16[QT];  // K16: d = 32 * DK32 + 4g .. +3
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qi = q0 + t * 16 + l15;
#pragma unroll
    for (int kc = 0; kc < DK32; ++kc) {
      const int d = kc * 32 + g * 8;
      qf[t][kc] = (qi < p.nq && d < p.head_dim) ? ld16(qp + (int64_t)qi * p.q_row + d) : zero4;
      if (FOLD) {
        float f[8];
        unpack8<DT>(qf[t][kc], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] *= p.scale_log2;
        qf[t][kc] = pack8<DT>(f);
      }
    }
    qf16[t] = (u32x2){0u, 0u};
    if (K16) {
      const int d = DK32 * 32 + g * 4;
      if (qi < p.nq && d < p.head_dim) qf16[t] = *reinterpret_cast<const u32x2*>(qp + (int64_t)qi * p.q_row + d);
      if (FOLD) {
        const float f0 = Elem<DT>::to_f((u16)(qf16[t][0] & 0xffffu)) * p.scale_log2, f1 = Elem<DT>::to_f((u16)(qf16[t][0] >> 16)) * p.scale_log2;
        const float f2 = Elem<DT>::to_f((u16)(qf16[t][1] & 0xffffu)) * p.scale_log2, f3 = Elem<DT>::to_f((u16)(qf16[t][1] >> 16)) * p.scale_log2;
        qf16[t][0] = pack2<DT>(f0, f1);
        qf16[t][1] = pack2<DT>(f2, f3);
      }
    }
  }
  // FOLD: the lanes of row group fold_g hold k-slot head_dim of k32 chunk fold_kc, element 0 of their fragment
  // (K16 and head_dim >= 32 * DK32: the 16-deep chunk's row group (head_dim - 32 * DK32) / 4, element 0)
  const int fold_kc = p.head_dim >> 5, fold_g = (p.head_dim & 31) >> 3;
  const int fold_g16 = (p.head_dim - DK32 * 32) >> 2;
  auto set_qref = [&](int t, float mnew) {
    if (K16 && p.head_dim >= DK32 * 32) {
      if (g == fold_g16) qf16[t][0] = (qf16[t][0] & 0xffff0000u) | (unsigned)Elem<DT>::from_f(-mnew);
      return;
    }
#pragma unroll
    for (int kc = 0; kc < DK32; ++kc)
      if (kc == fold_kc && g == fold_g) qf[t][kc][0] = (qf[t][kc][0] & 0xffff0000u) | (unsigned)Elem<DT>::from_f(-mnew);
  };

  // DMA lane assignment: 8 rows x 8 chunks per wave instruction
  const int r8 = lane / CPR, cpos = lane % CPR;
  int d_row[GRP];
  unsigned d_off[GRP];
  bool d_ok[GRP];
#pragma unroll
  for (int i = 0; i < GRP; ++i) {
    const int row = (wid * GRP + i) * RPI + r8;
    const int chunk = cpos ^ swz(row);
    d_row[i] = row;
    d_ok[i] = chunk * 8 < p.head_dim;
    d_off[i] = (unsigned)((int64_t)row * p.k_row * 2 + chunk * 16);
  }
  const unsigned tile_adv = (unsigned)((int64_t)KB * p.k_row * 2);
  auto stage = [&](int kv0, int buf) {
    u16* Ks = smem + buf * TILE;
    u16* Vs = Ks + KB * ROW;
    const unsigned adv = (unsigned)(kv0 / KB) * tile_adv;
#pragma unroll
    for (int i = 0; i < GRP; ++i) {
      if (d_ok[i]) {
        const unsigned off = (kv0 + d_row[i] < p.nk) ? d_off[i] + adv : 0xFFFFFFF0u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_k, (__attribute__((address_space(3))) void*)(Ks + (wid * GRP + i) * RPI * ROW), 16, off, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_v, (__attribute__((address_space(3))) void*)(Vs + (wid * GRP + i) * RPI * ROW), 16, off, 0, 0, 0);
      }
    }
  };

  f32x4 oacc[QT][DV16];
  float mref[QT], lrun[QT];  // mref: reference maximum, already multiplied by scale*log2(e)
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    mref[t] = FOLD ? 0.f : -INFINITY;  // FOLD: the Q slot starts at -0; tile 0 always re-bases
    lrun[t] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DV16; ++dt) oacc[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  typedef short v4s __attribute__((ext_vector_type(4)));

  __syncthreads();  // zero fill complete before the first transfer lands
  stage(0, 0);
  __syncthreads();  // (vmcnt(0) + barrier)

  auto tile_body = [&](int kv0, int iter, auto tail_c, auto buf_c) {
    constexpr bool TAIL = decltype(tail_c)::value;
    // buf_c: the LDS buffer of this tile as a compile-time constant (the main loop below runs tiles in pairs) -- every fragment address is then
    // a loop-invariant base plus an immediate offset, instead of ~20 VALU instructions per tile that add the buffer's offset (the kernel is bound
    // by VALU issue: DESIGN.md section 9); -1 = take it from the iteration count
    constexpr int BUFC = decltype(buf_c)::value;
    const int buf = BUFC >= 0 ? BUFC : (iter & 1);
    if (kv0 + KB < p.nk) stage(kv0 + KB, buf ^ 1);
    const u16* Ks = smem + buf * TILE;
    const u16* Vs = Ks + KB * ROW;

    f32x4 sacc[QT][KT];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) sacc[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < DK32; ++kc) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int row = kt * 16 + l15;
        const u32x4 kf = ld16(Ks + row * ROW + (((kc * 4 + g) ^ swz(row)) << 3));
#pragma unroll
        for (int t = 0; t < QT; ++t) sacc[t][kt] = Elem<DT>::mfma(kf, qf[t][kc], sacc[t][kt]);
      }
    }
    if (K16) {
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int row = kt * 16 + l15;
        const u32x2 kf = *reinterpret_cast<const u32x2*>(Ks + row * ROW + (((DK32 * 4 + (g >> 1)) ^ swz(row)) << 3) + (g & 1) * 4);
#pragma unroll
        for (int t = 0; t < QT; ++t) sacc[t][kt] = Elem<DT>::mfma16(kf, qf16[t], sacc[t][kt]);
      }
    }

    u32x4 pf[QT][KC];
    // Softmax in the exp2 domain with a DEFERRED running maximum (the reference point mref only moves
    // when some score exceeds it by more than 2^8, so the rescale of O is a rare wave-uniform branch and
    // p = 2^(s - mref) <= 256 stays well inside fp16/bf16 range; the final division by the row sum makes
    // the result independent of the reference point).  No cross-lane traffic on the common path: each
    // lane compares its own 16 scores with mref, the ballot ORs the lanes.
    float mloc[QT];
    bool grow = FOLD && iter == 0;
    float shift[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      shift[t] = 0.f;
      if (TAIL) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kv0 + kt * 16 + g * 4 + r >= p.nk) sacc[t][kt][r] = -INFINITY;
      }
      float m = vmax3(sacc[t][0][0], sacc[t][0][1], sacc[t][0][2]);
      m = vmax3(m, sacc[t][0][3], sacc[t][1][0]);
#pragma unroll
      for (int kt = 1; kt < KT; ++kt) {
        m = vmax3(m, sacc[t][kt][1], sacc[t][kt][2]);
        if (kt + 1 < KT) m = vmax3(m, sacc[t][kt][3], sacc[t][kt + 1][0]);
        else m = vmax2(m, sacc[t][kt][3]);
      }
      mloc[t] = m;
      if (FOLD) grow |= m > 8.0f;  // scores are already relative to mref
      else grow |= m * p.scale_log2 > mref[t] + 8.0f;
    }
    const bool rebase = __builtin_amdgcn_ballot_w64(grow) != 0ull;
    if (rebase) {
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        float mnew;
        if (FOLD) {
          const float mx = rowgroup_max(mloc[t]);
          // the new reference must be representable in the activation type: it travels in the Q fragment
          mnew = Elem<DT>::to_f(Elem<DT>::from_f(mref[t] + (iter == 0 ? mx : fmaxf(mx, 0.f))));
          shift[t] = mnew - mref[t];
          set_qref(t, mnew);
        } else {
          mnew = fmaxf(mref[t], rowgroup_max(mloc[t]) * p.scale_log2);
        }
        const float alpha = __builtin_amdgcn_exp2f(mref[t] - mnew);
        mref[t] = mnew;
        if (!SR) lrun[t] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DV16; ++dt)
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[t][dt][r] *= alpha;
      }
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (FOLD) {
        if (rebase) {  // this tile's scores are still relative to the previous reference
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[t][kt][r] = __builtin_amdgcn_exp2f(sacc[t][kt][r] - shift[t]);
        } else {
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[t][kt][r] = __builtin_amdgcn_exp2f(sacc[t][kt][r]);
        }
      } else {
        const float nmc = -mref[t];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc[t][kt][r] = __builtin_amdgcn_exp2f(fmaf(sacc[t][kt][r], p.scale_log2, nmc));
      }
      if (!SR) {
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) psum += sacc[t][kt][r];
        lrun[t] += psum;  // per-lane partial; the four row groups are added after the last tile
      }
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        pf[t][c][0] = pack2_prob<DT>(sacc[t][2 * c][0], sacc[t][2 * c][1]);
        pf[t][c][1] = pack2_prob<DT>(sacc[t][2 * c][2], sacc[t][2 * c][3]);
        pf[t][c][2] = pack2_prob<DT>(sacc[t][2 * c + 1][0], sacc[t][2 * c + 1][1]);
        pf[t][c][3] = pack2_prob<DT>(sacc[t][2 * c + 1][2], sacc[t][2 * c + 1][3]);
      }
    }

    // O^T += V^T P^T; V^T fragments by transpose reads of the row-major V tile
#pragma unroll
    for (int c = 0; c < KC; ++c) {
#pragma unroll
      for (int dt = 0; dt < DV16; ++dt) {
        const int col = dt * 16 + 4 * (l15 & 3);
        const int r0 = c * 32 + 4 * g + (l15 >> 2), r1 = r0 + 16;
        const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s*)(Vs + r0 * ROW + (((col >> 3) ^ swz(r0)) << 3) + (col & 7)));
        const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) v4s*)(Vs + r1 * ROW + (((col >> 3) ^ swz(r1)) << 3) + (col & 7)));
        const u32x2 lo2 = __builtin_bit_cast(u32x2, lo), hi2 = __builtin_bit_cast(u32x2, hi);
        const u32x4 vf = {lo2[0], lo2[1], hi2[0], hi2[1]};
#pragma unroll
        for (int t = 0; t < QT; ++t) oacc[t][dt] = Elem<DT>::mfma(vf, pf[t][c], oacc[t][dt]);
      }
    }
    __syncthreads();  // next tile landed (vmcnt(0)) and this tile is no longer read
  };
  {
    const int nfull = p.nk / KB;
    int iter = 0;
    for (; iter + 1 < nfull; iter += 2) {
      tile_body(iter * KB, iter, BoolC<false>{}, std::integral_constant<int, 0>{});
      tile_body((iter + 1) * KB, iter + 1, BoolC<false>{}, std::integral_constant<int, 1>{});
    }
    for (; iter < nfull; ++iter) tile_body(iter * KB, iter, BoolC<false>{}, std::integral_constant<int, -1>{});
    if (nfull * KB < p.nk) tile_body(nfull * KB, iter, BoolC<true>{}, std::integral_constant<int, -1>{});
  }

#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int qi = q0 + t * 16 + l15;
    float lsum;
    if (SR) {  // the ones column is row head_dim of O^T: tile head_dim/16, row group (head_dim%16)/4, r = 0
      const int hd = p.head_dim;
      lsum = __shfl(oacc[t][DV16 - 1][0], ((hd & 15) >> 2) * 16 + l15);
    } else {
      lsum = lrun[t];
      lsum += __shfl_xor(lsum, 16);
      lsum += __shfl_xor(lsum, 32);
    }
    if (qi >= p.nq) continue;
    const float inv = p.out_scale / lsum;
    u16* orow = op + (int64_t)qi * p.o_row;
#pragma unroll
    for (int dt = 0; dt < DV16; ++dt) {
      const int dv = dt * 16 + g * 4;
      if (dv >= p.head_dim) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = oacc[t][dt][r] * inv;
      if (p.accumulate) {
        u32x2 old = *reinterpret_cast<const u32x2*>(orow + dv);
        v[0] += Elem<DT>::to_f((u16)(old[0] & 0xffffu));
        v[1] += Elem<DT>::to_f((u16)(old[0] >> 16));
        v[2] += Elem<DT>::to_f((u16)(old[1] & 0xffffu));
        v[3] += Elem<DT>::to_f((u16)(old[1] >> 16));
      }
      u32x2 o;
      o[0] = pack2<DT>(v[0], v[1]);
      o[1] = pack2<DT>(v[2], v[3]);
      *reinterpret_cast<u32x2*>(orow + dv) = o;
    }
  }
}

// Which kernel a launch takes.  ONE function decides (ca_attention_plan_name reports it, the launchers switch on it), so a
// parity test can assert that the headline shapes really ran on the kernels the design names.
enum AttnPlan { AP_SHORT, AP_DMA80, AP_DMA40_K16, AP_DMA_FOLD, AP_DMA_SR, AP_DMA, AP_TINY16, AP_TINY32, AP_GENERIC };

static bool attn_short_eligible(const AttnKParams& p);

static AttnPlan attn_plan(const AttnKParams& p) {
  const int d = p.head_dim;
  if (attn_short_eligible(p)) return AP_SHORT;
  const bool dma_ok = !p.causal && !p.key_mask && p.nk >= 256 && p.k_row % 8 == 0 &&
                      ((int64_t)(p.nk - 1) * p.k_row + p.head_dim) * 2 < (int64_t)0xFFFFFF00ll;
  static const int dma80_env = CA_KNOB("CA_ATTN_DMA80", 1);
  // d = 80 on the LDS-DMA kernel: 32 + 32 + 16 deep, 256-byte LDS rows (no free pad column: neither the ones column nor the folded maximum)
  if (dma80_env && d > 64 && d <= 80 && d % 16 == 0 && dma_ok) return AP_DMA80;
  if (p.nq <= 16 && p.nk <= 32 && !p.causal && !p.key_mask) return AP_TINY16;
  if (p.nq <= 32 && p.nk <= 32 && !p.causal && !p.key_mask) return AP_TINY32;
  static const int dma_env = CA_KNOB("CA_ATTN_DMA", 1);
  if (d <= 64 && dma_env && dma_ok) {
    const int dv16 = d <= 32 ? 2 : d <= 48 ? 3 : 4, dk32 = d <= 32 ? 1 : 2;
    // the ones column needs a free, 16-byte aligned pad chunk inside the last 16-wide dv tile
    static const int sr_env = CA_KNOB("CA_ATTN_SR", 1);
    const bool sr = sr_env && d % 8 == 0 && d / 16 == dv16 - 1;
    static const int fold_env = CA_KNOB("CA_ATTN_FOLD", 1);
    const bool fold = sr && fold_env && d + 8 <= dk32 * 32;
    static const int k16_env = CA_KNOB("CA_ATTN_K16", 1);
    if (fold && k16_env && dk32 == 2 && dv16 == 3 && d >= 32 && d <= 44) return AP_DMA40_K16;  // head_dim 40: 32 + 16 instead of 64 deep
    return fold ? AP_DMA_FOLD : sr ? AP_DMA_SR : AP_DMA;
  }
  return AP_GENERIC;
}

static const char* attn_plan_label(AttnPlan pl) {
  switch (pl) {
    case AP_SHORT: return "attn_short";
    case AP_DMA80: return "attn_dma80";
    case AP_DMA40_K16: return "attn_dma40";
    case AP_DMA_FOLD: return "attn_dma_fold";
    case AP_DMA_SR: return "attn_dma_sr";
    case AP_DMA: return "attn_dma";
    case AP_TINY16: return "attn_tiny16";
    case AP_TINY32: return "attn_tiny32";
    default: return "attn_generic";
  }
}

template <int DT, int DK32, int DV16>
void launch_attn_d(const AttnKParams& p0, AttnPlan plan, hipStream_t st) {
  AttnKParams p = p0;
  if (plan == AP_TINY16) {
    p.qblocks = 1;
    hipLaunchKernelGGL((k_attn<DT, DK32, DV16, 1, 1, 32, false>), dim3((unsigned)(p.batches * p.heads)), dim3(64), 0, st, p);
  } else if (plan == AP_TINY32) {
    p.qblocks = 1;
    hipLaunchKernelGGL((k_attn<DT, DK32, DV16, 2, 1, 32, false>), dim3((unsigned)(p.batches * p.heads)), dim3(64), 0, st, p);
  } else {
    // register prefetch of the next K/V tile only where the staging registers fit (head_dim <= 96)
    static const int pf_env = CA_KNOB("CA_ATTN_PF", 1);  // tuning knob
    p.qblocks = ceil_div_i(p.nq, 128);
    const dim3 grid((unsigned)(p.qblocks * p.batches * p.heads));
    if constexpr (DK32 <= 2) {
      if (plan == AP_DMA40_K16) {
        if constexpr (DK32 == 2 && DV16 == 3) hipLaunchKernelGGL((k_attn_dma<DT, 1, 3, 2, 4, 64, true, true, true>), grid, dim3(256), 0, st, p);
        return;
      }
      if (plan == AP_DMA_FOLD) { hipLaunchKernelGGL((k_attn_dma<DT, DK32, DV16, 2, 4, 64, true, true>), grid, dim3(256), 0, st, p); return; }
      if (plan == AP_DMA_SR) { hipLaunchKernelGGL((k_attn_dma<DT, DK32, DV16, 2, 4, 64, true, false>), grid, dim3(256), 0, st, p); return; }
      if (plan == AP_DMA) { hipLaunchKernelGGL((k_attn_dma<DT, DK32, DV16, 2, 4, 64, false, false>), grid, dim3(256), 0, st, p); return; }
    }
    static const int var_env = CA_KNOB("CA_ATTN_VAR", 0);  // experiments (d <= 48 only)
    if (DK32 == 2 && DV16 == 3 && var_env == 1) {  // KB = 128
      hipLaunchKernelGGL((k_attn<DT, 2, 3, 2, 4, 128, true>), grid, dim3(256), 0, st, p);
    } else if (DK32 == 2 && DV16 == 3 && var_env == 2) {  // 8 waves, 256 queries per block
      p.qblocks = ceil_div_i(p.nq, 256);
      hipLaunchKernelGGL((k_attn<DT, 2, 3, 2, 8, 64, true>), dim3((unsigned)(p.qblocks * p.batches * p.heads)), dim3(512), 0, st, p);
    } else if (DK32 == 2 && DV16 == 3 && var_env == 3) {  // 1 q-tile per wave, 64 queries per block
      p.qblocks = ceil_div_i(p.nq, 64);
      hipLaunchKernelGGL((k_attn<DT, 2, 3, 1, 4, 64, true>), dim3((unsigned)(p.qblocks * p.batches * p.heads)), dim3(256), 0, st, p);
    } else if (DK32 == 2 && DV16 == 3 && var_env == 4) {  // 8 waves + KB 128
      p.qblocks = ceil_div_i(p.nq, 256);
      hipLaunchKernelGGL((k_attn<DT, 2, 3, 2, 8, 128, true>), dim3((unsigned)(p.qblocks * p.batches * p.heads)), dim3(512), 0, st, p);
    } else if (DK32 <= 3 && pf_env) hipLaunchKernelGGL((k_attn<DT, DK32, DV16, 2, 4, 64, (DK32 <= 3)>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_attn<DT, DK32, DV16, 2, 4, 64, false>), grid, dim3(256), 0, st, p);
  }
}


// ------------------------------------------------------------------------------------------------------------------------------
// Text cross-attention (round 4): nk = 77 keys, 8 heads of 40 / 80, thousands of queries per image.
//
// The generic kernel above gives a block 128 queries of ONE head and stages K / V through LDS per block: 8192 blocks at the
// 64x64-latent level, each a chain of dependent global round trips (Q, K/V tile 0, K/V tile 1, epilogue) around a few hundred
// MFMAs -- 100 us for 168 MB (64x64 latents), 107 us for 84 MB (32x32): latency, not bandwidth.  Here:
//   * a block is 8 waves = the 8 HEADS of the same query rows (every 128-byte line of q and o is used whole while it is hot);
//   * K and V of the wave's head live in REGISTERS for the whole launch (77 keys: five 16-key tiles): K as A fragments
//     pre-multiplied by scale * log2(e), V as V^T A fragments gathered once with the key order of a 32-key chunk permuted to
//     the order the S^T accumulators hold it (keys 4g + r of tile 2c, then of tile 2c + 1) -- no LDS, no barriers, waves free;
//   * a block walks QC query tiles per item (persistent over items, K / V reloaded only when the text batch changes), Q
//     fragments two tiles ahead;
//   * per 16-query tile: S^T = K Q^T (16x16x32 + one 16x16x16 for d = 32..39 / 64..79), one max / exp2 / sum over the
//     lane's 20 scores (two v_permlane swaps each for the four row groups), O^T = V^T P^T, 8-byte stores.
template <int DT, int D, int QC>
__global__ __launch_bounds__(512, 2) void k_attn_short(AttnKParams p, int chunks, int items) {
  constexpr int C32 = D / 32;        // full 32-deep chunks of the head dimension (1, 2); the rest (8 / 16 wide) is one 16-deep chunk
  constexpr int DV = (D + 15) / 16;  // d_v tiles (3, 5)
  constexpr int KT = 5;              // 16-key tiles: nk <= 80
  constexpr unsigned OOB_V = 0x80000000u;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int head = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;

  u32x4 kf[KT][C32], vf[DV][2];
  u32x2 kf16[KT], vf16[DV];
  int cur_zk = -1;

  const int per = (items + (int)gridDim.x - 1) / (int)gridDim.x;
  const int i0 = (int)blockIdx.x * per, i1 = i0 + per < items ? i0 + per : items;
  for (int item = i0; item < i1; ++item) {
    const int z = item / chunks, ch = item - z * chunks;
    const int zo = z / p.inner_count, zi = z - zo * p.inner_count;
    const int zk = (z / p.kv_div) % p.kv_mod;
    if (zk != cur_zk) {  // (block-uniform)
      cur_zk = zk;
      const int zko = zk / p.kv_inner_count, zki = zk - zko * p.kv_inner_count;
      const int64_t kvoff = zko * p.k_outer + zki * p.k_inner + (int64_t)head * D;
      const u16* kp = p.k + kvoff;
      const u16* vp = p.v + kvoff;
      const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        const int key = kt * 16 + l15;
#pragma unroll
        for (int c = 0; c < C32; ++c) {
          u32x4 v = key < p.nk ? ld16(kp + (int64_t)key * p.k_row + c * 32 + g * 8) : zero4;
          float f[8];
          unpack8<DT>(v, f);
#pragma unroll
          for (int j = 0; j < 8; ++j) f[j] *= p.scale_log2;
          kf[kt][c] = pack8<DT>(f);
        }
        const int d16 = C32 * 32 + g * 4;
        u32x2 w = {0u, 0u};
        if (key < p.nk && d16 < D) w = *reinterpret_cast<const u32x2*>(kp + (int64_t)key * p.k_row + d16);
        kf16[kt] = (u32x2){pack2<DT>(Elem<DT>::to_f((u16)(w[0] & 0xffffu)) * p.scale_log2, Elem<DT>::to_f((u16)(w[0] >> 16)) * p.scale_log2),
                           pack2<DT>(Elem<DT>::to_f((u16)(w[1] & 0xffffu)) * p.scale_log2, Elem<DT>::to_f((u16)(w[1] >> 16)) * p.scale_log2)};
      }
      // V^T fragments: row d_v = 16 j + l15; k-slot e of 32-key chunk c is key 32 c + 4 g + e (e < 4) / 32 c + 16 + 4 g + (e - 4)
#pragma unroll
      for (int j = 0; j < DV; ++j) {
        const int dv = 16 * j + l15;
        const bool dok = dv < D;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          unsigned w[4];
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            const int k0 = 32 * c + (e2 >> 1) * 16 + 4 * g + (e2 & 1) * 2;
            const unsigned a = (dok && k0 < p.nk) ? (unsigned)vp[(int64_t)k0 * p.k_row + dv] : 0u;
            const unsigned b = (dok && k0 + 1 < p.nk) ? (unsigned)vp[(int64_t)(k0 + 1) * p.k_row + dv] : 0u;
            w[e2] = a | (b << 16);
          }
          vf[j][c] = (u32x4){w[0], w[1], w[2], w[3]};
        }
        unsigned w2[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          const int k0 = 64 + 4 * g + 2 * e2;
          const unsigned a = (dok && k0 < p.nk) ? (unsigned)vp[(int64_t)k0 * p.k_row + dv] : 0u;
          const unsigned b = (dok && k0 + 1 < p.nk) ? (unsigned)vp[(int64_t)(k0 + 1) * p.k_row + dv] : 0u;
          w2[e2] = a | (b << 16);
        }
        vf16[j] = (u32x2){w2[0], w2[1]};
      }
    }

    const u16* qp = p.q + zo * p.q_outer + zi * p.q_inner + (int64_t)head * D;
    u16* op = p.o + zo * p.o_outer + zi * p.o_inner + (int64_t)head * D;
    const __amdgpu_buffer_rsrc_t rs_q = __builtin_amdgcn_make_buffer_rsrc((void*)qp, 0, (unsigned)(((int64_t)(p.nq - 1) * p.q_row + D) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)op, 0, (unsigned)(((int64_t)(p.nq - 1) * p.o_row + D) * 2), 0x00020000);
    const int t0 = ch * QC;
    u32x4 qf[3][C32];
    u32x2 qf16[3];
    auto q_load = [&](int tt, int buf) __attribute__((always_inline)) {
      const int qi = (t0 + tt) * 16 + l15;
      const unsigned ro = qi < p.nq ? (unsigned)qi * (unsigned)p.q_row * 2u : OOB_V;
#pragma unroll
      for (int c = 0; c < C32; ++c) qf[buf][c] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_q, ro + (unsigned)(c * 32 + g * 8) * 2u, 0, 0));
      const int d16 = C32 * 32 + g * 4;
      qf16[buf] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_q, d16 < D ? ro + (unsigned)d16 * 2u : OOB_V, 0, 0));
    };
    q_load(0, 0);
    q_load(1, 1);
#pragma unroll
    for (int tt = 0; tt < QC; ++tt) {
      const int buf = tt % 3;
      if (tt + 2 < QC) q_load(tt + 2, (tt + 2) % 3);
      if ((t0 + tt) * 16 >= p.nq) continue;  // (wave-uniform)
      // Dependent MFMAs of DIFFERENT shapes are kept whole groups apart (sched_barrier pins the order): hipcc places too few wait
      // states between a 16x16x16 and a 16x16x32 that accumulates onto it -- rows 2, 3 of the small one's result arrived after
      // the big one had read them (measured: d_v = 2, 3 mod 4 of every output lost the keys 64.. and kept the previous tile's sum).
      f32x4 s[KT];
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) s[kt] = Elem<DT>::mfma(kf[kt][0], qf[buf][0], (f32x4){0.f, 0.f, 0.f, 0.f});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 1; c < C32; ++c) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) s[kt] = Elem<DT>::mfma(kf[kt][c], qf[buf][c], s[kt]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) s[kt] = Elem<DT>::mfma16(kf16[kt], qf16[buf], s[kt]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (64 + 4 * g + r >= p.nk) s[4][r] = -INFINITY;  // (64 < nk <= 80: only the last tile is ragged)
      float m = vmax3(s[0][0], s[0][1], s[0][2]);
      m = vmax3(m, s[0][3], s[1][0]);
#pragma unroll
      for (int kt = 1; kt < KT; ++kt) {
        m = vmax3(m, s[kt][1], s[kt][2]);
        if (kt + 1 < KT) m = vmax3(m, s[kt][3], s[kt + 1][0]);
        else m = vmax2(m, s[kt][3]);
      }
      m = rowgroup_max(m);
      float l = 0.f;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[kt][r] = __builtin_amdgcn_exp2f(s[kt][r] - m);
          l += s[kt][r];
        }
      const float inv = p.out_scale * __builtin_amdgcn_rcpf(rowgroup_sum(l));
      u32x4 pf[2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
        pf[c] = (u32x4){pack2_prob<DT>(s[2 * c][0], s[2 * c][1]), pack2_prob<DT>(s[2 * c][2], s[2 * c][3]), pack2_prob<DT>(s[2 * c + 1][0], s[2 * c + 1][1]),
                        pack2_prob<DT>(s[2 * c + 1][2], s[2 * c + 1][3])};
      const u32x2 p16 = {pack2_prob<DT>(s[4][0], s[4][1]), pack2_prob<DT>(s[4][2], s[4][3])};
      const int qi = (t0 + tt) * 16 + l15;
      const unsigned ro = qi < p.nq ? (unsigned)qi * (unsigned)p.o_row * 2u : OOB_V;
      f32x4 o[DV];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < DV; ++j) o[j] = Elem<DT>::mfma(vf[j][0], pf[0], (f32x4){0.f, 0.f, 0.f, 0.f});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < DV; ++j) o[j] = Elem<DT>::mfma(vf[j][1], pf[1], o[j]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < DV; ++j) o[j] = Elem<DT>::mfma16(vf16[j], p16, o[j]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < DV; ++j) {
        const int dv = 16 * j + 4 * g;
        __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack2<DT>(o[j][0] * inv, o[j][1] * inv), pack2<DT>(o[j][2] * inv, o[j][3] * inv)}, rs_o,
                                              dv < D ? ro + (unsigned)dv * 2u : OOB_V, 0, 0);
      }
    }
  }
}

// 1: this launch is the text cross-attention k_attn_short takes
static bool attn_short_eligible(const AttnKParams& p) {
  static const int env = CA_KNOB("CA_ATTN_SHORT", 1);
  if (!env) return false;
  if (p.heads != 8 || (p.head_dim != 40 && p.head_dim != 80)) return false;
  if (p.nk <= 64 || p.nk > 80 || p.nq < 256 || p.accumulate || p.causal || p.key_mask) return false;
  if (p.q_row % 8 || p.o_row % 4 || p.k_row % 4) return false;
  const int64_t lim = 0x7FFFFF00ll;
  return ((int64_t)(p.nq - 1) * p.q_row + p.head_dim) * 2 < lim && ((int64_t)(p.nq - 1) * p.o_row + p.head_dim) * 2 < lim;
}

template <int DT>
static void launch_attn_short(const AttnKParams& p, hipStream_t st) {
  const int cus = ar_cu_count();  // (per device, ca_gemm_ar.hip)
  const int qtiles = ceil_div_i(p.nq, 16);
  // query tiles per item: whole items per block, about one item per CU or more
  const int qc = (int64_t)p.batches * ceil_div_i(qtiles, 16) >= cus ? 16 : 8;
  const int chunks = ceil_div_i(qtiles, qc);
  const int items = p.batches * chunks;
  const unsigned grid = (unsigned)(items < cus ? items : cus);
  if (p.head_dim == 40) {
    if (qc == 16) hipLaunchKernelGGL((k_attn_short<DT, 40, 16>), dim3(grid), dim3(512), 0, st, p, chunks, items);
    else hipLaunchKernelGGL((k_attn_short<DT, 40, 8>), dim3(grid), dim3(512), 0, st, p, chunks, items);
  } else {
    if (qc == 16) hipLaunchKernelGGL((k_attn_short<DT, 80, 16>), dim3(grid), dim3(512), 0, st, p, chunks, items);
    else hipLaunchKernelGGL((k_attn_short<DT, 80, 8>), dim3(grid), dim3(512), 0, st, p, chunks, items);
  }
}

template <int DT>
int launch_attn(const AttnKParams& p, hipStream_t st) {
  const int d = p.head_dim;
  const AttnPlan plan = attn_plan(p);
  if (plan == AP_SHORT) {
    launch_attn_short<DT>(p, st);
    return CA_OK;
  }
  if (plan == AP_DMA80) {
    AttnKParams q = p;
    q.qblocks = ceil_div_i(p.nq, 128);
    hipLaunchKernelGGL((k_attn_dma<DT, 2, 5, 2, 4, 64, false, false, true>), dim3((unsigned)(q.qblocks * p.batches * p.heads)), dim3(256), 0, st, q);
    return CA_OK;
  }
  if (d <= 32) launch_attn_d<DT, 1, 2>(p, plan, st);
  else if (d <= 48) launch_attn_d<DT, 2, 3>(p, plan, st);
  else if (d <= 64) launch_attn_d<DT, 2, 4>(p, plan, st);
  else if (d <= 80) launch_attn_d<DT, 3, 5>(p, plan, st);
  else if (d <= 128) launch_attn_d<DT, 4, 8>(p, plan, st);
  else launch_attn_d<DT, 5, 10>(p, plan, st);
  return CA_OK;
}

}  // namespace

static int attn_params(const ca_attn_args* a, AttnKParams& p);

extern "C" int ca_attention_plan_name(const ca_attn_args* a, char* buf, int32_t len) {
  CA_REQUIRE(buf != nullptr && len > 0, "ca_attention_plan_name: no buffer");
  AttnKParams p{};
  const int rc = attn_params(a, p);
  if (rc != CA_OK) return rc;
  snprintf(buf, (size_t)len, "%s", attn_plan_label(attn_plan(p)));
  return CA_OK;
}

extern "C" int ca_attention(const ca_attn_args* a, void* stream) {
  AttnKParams p{};
  const int rc = attn_params(a, p);
  if (rc != CA_OK) return rc;
  if (a->dtype == CA_BF16) launch_attn<CA_BF16>(p, (hipStream_t)stream);
  else launch_attn<CA_F16>(p, (hipStream_t)stream);
  CA_CHECK_LAUNCH("ca_attention");
  return CA_OK;
}

static int attn_params(const ca_attn_args* a, AttnKParams& p) {
  CA_REQUIRE(a != nullptr, "ca_attention: null args");
  CA_REQUIRE(a->q && a->k && a->v && a->o, "ca_attention: null operand");
  CA_REQUIRE(a->head_dim > 0 && a->head_dim % 8 == 0 && a->head_dim <= 160, "ca_attention: head_dim=%d must be a multiple of 8 and <= 160", a->head_dim);
  CA_REQUIRE(a->batches > 0 && a->heads > 0 && a->nq > 0 && a->nk > 0, "ca_attention: bad sizes");
  CA_REQUIRE(a->inner_count > 0 && a->kv_inner_count > 0 && a->kv_div > 0, "ca_attention: bad batch decomposition");
  CA_REQUIRE(a->q_row % 8 == 0 && a->k_row % 8 == 0 && a->o_row % 4 == 0, "ca_attention: row strides misaligned");
  CA_REQUIRE(a->q_outer % 8 == 0 && a->q_inner % 8 == 0 && a->k_outer % 8 == 0 && a->k_inner % 8 == 0 && a->o_outer % 4 == 0 && a->o_inner % 4 == 0,
             "ca_attention: batch strides misaligned");
  CA_REQUIRE(a->dtype == CA_BF16 || a->dtype == CA_F16, "ca_attention: dtype %d", a->dtype);
  CA_REQUIRE((int64_t)a->batches * a->heads * ceil_div_i(a->nq, 128) < (1ll << 31), "ca_attention: grid too large");
  p.q = (const u16*)a->q;
  p.k = (const u16*)a->k;
  p.v = (const u16*)a->v;
  p.o = (u16*)a->o;
  p.q_outer = a->q_outer; p.q_inner = a->q_inner; p.q_row = a->q_row;
  p.o_outer = a->o_outer; p.o_inner = a->o_inner; p.o_row = a->o_row;
  p.k_outer = a->k_outer; p.k_inner = a->k_inner; p.k_row = a->k_row;
  p.inner_count = a->inner_count;
  p.kv_inner_count = a->kv_inner_count;
  p.kv_div = a->kv_div;
  p.kv_mod = a->kv_mod > 0 ? a->kv_mod : a->batches;
  p.batches = a->batches;
  p.heads = a->heads;
  p.head_dim = a->head_dim;
  p.nq = a->nq;
  p.nk = a->nk;
  p.scale_log2 = a->scale * 1.4426950408889634f;
  p.out_scale = a->out_scale;
  p.accumulate = a->accumulate;
  p.causal = a->causal ? 1 : 0;
  CA_REQUIRE(!a->causal || a->nq == a->nk, "ca_attention: causal needs nq == nk");
  p.key_mask = a->key_mask;
  p.key_mask_stride = a->key_mask_stride;
  CA_REQUIRE(!a->key_mask || a->key_mask_stride >= a->nk, "ca_attention: key_mask_stride=%lld < nk", (long long)a->key_mask_stride);
  {
    const int d = a->head_dim;
    const int dvp = d <= 32 ? 32 : d <= 48 ? 48 : d <= 64 ? 64 : d <= 80 ? 80 : d <= 128 ? 128 : 160;
    p.sum_row = (d < dvp && d % 4 == 0) ? 1 : 0;
  }
  return CA_OK;
}

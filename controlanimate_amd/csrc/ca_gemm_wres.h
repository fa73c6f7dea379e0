// Weight-resident streaming GEMM for the K = 320 layers of the 64x64-latent level (q|k|v, out-projections, GEGLU
// projection, proj_in / proj_out: M = 131072 rows per step and up).  These launches sat off both roofs with the tiled
// kernels (131072x320x320: 13 % of the MFMA peak at < 40 % of the HBM rate): a block tile lives for five K tiles, each
// a dependent LDS-DMA round trip, and more than half of the bytes a tile pulls from L2 are the SAME 100 KB of weights.
//
//   * a persistent block keeps its 160-column panel of W (160 x 320, 100 KB, rows padded to 656 B: conflict-free
//     fragment reads) in LDS for its whole life -- loaded once, one barrier, none afterwards;
//   * each of the 8 waves streams its OWN 32-row slabs of A through a wave-private ring of three 2 KB slots
//     (32 rows x 32 K per slot, LDS-DMA, counted vmcnt): no barrier anywhere in the stream, a wave's epilogue
//     (stores, residual reads) overlaps the other waves' MFMAs the way independent blocks would;
//   * per slab a wave computes 32 x 160 outputs (2 x 10 MFMA tiles, 80 accumulator registers) and writes them from
//     the accumulator layout (a lane holds 4 consecutive columns of one row: 8-byte stores, 32 B per row and
//     instruction, ten instructions complete 320 contiguous bytes of each of 16 rows).
//
// Work split: block b runs on XCD b % 8; the blocks of an XCD that share a slab sequence ("lane") differ only in their
// column panel, so a slab is read from HBM once and from that XCD's L2 by the other panels.
//     slot = b / 8, panel = slot % panels, lane = b % 8 + 8 * (slot / panels); slab chunks (256 rows) lane, lane + lanes, ...
//
// Epilogue arithmetic and rounding order are those of gemm_epilogue (ca_gemm_core.h): round((acc [LN fold] + bias +
// rowbias) * alpha), then + residual, * post, activation, GEGLU, round.
//
// Round 3: the wave no longer waits on the VMEM counter for its ring chunks.  A counted `s_waitcnt vmcnt(4)` -- "all but the
// two youngest chunks have landed" -- also waited for the 20 stores of the previous slab's epilogue (shared counter, stores
// retire out of order with respect to loads): ~2-4 us per slab against ~1.5 us of MFMA work, two thirds of the gap between
// this kernel and its K loop alone.  Each chunk's two DMA pieces are now followed by a one-lane
// 4-byte LDS-DMA that fetches the chunk's sequence number into a flag word of the wave (ca_gemm_seq.h); the wave reads the
// NEXT chunk's flag in the shadow of the current chunk's MFMAs and only spins when it is not there yet.  Stores are
// fire-and-forget.
#include "ca_gemm_seq.h"

// Round 3, 16-byte stores from the MFMA layout: a lane holds 4 consecutive output columns per MFMA tile (8-byte pieces,
// 32 B per row and instruction; with GEGLU 2 outputs = 4-byte pieces).  Which weight row sits in which LDS row of the
// panel is free, so the rows of a PAIR of MFMA tiles are interleaved in groups of four and a lane holds 8 consecutive
// columns: 131072x320x320 + residual 65.6 -> 52.3 us, x960 183 -> 159, x1280 224 -> 186 (same box, tools/wres_check.py).
// GEGLU keeps the natural order and its 4-byte pieces: interleaving FOUR tiles for 16-byte stores of 8 consecutive outputs
// was measured too and is slower (397 vs 367 us at 131072x2560x320: the epilogue is VALU-bound there, and the wider
// register tuples cost more than the stores save).  column (inside the 160-column panel) of fragment row i of MFMA tile j:
__device__ __forceinline__ int ca_wres_col(int j, int i, bool geglu) {
  if (geglu) return 16 * j + i;  // (GEGLU keeps the natural order: see below)
  return 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3);
}

template <int DT>
__global__ __launch_bounds__(512, 1) void k_gemm_wres(GemmKParams p, int panels, int lanes, int chunks, unsigned rb_bytes, unsigned c_bytes, unsigned res_bytes) {
  constexpr int K = 320, PN = 160, TN = 10, KQ = 10;  // panel width, n tiles per wave, ring chunks (32 K) per slab
  constexpr int WLD_B = 656;                          // bytes per padded W row (41 16-byte chunks)
  constexpr int W_BYTES = PN * WLD_B;                 // 104960
  constexpr int SLOT = 2048, RING = 3 * SLOT;
  constexpr int OFF_PAR = W_BYTES + 8 * RING;  // bias[160] | colsum[160] (fp32), loaded once with the weights
  constexpr int OFF_RB = OFF_PAR + 1280;       // per wave: the 160 rowbias values of the current slab's row group (1 KB)
  constexpr int OFF_FLAG = OFF_RB + 8 * 1024;  // per wave: one flag word per ring slot (written by a one-lane LDS-DMA)
  __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_FLAG + 8 * 16];
  static_assert(OFF_FLAG + 8 * 16 <= 160 * 1024, "LDS");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;

  const int slot_b = blockIdx.x >> 3;
  const int panel = slot_b % panels;
  const int my_lane = (blockIdx.x & 7) + 8 * (slot_b / panels);
  if (my_lane >= lanes) return;
  const int n0 = panel * PN;

  // ---- the weight panel: once
  const bool geglu = p.geglu != 0;
  for (int q = tid; q < PN * (K / 8); q += 512) {
    const int row = q / (K / 8), c = q - row * (K / 8);  // LDS row = fragment row (row & 15) of MFMA tile (row >> 4)
    *reinterpret_cast<u32x4*>(smem + row * WLD_B + c * 16) = ld16(p.w + (int64_t)(n0 + ca_wres_col(row >> 4, row & 15, geglu)) * K + c * 8);
  }
  if (tid < 2 * PN) {
    const float* src = tid < PN ? p.bias : p.ln_colsum;
    const int n = tid < PN ? tid : tid - PN;
    reinterpret_cast<float*>(smem + OFF_PAR)[tid] = src ? src[n0 + n] : 0.f;
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rs_rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.rowbias ? (const void*)p.rowbias : (const void*)p.w), 0, rb_bytes, 0x00020000);
  unsigned char* rb_patch = smem + OFF_RB + wid * 1024;
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? (const void*)p.res : (const void*)p.c), 0, p.res ? res_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, p.a2 ? p.a2_bytes : p.a_bytes, 0x00020000);
  unsigned char* ring = smem + W_BYTES + wid * RING;
  const __amdgpu_buffer_rsrc_t rs_seq = __builtin_amdgcn_make_buffer_rsrc((void*)ca_seq_table.v, 0, 4096u, 0x00020000);
  unsigned char* const my_flags = smem + OFF_FLAG + wid * 16;
  const unsigned flag_addr0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)my_flags;
  if (lane < 4) {  // (wave-private words: no barrier, but the write must have completed before a DMA can land on it)
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(flag_addr0 + (unsigned)lane * 4u), "v"(0xFFFFFFFFu) : "memory");
  }

  // ---- stream head (what the next DMA pair fetches): slab chunk d_c, ring chunk d_k of it, slot d_s
  int d_c = my_lane, d_k = 0, d_s = 0;
  int d_n = 0;  // chunks issued by this wave so far: chunk n carries sequence number n & 1023 in flag word n % 3
  unsigned d_v1[2], d_v2[2];
  auto head_slab = [&]() {
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));  // (keeps the address arithmetic where it is used: see ca_gemm_pp3.h)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int m = d_c * 256 + wid * 32 + h * 16 + (lane_o >> 2);
      const bool ok = m < p.m;
      d_v1[h] = ok ? (unsigned)m * (unsigned)p.lda * 2u + (unsigned)(lane_o & 3) * 16u : DMA_OOB;
      d_v2[h] = ok ? (unsigned)m * (unsigned)p.lda2 * 2u + (unsigned)(lane_o & 3) * 16u : DMA_OOB;
    }
  };
  auto issue = [&]() {  // one ring chunk = 2 DMA instructions (16 rows x 64 B each)
    const int k0 = d_k * 32;
    unsigned char* dst = ring + d_s * SLOT;
    if (k0 >= p.c1) {
      const unsigned so = (unsigned)(k0 - p.c1) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (__attribute__((address_space(3))) void*)dst, 16, d_v2[0], so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (__attribute__((address_space(3))) void*)(dst + 1024), 16, d_v2[1], so, 0, 0);
    } else {
      const unsigned so = (unsigned)k0 * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)dst, 16, d_v1[0], so, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(dst + 1024), 16, d_v1[1], so, 0, 0);
    }
    {  // the chunk's flag: ONE lane fetches its sequence number behind the two pieces (loads return in order)
      int lane_f = lane;
      asm volatile("" : "+v"(lane_f));
      if (lane_f == 0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_seq, (__attribute__((address_space(3))) void*)(my_flags + d_s * 4), 4, 0u, (unsigned)(d_n & 1023) * 4u, 0, 0);
    }
    ++d_n;
    d_s = d_s == 2 ? 0 : d_s + 1;
    if (++d_k == KQ) {
      d_k = 0;
      d_c += lanes;  // (past the last chunk every row is out of range: the DMA writes zeros nobody reads)
      head_slab();
    }
  };

  // chunk n has landed when flag word n % 3 shows n & 1023.  `have` = a value of that word read earlier (0xFFFFFFFE: none);
  // the slow path re-reads with a short sleep and gives up after ~2^22 polls (a hung wave would take the device down).
  auto chunk_wait = [&](int n, unsigned have) __attribute__((always_inline)) {
    const unsigned want = (unsigned)(n & 1023);
    if ((unsigned)__builtin_amdgcn_readfirstlane(have) == want) return;
    const unsigned addr = flag_addr0 + (unsigned)(n % 3) * 4u;
    for (unsigned spins = 0; spins < (1u << 22); ++spins) {
      unsigned v;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
      if ((unsigned)__builtin_amdgcn_readfirstlane(v) == want) return;
      __builtin_amdgcn_s_sleep(1);
    }
    // gave up polling (a pre-empted or very slow DMA): loads return in order, so draining the wave's VMEM counter is the
    // correct -- merely slower -- way to know the chunk has landed; never continue on stale LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  int r_n = 0;  // chunk computed next

  head_slab();
  issue();
  issue();
  chunk_wait(0, 0xFFFFFFFEu);

  const int fa_lane = l15 * 64 + g * 16;                    // A fragment: row l15 (+16), 16 bytes at k = g*8
  const unsigned char* wb = smem + l15 * WLD_B + g * 16;    // W fragment of n tile j, chunk kq: + j*16*WLD_B + kq*64
  int r_s = 0;                                              // ring slot of the chunk computed next

  for (int c = my_lane; c < chunks; c += lanes) {
    const int m0 = c * 256 + wid * 32;
    f32x4 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // LayerNorm statistics of this lane's two rows, and the row group's bias (one group per 32-row slab: the
    // launcher requires rows_per_group % 32 == 0) by LDS-DMA into the wave's patch -- both long landed at the epilogue
    float2 st[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
    float ln_s[2] = {0.f, 0.f}, ln_ss[2] = {0.f, 0.f};
    if (p.ln_stats) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 16 + l15;
        if (m < p.m) st[i] = *reinterpret_cast<const float2*>(p.ln_stats + (int64_t)m * 2);
      }
    }
    if (p.rowbias) {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const unsigned off = (lane_o < PN / 4 && m0 < p.m) ? ((unsigned)(m0 / p.rows_per_group) * (unsigned)p.ld_rowbias + (unsigned)n0) * 4u + (unsigned)lane_o * 16u : DMA_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_rb, (__attribute__((address_space(3))) void*)rb_patch, 16, off, 0, 0, 0);
    }
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
      issue();  // two chunks ahead, into the slot whose fragments were consumed in the previous iteration
      // (this chunk was confirmed landed at the end of the previous iteration -- or before the loop -- by its flag word)
      const unsigned char* as = ring + r_s * SLOT + fa_lane;
      // all twelve fragments of the chunk are requested at once; LDS returns them in order, so MFMA pair j starts as
      // soon as fragment j is in (counted lgkmcnt) while the rest stream in behind it
      __builtin_amdgcn_sched_barrier(0);
      const u32x4 fa0 = *reinterpret_cast<const u32x4*>(as);
      const u32x4 fa1 = *reinterpret_cast<const u32x4*>(as + 1024);
      u32x4 fb[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        __builtin_amdgcn_sched_barrier(0);  // (in this order: the waits below count on it)
        fb[j] = *reinterpret_cast<const u32x4*>(wb + j * 16 * WLD_B + kq * 64);
      }
      __builtin_amdgcn_sched_barrier(0);
#define CA_WRES_PAIR(J, CNT)                                        \
  asm volatile("s_waitcnt lgkmcnt(" #CNT ")" ::: "memory");         \
  __builtin_amdgcn_sched_barrier(0);                                \
  acc[0][J] = Elem<DT>::mfma(fb[J], fa0, acc[0][J]);                \
  acc[1][J] = Elem<DT>::mfma(fb[J], fa1, acc[1][J]);                \
  __builtin_amdgcn_sched_barrier(0);
      CA_WRES_PAIR(0, 9)
      if (p.ln_inline) {
        // LayerNorm statistics of the rows this wave streams anyway (ca_gemm_args.ln_eps): sum and sum of squares of
        // this lane's 8-element slice of rows l15 and l15 + 16, in the shadow of the MFMAs; reduced over the four
        // 8-element lane groups after the K loop.
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // (v_dot2_f32_f16 via __builtin_amdgcn_fdot2 gave 7e-2 relative error here on gfx950 -- not pursued; plain
          //  conversions + FMAs cost ~200 VALU cycles per chunk against 320 cycles of MFMA issue)
          const float a0 = Elem<DT>::to_f((u16)(fa0[q] & 0xffffu)), a1 = Elem<DT>::to_f((u16)(fa0[q] >> 16));
          const float b0 = Elem<DT>::to_f((u16)(fa1[q] & 0xffffu)), b1 = Elem<DT>::to_f((u16)(fa1[q] >> 16));
          ln_s[0] += a0 + a1;
          ln_ss[0] = fmaf(a0, a0, fmaf(a1, a1, ln_ss[0]));
          ln_s[1] += b0 + b1;
          ln_ss[1] = fmaf(b0, b0, fmaf(b1, b1, ln_ss[1]));
        }
      }
      CA_WRES_PAIR(1, 8) CA_WRES_PAIR(2, 7) CA_WRES_PAIR(3, 6) CA_WRES_PAIR(4, 5)
      // the NEXT chunk's flag word, read behind the fragment reads still in flight (fb[5..9]): one more LDS operation
      // outstanding from here on, hence the counts of the remaining pairs are one higher than their fragment index suggests
      unsigned flag_next;
      asm volatile("ds_read_b32 %0, %1" : "=v"(flag_next) : "v"(flag_addr0 + (unsigned)((r_n + 1) % 3) * 4u) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      CA_WRES_PAIR(5, 5) CA_WRES_PAIR(6, 4) CA_WRES_PAIR(7, 3) CA_WRES_PAIR(8, 2) CA_WRES_PAIR(9, 1)
#undef CA_WRES_PAIR
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(flag_next)::"memory");
      // (lgkmcnt(0) above: every read of the slot has returned before the next issue() re-fills it)
      r_s = r_s == 2 ? 0 : r_s + 1;
      ++r_n;
      chunk_wait(r_n, flag_next);
    }

    if (p.ln_inline) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float a = ln_s[i], b = ln_ss[i];
        a += __shfl_xor(a, 16);
        b += __shfl_xor(b, 16);
        a += __shfl_xor(a, 32);
        b += __shfl_xor(b, 32);
        const float mean = a * (1.f / K);
        st[i] = make_float2(mean, rsqrtf(fmaxf(b * (1.f / K) - mean * mean, 0.f) + p.ln_eps));  // (= k_ln_stats)
      }
    }
    // ---- epilogue of this wave's 32 x 160 patch, from the accumulators.  Every load is issued before the first
    // store (the VMEM counter is shared and in order: a load issued after a store could only be awaited together
    // with that store's completion): residual quads up front, LayerNorm statistics before the K loop, bias / column
    // sums / row bias from LDS.
    if (p.dbg == 1) {
      if (acc[0][0][0] == 12345.678f) *reinterpret_cast<float*>(p.c) = acc[1][TN - 1][1];
      continue;
    }
    // (the lane id is made opaque here: hipcc otherwise hoists every lane-dependent address of the three epilogue variants
    //  out of the slab loop and spills them -- 72 VGPRs at the first attempt)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int g = lane_e >> 4, l15 = lane_e & 15;
    u32x4 rr[2][TN / 2];  // per tile pair q: the lane's 8 consecutive columns 32q + 8g .. +7
    if (p.res) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < TN / 2; ++q) {
          const int m = m0 + i * 16 + l15;
          const unsigned off = m < p.m ? ((unsigned)m * (unsigned)p.ld_res + (unsigned)(n0 + g * 8)) * 2u : 0x80000000u;  // (out of range reads 0)
          rr[i][q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, off, q * 64, 0));
        }
    }
    if (p.alpha == 1.f && p.post == 1.f && p.act == CA_ACT_NONE && !p.geglu) {
      // the common cases (projections with bias, residual and / or a folded LayerNorm): ~12-20 VALU instructions per
      // fragment instead of ~45 -- the epilogue's VALU time is of the order of the slab's MFMA time, and it is the part
      // that does not scale away
      unsigned off[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 16 + l15;
        off[i] = m < p.m ? ((unsigned)m * (unsigned)p.ldc + (unsigned)(n0 + g * 8)) * 2u : 0x80000000u;
      }
#pragma unroll
      for (int q = 0; q < TN / 2; ++q) {
        f32x4 bb[2], cs[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int col = 32 * q + 8 * g + 4 * u;  // = ca_wres_col(2q + u, 4g .. 4g+3)
          bb[u] = *reinterpret_cast<const f32x4*>(smem + OFF_PAR + col * 4);
          cs[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (p.ln_colsum) cs[u] = *reinterpret_cast<const f32x4*>(smem + OFF_PAR + (PN + col) * 4);
          if (p.rowbias) {  // (the general path adds bias then row bias to the value; here their sum first: last-bit differences only)
            const f32x4 rb = *reinterpret_cast<const f32x4*>(rb_patch + col * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) bb[u][r] += rb[r];
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          u32x4 w;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            f32x4 v = acc[i][2 * q + u];
            if (p.ln_colsum) {  // rstd * (x W'^T - mean * colsum(W')): same operation order as the general path
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = st[i].y * (v[r] - st[i].x * cs[u][r]);
            }
            w[2 * u] = pack2<DT>(v[0] + bb[u][0], v[1] + bb[u][1]);
            w[2 * u + 1] = pack2<DT>(v[2] + bb[u][2], v[3] + bb[u][3]);
          }
          if (p.res) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (DT == CA_F16) {  // fp16 + fp16 is exact in fp32, so the packed add rounds exactly like the fp32 path
                // (inline asm: hipcc 7.2 miscompiled the ext_vector _Float16 addition here -- the second add reused the
                //  first one's result and operand)
                unsigned s_;
                asm("v_pk_add_f16 %0, %1, %2" : "=v"(s_) : "v"(w[k]), "v"(rr[i][q][k]));
                w[k] = s_;
              } else {
                w[k] = pack2<DT>(Elem<DT>::to_f((u16)(w[k] & 0xffffu)) + Elem<DT>::to_f((u16)(rr[i][q][k] & 0xffffu)),
                                 Elem<DT>::to_f((u16)(w[k] >> 16)) + Elem<DT>::to_f((u16)(rr[i][q][k] >> 16)));
              }
            }
          }
          // (the column offset goes into the instruction's immediate field, NOT the SGPR soffset operand: on gfx950 a
          //  buffer_store_dwordx4 with a register soffset followed directly by a VALU write of its data registers stored
          //  the NEW values for lanes 12-15 of each 16 -- hipcc 7.2 only keeps the one wait state for immediate offsets;
          //  DESIGN.md section 3, "16-byte store hazard")
          __builtin_amdgcn_raw_buffer_store_b128(w, rs_c, off[i] + (unsigned)(q * 64), 0, 0);
        }
      }
      continue;
    }
    if (p.geglu && !p.rowbias && !p.res && p.alpha == 1.f && p.post == 1.f && p.act == CA_ACT_NONE) {
      // the feed-forward projection (folded LayerNorm, bias, GEGLU): the same arithmetic as the general path below
      // without its per-fragment branches and row-bias / residual / scale steps
      unsigned off[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 16 + l15;
        off[i] = m < p.m ? ((unsigned)m * (unsigned)p.ldc + (unsigned)((n0 >> 1) + g * 2)) * 2u : DMA_OOB;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4 bi = *reinterpret_cast<const f32x4*>(smem + OFF_PAR + (j * 16 + g * 4) * 4);
        f32x4 cs = {0.f, 0.f, 0.f, 0.f};
        if (p.ln_colsum) cs = *reinterpret_cast<const f32x4*>(smem + OFF_PAR + (PN + j * 16 + g * 4) * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          f32x4 v = acc[i][j];
          if (p.ln_colsum) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = st[i].y * (v[r] - st[i].x * cs[r]);
          }
          float h[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) h[r] = Elem<DT>::to_f(Elem<DT>::from_f(v[r] + bi[r]));  // (the Linear's output is rounded first)
          const f32x2 gg = gelu_erf_f2((f32x2){h[1], h[3]});
          __builtin_amdgcn_raw_buffer_store_b32(pack2<DT>(h[0] * gg[0], h[2] * gg[1]), rs_c, off[i], j * 16, 0);
        }
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int nl = ca_wres_col(j, 4 * g, geglu);  // the lane's 4 consecutive columns of this tile start here (see ca_wres_col)
      const f32x4 bi = *reinterpret_cast<const f32x4*>(smem + OFF_PAR + nl * 4);
      f32x4 cs = {0.f, 0.f, 0.f, 0.f}, rb = {0.f, 0.f, 0.f, 0.f};
      if (p.ln_colsum) cs = *reinterpret_cast<const f32x4*>(smem + OFF_PAR + (PN + nl) * 4);
      if (p.rowbias) rb = *reinterpret_cast<const f32x4*>(rb_patch + nl * 4);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int m = m0 + i * 16 + l15;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
        if (p.ln_colsum) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = st[i].y * (v[r] - st[i].x * cs[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = Elem<DT>::to_f(Elem<DT>::from_f((v[r] + bi[r] + rb[r]) * p.alpha));
        if (p.res) {  // (never with GEGLU: rr holds the 8 columns 32q + 8g .. of tile pair q = j / 2, this tile's 4 are half j & 1)
          const unsigned r0 = rr[i][j >> 1][2 * (j & 1)], r1 = rr[i][j >> 1][2 * (j & 1) + 1];
          v[0] += Elem<DT>::to_f((u16)(r0 & 0xffffu));
          v[1] += Elem<DT>::to_f((u16)(r0 >> 16));
          v[2] += Elem<DT>::to_f((u16)(r1 & 0xffffu));
          v[3] += Elem<DT>::to_f((u16)(r1 >> 16));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= p.post;
        if (p.act != CA_ACT_NONE) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = act_f(v[r], p.act);
        }
        // (rows past M: the offset is out of the descriptor's range, the hardware drops the store)
        if (p.geglu) {
          const f32x2 gg = gelu_erf_f2((f32x2){v[1], v[3]});
          const unsigned w = pack2<DT>(v[0] * gg[0], v[2] * gg[1]);
          const unsigned off = m < p.m ? ((unsigned)m * (unsigned)p.ldc + (unsigned)((n0 >> 1) + (nl >> 1))) * 2u : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b32(w, rs_c, off, 0, 0);
        } else {
          u32x2 w;
          w[0] = pack2<DT>(v[0], v[1]);
          w[1] = pack2<DT>(v[2], v[3]);
          const unsigned off = m < p.m ? ((unsigned)m * (unsigned)p.ldc + (unsigned)(n0 + nl)) * 2u : 0x80000000u;
          __builtin_amdgcn_raw_buffer_store_b64(w, rs_c, off, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the row-bias patch is re-filled at the next slab's start)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two chunks fetched past the end
}

// Activation-resident GEMM for the K = 320 projections of the 64x64-latent level (round 4): q|k|v (N = 960), the GEGLU
// feed-forward projection (N = 2560), proj_in / proj_out / to_out / to_q (N = 320) at M = 131072 rows and up.
//
// Why another kernel: the weight-resident kernel (ca_gemm_wres.h) keeps a 160-column panel of W in LDS and gives every wave a
// 32 x 160 output patch -- (32 + 160) fragment rows = 12 ds_read_b128 per 20 MFMAs, every one of them re-read from LDS for every
// 32-row slab, and a 20-MFMA chunk per LDS round trip.  Measured in the step: 0.55 PFLOP/s on the N >= 960 members
// (131072x2560x320 GEGLU 380 us, x960 146 us), the matrix pipes 21 % busy.  Here the roles are swapped:
//
//   * a block owns a 128-row tile of A: 128 x 320 = 80 KB, the WHOLE K extent, in LDS (LDS-DMA, XOR-swizzled rows, one barrier
//     per row tile); two blocks of four waves share a CU (2 x 80 KB = all of the 160 KB), so one block's MFMAs cover the other's
//     tile load and epilogues -- no flags, no ping-pong barriers, no counted waits inside the K loop;
//   * a wave computes 128 x 64 outputs per item (8 x 4 MFMA tiles, 128 accumulators): per 32-deep chunk 8 A fragments from LDS
//     (8 KB per 32 MFMAs: 0.4 of the weight-resident kernel's LDS bytes per MFMA) and 4 W fragments that bypass LDS
//     altogether: W fragments are private to a wave here (the four waves hold different column panels), so they come straight
//     from L2 into registers out of a FRAGMENT-ORDERED copy of W (ca_pack_w_frag: 1 KB contiguous per wave instruction instead
//     of the 16 rows x 64 B a row-major W would give) -- 128 B of L2 traffic per MFMA, 20-30 B/clk per CU;
//   * the column loop is j-major: after the eight MFMAs of column tile j its W fragment is dead and the next chunk's is
//     fetched INTO THE SAME REGISTERS (needed 32 MFMAs later); the next chunk's A fragments replace the current ones one by one
//     behind the MFMAs of the last column tile.  Nothing is double-buffered: 128 + 32 + 16 registers.  (A 128 x 80 patch --
//     160 accumulators, N / 80 panels in whole rounds of four waves -- was built first: hipcc could not hold the epilogue's
//     parameters beside 160 accumulators and spilled 40..90 registers, among them the K loop's address registers.)
//   * epilogue straight from the accumulators with 16-byte stores (the weight-row interleave of ca_gemm_ps.h), parameters
//     (bias, row bias, column sums, LayerNorm statistics) loaded from L2 at its start -- the block's LDS is all tile.
//
// LayerNorm statistics in the kernel (ca_gemm_args.ln_colsum without ln_stats): two threads per row sum x and x^2 of the tile
// once it has landed and leave (mean, rstd) in a caller-owned scratch of M x 8 bytes (ca_gemm_workspace_bytes), which the
// epilogues read back (same block, behind a barrier).
//
// Requirements (ar_eligible in ca_gemm.hip): K = 320 from one source, N % 64 == 0, fp16 / bf16 output, alpha = post = 1, no
// activation, no row sums, 32-bit byte offsets, row-bias groups of a multiple of 128 rows, GEGLU only without residual.

// column (inside a wave's 64-column panel) of fragment row i of MFMA tile j: rows of a PAIR of tiles interleaved in fours so
// that a lane holds 8 consecutive output columns (16-byte stores); GEGLU: four tiles interleaved -> 8 consecutive OUTPUTS
__device__ __forceinline__ int ca_ar_col(int j, int i, bool geglu) {
  if (geglu) return 16 * (i >> 2) + 4 * j + (i & 3);
  return 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3);
}

// fragment-ordered W: element e of lane L's 16 bytes of MFMA tile j of 32-deep chunk kq of 64-column panel pn is
//   W[pn * 64 + ca_ar_col(j, L & 15, geglu)][kq * 32 + (L >> 4) * 8 + e]        at (((pn * 10 + kq) * 4 + j) * 64 + L) * 8 + e
__global__ __launch_bounds__(256) void k_pack_w_frag(const u16* __restrict__ w, u16* __restrict__ dst, int n, int geglu) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one 16-byte piece each
  if (idx >= (int64_t)n * 40) return;
  const int L = (int)(idx & 63);
  int64_t t = idx >> 6;
  const int j = (int)(t & 3);
  t >>= 2;
  const int kq = (int)(t % 10);
  const int pn = (int)(t / 10);
  const int row = pn * 64 + ca_ar_col(j, L & 15, geglu != 0);
  st16(dst + idx * 8, ld16(w + (int64_t)row * 320 + kq * 32 + (L >> 4) * 8));
}

// EPI = 0: bias, row bias, residual.  EPI = 1: folded LayerNorm ((mean, rstd) per row), bias, row bias.  EPI = 2: folded
// LayerNorm, bias, GEGLU.
template <int DT, int EPI>
__global__ __launch_bounds__(256, 2) void k_gemm_ar(GemmKParams p, const u16* __restrict__ wf, int tiles_m, unsigned c_bytes, unsigned res_bytes,
                                                     float* __restrict__ stats_ws) {
  constexpr int K = 320, KQ = 10, TM = 8, TN = 4, BM = 128, PN = 64;
  constexpr int ROWB = K * 2;  // 640 bytes per row, 40 chunks of 16 bytes; chunk c of row r sits at chunk c ^ ((r >> 1) & 7)
  __shared__ __attribute__((aligned(16))) unsigned char smem[BM * ROWB];
  static_assert(2 * BM * ROWB <= 160 * 1024, "two blocks per CU");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  constexpr bool geglu = EPI == 2;
  constexpr unsigned OOB_V = 0x80000000u;

  const int panels = p.n / PN;

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wf, 0, (unsigned)p.n * (unsigned)(K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? (const void*)p.res : (const void*)p.c), 0, p.res ? res_bytes : 0u, 0x00020000);
  // epilogue operands: an absent one has a descriptor of size 0 and reads zeros -- exact identities, no branches (ca_gemm_ps.h)
  const __amdgpu_buffer_rsrc_t rs_bi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)wf), 0, p.bias ? (unsigned)p.n * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ln_colsum ? (const void*)p.ln_colsum : (const void*)wf), 0, p.ln_colsum ? (unsigned)p.n * 4u : 0u, 0x00020000);
  const unsigned rb_groups = p.rowbias ? (unsigned)((p.m + p.rows_per_group - 1) / p.rows_per_group) : 0u;
  const __amdgpu_buffer_rsrc_t rs_rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.rowbias ? (const void*)p.rowbias : (const void*)wf), 0,
                                                                         p.rowbias ? (unsigned)(((int64_t)(rb_groups - 1) * p.ld_rowbias + p.n) * 4) : 0u, 0x00020000);
  const float* const st_src = p.ln_stats ? p.ln_stats : stats_ws;  // (mean, rstd) per row: the caller's, or this kernel's own
  const __amdgpu_buffer_rsrc_t rs_st = __builtin_amdgcn_make_buffer_rsrc((void*)(st_src ? (const void*)st_src : (const void*)wf), 0, st_src ? (unsigned)p.m * 8u : 0u, 0x00020000);
  const float rstd_id = st_src ? 0.f : 1.f;  // no statistics: (mean, rstd) reads (0, 0) -> (0, 1), 1 * (x - 0 * cs) = x

  // A fragment of row tile i, chunk kq: row i*16 + l15, logical chunk 4 kq + g.  With f = (row >> 1) & 7 = (l15 >> 1) & 7 the
  // physical chunk is 8 (kq >> 1) + 4 ((kq & 1) ^ (f >> 2)) + (g ^ (f & 3)): even kq at fa_lane + (kq >> 1) * 128, odd kq at
  // (fa_lane ^ 64) + (kq >> 1) * 128 (bit 6 of fa_lane is f >> 2 alone: 640 = 5 * 128).  Conflict-free for ds_read_b128: the 16
  // lanes of a service group hold 16 different l15, 8 l15 + chunk covers the 16 slots of the 256-byte bank row once.
  const int f_sw = (l15 >> 1) & 7;
  const int fa_lane = l15 * ROWB + (((f_sw >> 2) << 2) + (g ^ (f_sw & 3))) * 16;
  // four base addresses (row tiles 0..3 / 4..7 x even / odd chunk) so that every fragment address is base + a 16-bit immediate:
  // left to itself hipcc keeps one precomputed address per (row tile, parity) beyond the immediate's reach and SPILLS them --
  // a scratch reload inside the K loop is a VMEM load that waits behind the whole W stream
  int fa_b[2][2] = {{fa_lane, fa_lane ^ 64}, {fa_lane + 4 * 16 * ROWB, (fa_lane ^ 64) + 4 * 16 * ROWB}};
  asm volatile("" : "+v"(fa_b[0][0]), "+v"(fa_b[0][1]), "+v"(fa_b[1][0]), "+v"(fa_b[1][1]));

  for (int tile_i = blockIdx.x; tile_i < tiles_m; tile_i += gridDim.x) {
#ifdef CA_EXPERIMENTS
    const int tile = p.dbg == 7 ? tiles_m - 1 - tile_i : tile_i;  // (experiment: rows last to first -- what the memory-side cache keeps of the output)
#else
    const int tile = tile_i;
#endif
    const int m0 = tile * BM;
    __syncthreads();  // every wave has finished its reads of the previous tile
    {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));  // (opaque: keeps the address arithmetic inside the tile loop)
#pragma unroll
      for (int q = 0; q < 20; ++q) {
        const unsigned idx = (unsigned)((wid * 20 + q) * 64 + lane_o);  // linear 16-byte piece of the tile image
        const unsigned r = __umulhi(idx >> 3, 0xCCCCCCCDu) >> 2;        // idx / 40
        const unsigned cp = idx - r * 40u;                              // physical chunk
        const unsigned c = cp ^ ((r >> 1) & 7u);
        const unsigned off = (m0 + (int)r) < p.m ? (unsigned)(m0 + (int)r) * (unsigned)p.lda * 2u + c * 16u : OOB_V;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(smem + (wid * 20 + q) * 1024), 16, off, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.ln_inline) {
      // LayerNorm statistics of the tile's rows: two threads per row, 20 pieces each (any 20 + 20 pieces of a row: sums)
      const int r = tid >> 1, h = tid & 1;
      const unsigned char* src = smem + r * ROWB + h * 320;
      float s = 0.f, ss = 0.f;
#pragma unroll 4
      for (int q = 0; q < 20; ++q) {
        const u32x4 v = ld16(src + ((q + r) % 20) * 16);  // (rotated by the row: spreads the rows of a wave over the banks)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = Elem<DT>::to_f((u16)(v[e] & 0xffffu)), a1 = Elem<DT>::to_f((u16)(v[e] >> 16));
          s += a0 + a1;
          ss = fmaf(a0, a0, fmaf(a1, a1, ss));
        }
      }
      s += __shfl_xor(s, 1);
      ss += __shfl_xor(ss, 1);
      const float mean = s * (1.f / K);
      const float rstd = rsqrtf(fmaxf(ss * (1.f / K) - mean * mean, 0.f) + p.ln_eps);  // (= k_ln_stats)
      if (h == 0 && m0 + r < p.m) *reinterpret_cast<float2*>(stats_ws + (int64_t)(m0 + r) * 2) = make_float2(mean, rstd);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }

    int lane_k = lane;
    asm volatile("" : "+v"(lane_k));
    const unsigned wv = (unsigned)lane_k * 16u;
    // item q of a tile = panel (q + tile) mod panels (the blocks of a round do not all pull the same W panel at the same time);
    // wave w takes items w, w + 4, ...
    auto panel_of = [&](int q) __attribute__((always_inline)) -> int { return (q + tile) % panels; };
    // W fragments two chunks ahead (fb[chunk & 1]): one chunk (24 MFMAs, ~500 cycles) does not cover an L2 round trip when every
    // CU streams W (measured: 131072x2560x320 GEGLU 292 us with one chunk of lead)
    u32x4 fa[TM], fb[2][TN];
    {
      const unsigned wb0 = (unsigned)panel_of(wid) * (unsigned)(KQ * TN * 1024);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, wb0 + (unsigned)(c * TN + j) * 1024u, 0));
    }
    for (int q = wid; q < panels; q += 4) {
      const int pn = panel_of(q);
      const int n0 = pn * PN;
      const unsigned wbase = (unsigned)pn * (unsigned)(KQ * TN * 1024);

      f32x4 acc[TM][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = ld16(smem + fa_b[i >> 2][0] + (i & 3) * 16 * ROWB);
      __builtin_amdgcn_sched_barrier(0);

#pragma unroll
      for (int kq = 0; kq < KQ; ++kq) {
        const int nk = kq + 1;
        const int fa_off = (nk >> 1) * 128;  // the next chunk: base of its parity + 128 bytes per chunk pair
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            if (kq == 0) acc[i][j] = Elem<DT>::mfma(fb[0][j], fa[i], (f32x4){0.f, 0.f, 0.f, 0.f});
            else acc[i][j] = Elem<DT>::mfma(fb[kq & 1][j], fa[i], acc[i][j]);
            if (j == TN - 1 && nk < KQ) {
              // the next chunk's A fragment i replaces this one right behind its last MFMA
              __builtin_amdgcn_sched_barrier(0);
              fa[i] = ld16(smem + fa_b[i >> 2][nk & 1] + fa_off + (i & 3) * 16 * ROWB);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          // ... and W fragment j of the chunk after the next behind the eight MFMAs of column tile j (needed 56 MFMAs later)
          if (kq + 2 < KQ) fb[kq & 1][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, wbase + (unsigned)(((kq + 2) * TN + j) * 1024), 0));
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      // ---------------------------------------------------------------- epilogue of the 128 x 64 patch
      // Loads and stores share one in-order counter: a load issued behind a store can only be awaited together with it
      // (~1-4 us when the chip writes at its HBM rate).  So: (0) the NEXT item's first W fragments and this item's parameters are
      // requested, (1) every value is computed and packed (the accumulators die row by row, the packed rows take half their
      // place; the residual is read one row ahead), (2) all stores.  No load of this item or of the next chunk 0 follows a store.
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));  // (opaque: hipcc otherwise hoists the lane-dependent addresses out of the item loop)
      const int l15 = lane_e & 15, g = lane_e >> 4;
      if (q + 4 < panels) {
        const unsigned wb1 = (unsigned)panel_of(q + 4) * (unsigned)(KQ * TN * 1024);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, wb1 + (unsigned)(c * TN + j) * 1024u, 0));
      }
      f32x4 bi[TN], cs[TN];
      const unsigned rb_off = (unsigned)(m0 / p.rows_per_group) * (unsigned)p.ld_rowbias * 4u;  // (rows_per_group % 128 == 0: one group per tile)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned c4 = (unsigned)(n0 + (geglu ? 16 * g + 4 * j : 32 * (j >> 1) + 8 * g + 4 * (j & 1))) * 4u;  // = ca_ar_col(j, 4 g, geglu)
        bi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bi, c4, 0, 0));
        if (EPI != 2) {  // bias + row bias first, as the weight-resident kernel's lean path
          const f32x4 rb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_rb, rb_off + c4, 0, 0));
          bi[j] += rb;
        }
        if (EPI != 0) cs[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_cs, c4, 0, 0));
      }
      // (statistics possibly this block's own stores of a moment ago: written through to L2 and awaited before the barrier; the
      //  vector L1 cannot hold an older copy -- a tile's 1 KB of statistics is touched by this block only, after the write)
      auto st_load = [&](int i) __attribute__((always_inline)) -> u32x2 {
        const int m = m0 + i * 16 + l15;
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_st, m < p.m ? (unsigned)m * 8u : OOB_V, 0, 0));
      };
      u32x2 st_nx[2] = {(u32x2){0u, 0u}, (u32x2){0u, 0u}};  // two rows ahead
      if (EPI != 0) {
        st_nx[0] = st_load(0);
        st_nx[1] = st_load(1);
      }
      constexpr int WR = geglu ? 4 : 8;  // packed registers per row: 32 outputs / 64 outputs
      unsigned w[TM][WR];
      u32x4 rr[2][2];  // residual, one row ahead
      auto res_load = [&](int i) __attribute__((always_inline)) {
        const int m = m0 + i * 16 + l15;
        const unsigned ro = m < p.m ? (unsigned)m * (unsigned)p.ld_res * 2u + (unsigned)n0 * 2u : OOB_V;
        rr[i & 1][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro + (unsigned)(8 * g) * 2u, 0, 0));
        rr[i & 1][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro + (unsigned)(32 + 8 * g) * 2u, 0, 0));
      };
      if (EPI == 0) res_load(0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (EPI == 0 && i + 1 < TM) res_load(i + 1);
        const float2 st = make_float2(__uint_as_float(st_nx[i & 1][0]), __uint_as_float(st_nx[i & 1][1]) + rstd_id);
        if (EPI != 0 && i + 2 < TM) st_nx[i & 1] = st_load(i + 2);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float x = acc[i][j][r];
            if (EPI != 0) x = st.y * (x - st.x * cs[j][r]);
            v[r] = x + bi[j][r];
          }
          if (EPI == 2) {
#ifdef CA_EXPERIMENTS
            if (p.dbg == 2 || p.dbg == 5) {  // (timing only: no GELU arithmetic)
              w[i][j] = pack2<DT>(v[0] + v[1], v[2] + v[3]);
              continue;
            }
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = Elem<DT>::to_f(Elem<DT>::from_f(v[r]));  // (the Linear's output is rounded first)
            const f32x2 gg = gelu_erf_f2((f32x2){v[1], v[3]});
            w[i][j] = pack2<DT>(v[0] * gg[0], v[2] * gg[1]);
          } else {
            w[i][2 * j] = pack2<DT>(v[0], v[1]);
            w[i][2 * j + 1] = pack2<DT>(v[2], v[3]);
          }
        }
        if (EPI == 0) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const unsigned r_ = rr[i & 1][k >> 2][k & 3];
            if (DT == CA_F16) {  // fp16 + fp16 is exact in fp32: the packed add rounds exactly like the fp32 path
              unsigned s_;
              asm("v_pk_add_f16 %0, %1, %2" : "=v"(s_) : "v"(w[i][k]), "v"(r_));
              w[i][k] = s_;
            } else {
              w[i][k] = pack2<DT>(Elem<DT>::to_f((u16)(w[i][k] & 0xffffu)) + Elem<DT>::to_f((u16)(r_ & 0xffffu)),
                                  Elem<DT>::to_f((u16)(w[i][k] >> 16)) + Elem<DT>::to_f((u16)(r_ >> 16)));
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + i * 16 + l15;
        const unsigned ro = (m < p.m && p.dbg != 1 && p.dbg != 5) ? (unsigned)m * (unsigned)p.ldc * 2u + (unsigned)(geglu ? (n0 >> 1) : n0) * 2u : OOB_V;  // (dbg 1: timing without stores)
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){w[i][0], w[i][1], w[i][2], w[i][3]}, rs_c, ro + (unsigned)(8 * g) * 2u, 0, 0);
        if (EPI != 2) __builtin_amdgcn_raw_buffer_store_b128((u32x4){w[i][4], w[i][5], w[i][6], w[i][7]}, rs_c, ro + (unsigned)(32 + 8 * g) * 2u, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// Fused feed-forward of the 64x64-latent level (round 4): y = GEGLU(LN(x) W1^T + b1) W2^T + b2 + residual in ONE launch, for
// C = 320 (inner width 1280): the 335 MB intermediate of a 131072-row launch never leaves the chip.
//
// Unfused (ca_gemm x 2) the pair costs ~445 us inside a graph: the GEGLU projection writes 335 MB at the chip's store rate, the
// output projection reads them back and is HBM-bound (503 MB).  Here a block owns a 128-row tile of x -- the activation-resident
// layout of ca_gemm_ar.h: 80 KB of LDS, XOR-swizzled, W fragments straight from L2 out of fragment-ordered copies -- and its
// eight waves split into two ROLES that share the four SIMDs pairwise:
//
//   producers (waves 0..3)  round r: the GEGLU item of panel 4r + w exactly as k_gemm_ar computes it (128 x 64 weight rows = 32
//                           outputs, 320 MFMAs, the polynomial-GELU epilogue), but the 128 x 32 block of h goes into an LDS
//                           staging tile H[r & 1] (128 rows x 128 h columns, 32 KB, 16-way XOR swizzle) instead of memory;
//   consumers (waves 4..7)  round r: y[128 x 80 of wave wc] += H[(r - 1) & 1] (128 x 128) . W2[80 x 128]^T: 160 MFMAs, A fragments
//                           from the staging tile, W2 fragments from L2; 160 accumulators stay in registers for the whole tile.
//
// Two barriers per round put the consumers' MFMAs beside the producers' epilogue VALU work (the part of the GEGLU kernel nothing
// overlapped with at two free-running blocks per CU): { producers: K loop | consumers: wait } B1 { producers: epilogue -> H |
// consumers: stage 2 of the previous round } B2.  Producers never store to memory, so nothing in their W stream waits behind a
// store (the in-order VMEM counter, ca_gemm_ar.h); consumers store once per tile.
// LayerNorm: four threads per row normalise the landed tile IN PLACE ((x - mean) * rstd rounded to the activation type, gamma / beta
// folded into W1 / bias1), so the GEGLU epilogue is acc + bias.  The residual of the reference's feed-forward is the block's own
// input (attention.py:350-357 `ff(norm3(x)) + x`): read from global like any residual (the LDS copy is normalised).
// s_memtime stamps of a round (tools/ff_stamps.py, experiments build): producer K loop 6.0-6.3 k cycles (19 per MFMA), epilogue
// 9.4 k beside the consumer's 160 MFMAs (4.9 k): on this chip a SIMD's VALU and matrix work do not overlap -- the consumer's MFMAs
// take 31 cycles each beside the epilogue, 17 alone -- so a round costs the SUM of its MFMA and VALU work whatever the schedule;
// what is left to gain is instruction count (the degree-9 GELU polynomial is 60 % of the epilogue).
//
// Fragment-ordered W2 (ca_pack_w2_frag): 16-byte piece L of MFMA tile j (5 per 80-column wave range, the 16-byte-store interleave of
// ca_gemm_ps.h) of consumer wave wc of h chunk pn (40 chunks of 32):  W2[wc * 80 + col5(j, L & 15)][pn * 32 + (L >> 4) * 8 : +8]
//   at (((pn * 4 + wc) * 5 + j) * 64 + L) * 8.

__device__ __forceinline__ int ca_ff_col5(int j, int i) {  // (= ca_ps_col without GEGLU)
  if (j == 4) return 64 + i;
  return 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3);
}

__global__ __launch_bounds__(256) void k_pack_w2_frag(const u16* __restrict__ w, u16* __restrict__ dst) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // one 16-byte piece each: 320 x 1280 / 8 = 51200
  if (idx >= 320 * 160) return;
  const int L = idx & 63;
  int t = idx >> 6;
  const int j = t % 5;
  t /= 5;
  const int wc = t & 3;
  const int pn = t >> 2;
  const int row = wc * 80 + ca_ff_col5(j, L & 15);
  st16(dst + (int64_t)idx * 8, ld16(w + (int64_t)row * 1280 + pn * 32 + (L >> 4) * 8));
}

struct FfParams {
  const u16* x;       // [M, 320] rows at stride lda
  const u16* w1f;     // GEGLU projection, fragment order of ca_gemm_ar.h (geglu = 1), LayerNorm gamma folded in
  const float* bias1; // [2560] (value / gate interleaved), W beta + b
  const float* cs1;   // [2560] column sums of the folded W1
  const u16* w2f;     // output projection in the order above
  const float* bias2; // [320] or NULL
  const u16* res;     // [M, 320] rows at stride ld_res, or NULL
  u16* y;             // [M, 320] rows at stride ldc
  const float* ln_stats;  // [M][2] (mean, rstd) or NULL: computed here
  int64_t lda, ldc, ld_res;
  int m;
  float ln_eps;
  unsigned x_bytes, y_bytes, res_bytes;
  AttnOutParams out;  // ABI v12: the transformer's proj_out + bias + residual behind the feed-forward (ca_attn_out.h), or wof = NULL
};

#ifdef CA_EXPERIMENTS
// (timing experiments: shader-clock stamps of block 0, wave 0 (producer) / wave 4 (consumer), first tile: tools/ff_stamps.py)
__device__ unsigned long long ca_ff_stamps[2][256];
#define CA_FF_STAMP(TAG)                                                                                         \
  if (blockIdx.x == 0 && tile == (int)blockIdx.x && (wid == 0 || wid == 4) && lane == 0 && stamp_i < 120) {     \
    ca_ff_stamps[wid >> 2][2 * stamp_i] = __builtin_readcyclecounter();                                          \
    ca_ff_stamps[wid >> 2][2 * stamp_i + 1] = (TAG);                                                              \
    ++stamp_i;                                                                                                   \
  }
#else
#define CA_FF_STAMP(TAG)
#endif

template <int DT, bool OUT>
__global__ __launch_bounds__(512, 2) void k_ff_fused(FfParams p, int tiles_m) {
#ifdef CA_EXPERIMENTS
  int stamp_i = 0;
#endif
  constexpr int K = 320, KQ = 10, TM = 8, TN = 4, BM = 128, PANELS = 40, ROUNDS = 10;
  constexpr int ROWB = K * 2;
  constexpr int H_BYTES = BM * 256;
  // three arrays, not one: hipcc drains the VMEM counter in front of every LDS access narrower than 16 bytes that MAY alias an
  // LDS-DMA destination -- only the x tile is one
  __shared__ __attribute__((aligned(16))) unsigned char smem[BM * ROWB];
  __shared__ __attribute__((aligned(16))) unsigned char smem_h[2 * H_BYTES];
  static_assert(BM * ROWB + 2 * H_BYTES <= 160 * 1024, "LDS");
  constexpr unsigned OOB_V = 0x80000000u;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wid < 4;
  const int wq = wid & 3;  // producer: h slot of the round; consumer: 80-column range of y
  const int g = lane >> 4, l15 = lane & 15;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1f, 0, 2560u * 640u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2f, 0, 320u * 2560u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias1, 0, 2560u * 4u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias2 ? (const void*)p.bias2 : (const void*)p.w2f), 0, p.bias2 ? 320u * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? (const void*)p.res : (const void*)p.y), 0, p.res ? p.res_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.y_bytes, 0x00020000);
  // the output stage (proj_out of the transformer the feed-forward ends: y_out = y Wout^T + b_out + residual_out)
  constexpr bool with_out = OUT;  // (a template parameter: the plain instantiation keeps its register allocation)
  const __amdgpu_buffer_rsrc_t rs_wo = __builtin_amdgcn_make_buffer_rsrc((void*)(with_out ? (const void*)p.out.wof : (const void*)p.w2f), 0, with_out ? (unsigned)CA_WOUT_ELEMS * 2u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out.bias ? (const void*)p.out.bias : (const void*)p.w2f), 0, p.out.bias ? 320u * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_ro = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out.res ? (const void*)p.out.res : (const void*)p.w2f), 0, p.out.res ? p.out.res_bytes : 0u, 0x00020000);

  // A-tile fragment addresses (ca_gemm_ar.h): four bases, every fragment = base + a 16-bit immediate
  const int f_sw = (l15 >> 1) & 7;
  const int fa_lane = l15 * ROWB + (((f_sw >> 2) << 2) + (g ^ (f_sw & 3))) * 16;
  int fa_b[2][2] = {{fa_lane, fa_lane ^ 64}, {fa_lane + 4 * 16 * ROWB, (fa_lane ^ 64) + 4 * 16 * ROWB}};
  asm volatile("" : "+v"(fa_b[0][0]), "+v"(fa_b[0][1]), "+v"(fa_b[1][0]), "+v"(fa_b[1][1]));
  int lane_k = lane;
  asm volatile("" : "+v"(lane_k));
  const unsigned wv = (unsigned)lane_k * 16u;

  // the x tile of a block's NEXT row tile is requested by the producers as soon as their last K loop of the current one is over
  // (round 9, behind B1): it lands beside the last epilogue, the consumers' drain round and their y epilogue
  auto x_tile_dma = [&](int tile_) __attribute__((always_inline)) {
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int m0_ = tile_ * BM;
#pragma unroll
    for (int q = 0; q < 20; ++q) {
      const unsigned idx = (unsigned)((wq * 20 + q) * 64 + lane_o);
      const unsigned r = __umulhi(idx >> 3, 0xCCCCCCCDu) >> 2;  // idx / 40
      const unsigned cp = idx - r * 40u;
      const unsigned c = cp ^ ((r >> 1) & 7u);
      const unsigned off = (m0_ + (int)r) < p.m ? (unsigned)(m0_ + (int)r) * (unsigned)p.lda * 2u + c * 16u : OOB_V;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + (wq * 20 + q) * 1024), 16, off, 0, 0, 0);
    }
  };
  if (producer && (int)blockIdx.x < tiles_m) x_tile_dma(blockIdx.x);

  for (int tile = blockIdx.x; tile < tiles_m; tile += gridDim.x) {
    const int m0 = tile * BM;
    // h chunk of (round r, slot s) = (4 r + s + rot) mod 40, rot = 4 (tile mod 8): the blocks of a round of tiles start on different
    // W panels (all on panel 0: 392-420 instead of 368-405 us).  y then sums its 40 chunks in an order that depends on the tile
    // index mod 8 only: rows 8 k tiles apart get the same bits for the same input -- the two CFG halves of identical inputs
    // (tests/test_fullsize_gpu.py) whenever an image is a whole multiple of 8 tiles (1024 rows: every latent size of the configs).
    const int rot = 4 * (tile & 7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (producers: this tile's x pieces; consumers: the previous tile's stores)
    __syncthreads();
    {  // LayerNorm: four threads per row, ten 16-byte pieces each.  The tile is normalised IN PLACE -- x' = (x - mean) * rstd, rounded
       // to the activation type as the reference's LayerNorm output is (gamma and beta live in W1 and bias1: layers.LnFold) -- so that
       // the GEGLU epilogue is acc + bias: folding (mean, rstd) into the epilogue instead (ca_gemm) costs two more VALU operations and
       // two more operands per value there, and on this chip a SIMD's VALU and matrix work do not overlap (tools/ff_stamps.py).
      const int r = tid >> 2, h = tid & 3;
      unsigned char* src = smem + r * ROWB + h * 160;
      u32x4 v[10];
#pragma unroll
      for (int q = 0; q < 10; ++q) v[q] = ld16(src + q * 16);
      float2 st;
      if (p.ln_stats) {
        st = (m0 + r < p.m) ? *reinterpret_cast<const float2*>(p.ln_stats + (int64_t)(m0 + r) * 2) : make_float2(0.f, 0.f);
      } else {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int q = 0; q < 10; ++q) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a0 = Elem<DT>::to_f((u16)(v[q][e] & 0xffffu)), a1 = Elem<DT>::to_f((u16)(v[q][e] >> 16));
            s += a0 + a1;
            ss = fmaf(a0, a0, fmaf(a1, a1, ss));
          }
        }
        s += __shfl_xor(s, 1);
        ss += __shfl_xor(ss, 1);
        s += __shfl_xor(s, 2);
        ss += __shfl_xor(ss, 2);
        const float mean = s * (1.f / K);
        st = make_float2(mean, rsqrtf(fmaxf(ss * (1.f / K) - mean * mean, 0.f) + p.ln_eps));
      }
      const float nb = -st.x * st.y;
#pragma unroll
      for (int q = 0; q < 10; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = Elem<DT>::to_f((u16)(v[q][e] & 0xffffu)), a1 = Elem<DT>::to_f((u16)(v[q][e] >> 16));
          v[q][e] = pack2<DT>(fmaf(a0, st.y, nb), fmaf(a1, st.y, nb));
        }
        st16(src + q * 16, v[q]);
      }
    }
    __syncthreads();

    CA_FF_STAMP(0)
    if (producer) {
      // ================================================================ producers: GEGLU items -> H
      u32x4 fa[TM], fb[2][TN];
      auto chunk_of = [&](int r) __attribute__((always_inline)) -> int {
        int c = 4 * r + wq + rot;
        return c >= PANELS ? c - PANELS : c;
      };
      {
        const unsigned wb0 = (unsigned)chunk_of(0) * (unsigned)(KQ * TN * 1024);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w1, wv, wb0 + (unsigned)(c * TN + j) * 1024u, 0));
      }
      for (int r = 0; r < ROUNDS; ++r) {
        const int pn = chunk_of(r);
        const unsigned wbase = (unsigned)pn * (unsigned)(KQ * TN * 1024);
        f32x4 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = ld16(smem + fa_b[i >> 2][0] + (i & 3) * 16 * ROWB);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
          const int nk = kq + 1;
          const int fa_off = (nk >> 1) * 128;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              if (kq == 0) acc[i][j] = Elem<DT>::mfma(fb[0][j], fa[i], (f32x4){0.f, 0.f, 0.f, 0.f});
              else acc[i][j] = Elem<DT>::mfma(fb[kq & 1][j], fa[i], acc[i][j]);
              if (j == TN - 1 && nk < KQ) {
                __builtin_amdgcn_sched_barrier(0);
                fa[i] = ld16(smem + fa_b[i >> 2][nk & 1] + fa_off + (i & 3) * 16 * ROWB);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            if (kq + 2 < KQ) fb[kq & 1][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w1, wv, wbase + (unsigned)(((kq + 2) * TN + j) * 1024), 0));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // ---- B1: the consumers' stage 2 of round r - 1 starts beside this epilogue
        CA_FF_STAMP(1)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        CA_FF_STAMP(2)
        // (every producer's reads of the x tile are over; with the output stage the buffer first holds the y tile: requested behind it)
        if (r + 1 == ROUNDS && !with_out && tile + (int)gridDim.x < tiles_m) x_tile_dma(tile + gridDim.x);
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int l15 = lane_e & 15, g = lane_e >> 4;
        if (r + 1 < ROUNDS) {  // the next round's first two W1 chunks
          const unsigned wb1 = (unsigned)chunk_of(r + 1) * (unsigned)(KQ * TN * 1024);
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w1, wv, wb1 + (unsigned)(c * TN + j) * 1024u, 0));
        }
        f32x4 bi[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b1, (unsigned)(pn * 64 + 16 * g + 4 * j) * 4u, 0, 0));
        unsigned char* hb = smem_h + (r & 1) * H_BYTES;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = i * 16 + l15;
          unsigned w[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            // (value, gate) pairs: the gates go through the activation type (the Linear's output as the GELU sees it), the values
            // meet them in fp32 -- one rounding of the product instead of two
            const float g0 = Elem<DT>::to_f(Elem<DT>::from_f(acc[i][j][1] + bi[j][1])), g1 = Elem<DT>::to_f(Elem<DT>::from_f(acc[i][j][3] + bi[j][3]));
            const f32x2 gg = gelu_erf_f2((f32x2){g0, g1});
            w[j] = pack2<DT>((acc[i][j][0] + bi[j][0]) * gg[0], (acc[i][j][2] + bi[j][2]) * gg[1]);
          }
          // h[row][slot wq: 32 columns, this lane's 8 at 8 g]: logical 16-byte chunk 4 wq + g of the 256-byte row, at chunk ^ (row & 15)
          *reinterpret_cast<u32x4*>(hb + row * 256 + (((wq * 4 + g) ^ l15) << 4)) = (u32x4){w[0], w[1], w[2], w[3]};
          __builtin_amdgcn_sched_barrier(0);
        }
        // ---- B2: H[r & 1] is complete; the consumers' reads of H[(r - 1) & 1] are over
        CA_FF_STAMP(3)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        CA_FF_STAMP(4)
      }
      // the drain round (consumers: stage 2 of round 9) and the consumers' epilogue: two more barriers
      __builtin_amdgcn_s_barrier();
      if (with_out) {  // rows 0..63 of the y tile the consumers leave in the x buffer: this wave's 64 x 80 patch of y Wout^T
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const unsigned row_base = (unsigned)(m0 + (lane_e & 15));
        AttnOutRegs R;
        attn_out_prefetch<DT>(R, p.out, rs_wo, rs_ro, wid, lane_e, row_base, 16u, 32u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the y tile is complete
        attn_out_run<DT>(R, smem, fa_b, p.out, rs_wo, rs_bo, rs_y, wid, lane_e, row_base, 16u, 32u, (unsigned)p.ldc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave has read the y tile: the next x tile may land
        if (tile + (int)gridDim.x < tiles_m) x_tile_dma(tile + gridDim.x);
      } else {
        __builtin_amdgcn_s_barrier();
      }
    } else {
      // ================================================================ consumers: y += H . W2^T
      f32x4 yacc[TM][5];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) yacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      u32x4 fb2[2][5];
      auto w2_load = [&](int r, int s, int b) __attribute__((always_inline)) {
        int c = 4 * r + s + rot;
        c = c >= PANELS ? c - PANELS : c;
        const unsigned base = (unsigned)((c * 4 + wq) * 5) * 1024u;
#pragma unroll
        for (int j = 0; j < 5; ++j) fb2[b][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, wv, base + (unsigned)j * 1024u, 0));
      };
      for (int r = 0; r <= ROUNDS; ++r) {
        if (r >= 1) w2_load(r - 1, 0, 0);  // (lands while the producers finish their K loop)
        CA_FF_STAMP(1)
        __builtin_amdgcn_s_barrier();      // B1
        CA_FF_STAMP(2)
        if (r >= 1) {
          const unsigned char* hb = smem_h + ((r - 1) & 1) * H_BYTES;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) w2_load(r - 1, s + 1, (s + 1) & 1);
            u32x4 fa2[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa2[i] = ld16(hb + (i * 16 + l15) * 256 + (((s * 4 + g) ^ l15) << 4));
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
              for (int i = 0; i < TM; ++i) yacc[i][j] = Elem<DT>::mfma(fb2[s & 1][j], fa2[i], yacc[i][j]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        CA_FF_STAMP(3)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (r < ROUNDS) __builtin_amdgcn_s_barrier();  // B2
        CA_FF_STAMP(4)
      }
      // ---- y epilogue: + bias + residual, 16-byte stores (pairs of MFMA tiles interleaved, ca_gemm_ps.h)
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int l15 = lane_e & 15, g = lane_e >> 4;
      const int n0 = wq * 80;
      f32x4 bi[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int c0 = n0 + (j < 4 ? 32 * (j >> 1) + 8 * g + 4 * (j & 1) : 64 + 4 * g);
        bi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b2, (unsigned)c0 * 4u, 0, 0));
      }
      u32x4 rr[2][2];
      u32x2 r8[2];
      auto res_load = [&](int i) __attribute__((always_inline)) {
        const int m = m0 + i * 16 + l15;
        const unsigned ro = m < p.m ? (unsigned)m * (unsigned)p.ld_res * 2u + (unsigned)n0 * 2u : OOB_V;
        rr[i & 1][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro + (unsigned)(8 * g) * 2u, 0, 0));
        rr[i & 1][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro + (unsigned)(32 + 8 * g) * 2u, 0, 0));
        r8[i & 1] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_res, ro + (unsigned)(64 + 4 * g) * 2u, 0, 0));
      };
      unsigned w[TM][10];
      res_load(0);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (i + 1 < TM) res_load(i + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          w[i][2 * j] = pack2<DT>(yacc[i][j][0] + bi[j][0], yacc[i][j][1] + bi[j][1]);
          w[i][2 * j + 1] = pack2<DT>(yacc[i][j][2] + bi[j][2], yacc[i][j][3] + bi[j][3]);
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) {
          const unsigned r_ = k < 8 ? rr[i & 1][k >> 2][k & 3] : r8[i & 1][k & 1];
          if (DT == CA_F16) {  // fp16 + fp16 is exact in fp32: the packed add rounds exactly like the fp32 path
            unsigned s_;
            asm("v_pk_add_f16 %0, %1, %2" : "=v"(s_) : "v"(w[i][k]), "v"(r_));
            w[i][k] = s_;
          } else {
            w[i][k] = pack2<DT>(Elem<DT>::to_f((u16)(w[i][k] & 0xffffu)) + Elem<DT>::to_f((u16)(r_ & 0xffffu)),
                                Elem<DT>::to_f((u16)(w[i][k] >> 16)) + Elem<DT>::to_f((u16)(r_ >> 16)));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (with_out) {
        // y (rounded to the activation type, as the two-launch path stores it) -> the x buffer in the tile layout, then rows 64..127
        // of the output stage; the producers run rows 0..63
        const unsigned row_base = (unsigned)(m0 + 64 + l15);
        const int fs = (l15 >> 1) & 7;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          unsigned char* rowp = smem + (i * 16 + l15) * ROWB;
          const int c0 = n0 + 8 * g, c1 = n0 + 32 + 8 * g, c2 = n0 + 64 + 4 * g;
          st16(rowp + (((c0 >> 3) ^ fs) << 4), (u32x4){w[i][0], w[i][1], w[i][2], w[i][3]});
          st16(rowp + (((c1 >> 3) ^ fs) << 4), (u32x4){w[i][4], w[i][5], w[i][6], w[i][7]});
          ca_lds_store8(rowp + (((c2 >> 3) ^ fs) << 4) + (c2 & 7) * 2, (u32x2){w[i][8], w[i][9]});
        }
        AttnOutRegs R;
        attn_out_prefetch<DT>(R, p.out, rs_wo, rs_ro, wid, lane_e, row_base, 16u, 32u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the y tile is complete (matches the producers' barrier)
        attn_out_run<DT>(R, smem, fa_b, p.out, rs_wo, rs_bo, rs_y, wid, lane_e, row_base, 16u, 32u, (unsigned)p.ldc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave has read the y tile
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int m = m0 + i * 16 + l15;
          const unsigned ro = m < p.m ? (unsigned)m * (unsigned)p.ldc * 2u + (unsigned)n0 * 2u : OOB_V;
          __builtin_amdgcn_raw_buffer_store_b128((u32x4){w[i][0], w[i][1], w[i][2], w[i][3]}, rs_y, ro + (unsigned)(8 * g) * 2u, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128((u32x4){w[i][4], w[i][5], w[i][6], w[i][7]}, rs_y, ro + (unsigned)(32 + 8 * g) * 2u, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64((u32x2){w[i][8], w[i][9]}, rs_y, ro + (unsigned)(64 + 4 * g) * 2u, 0, 0);
        }
        __builtin_amdgcn_s_barrier();  // (matches the producers' last barrier: the tile is done)
      }
    }
  }
}

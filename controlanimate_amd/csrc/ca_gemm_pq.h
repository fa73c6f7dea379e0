// Persistent streaming GEMM / implicit-GEMM 3x3 convolution, 256 x 320 block tile, 128 x 80 wave tiles (round 3).
//
// Why another tile: what bounds the 128 x 320 kernels (ca_gemm_pp2.h, ca_gemm_ps.h) is the LDS port -- a wave with a 64 x 80
// output patch reads (64 + 80) fragment rows of 64 bytes per 20 MFMAs, 115 B/clk per CU with eight waves against a port of
// 128 B/clk.  A 128 x 80 patch reads (128 + 80) rows per 40 MFMAs: 83 B/clk, and the block's global -> LDS stream drops from
// 44 to 28 B/clk per CU.  The price is registers (160 accumulators + 52 fragment registers of a 256-register budget at two
// waves per SIMD), so this kernel keeps only what the long-K launches need: bias, up to two row-bias groups, alpha, residual.
//
// Block: 8 waves = 2 groups (wr: 128-row halves) x 4 (wc: 80-column quarters).  A K tile (64 deep: A 32 KB | B0 16 KB | B1 24 KB,
// the layout and swizzle of ca_gemm_ps.h) is ONE DMA unit; two LDS buffers.  A K tile is computed in two phases, one per
// 32-deep half: { read 8 + 5 fragments ; barrier ; 40 MFMAs ; barrier }.  The groups run one barrier apart, so one group's MFMA
// segment covers the other's fragment reads.  Intervals (between barriers), K tile t:
//     I0: G0 reads h0(t)   | G1 MFMA h1(t-1)          I2: G0 reads h1(t) | G1 MFMA h0(t)
//     I1: G0 MFMA h0(t)    | G1 reads h0(t)           I3: G0 MFMA h1(t)  | G1 reads h1(t)
// Buffer (t-1) & 1 is free from I0 on (G1's last reads of it were in the interval before): K tile t + 1 goes into it, issued by
// every wave behind its reads of h0(t) (G0 in I0, G1 in I1), and every wave confirms its own pieces at the end of I3 (G0 behind
// its MFMAs, G1 behind its reads) by the LDS flag of ca_gemm_ps.h; the barrier that ends I3 publishes it and G0 reads the tile
// in I4.  A unit has two to three intervals to land; there is no `s_waitcnt vmcnt` in the loop.
//
// Persistent over tiles, epilogue straight from the accumulators with 16-byte stores (weight rows of MFMA-tile pairs
// interleaved: ca_ps_col), stores fire-and-forget, both groups' epilogues in the same interval -- all as in ca_gemm_ps.h.
// Requirements: N % 320 == 0, >= 2 K tiles, fp16 / bf16 output, no row sums / activation / second dense source, LayerNorm fold
// and GEGLU only without a residual (EPI = 1), 32-bit byte offsets, row-bias groups of a multiple of 128 rows; convolutions: pad 1, no upsampling, < 2^23
// input pixels.

// Timing-only ablations (never in a shipped library: results are wrong): -DCA_PQ_ABLATE=bits, 1 = no global -> LDS units after
// a block's first two, 2 = no fragment reads, 4 = no MFMAs, 8 = no barriers inside the K loop, 16 = every tile streams the operands of
// tile (0, 0) (the stream then comes out of L2: what the latency of the Infinity Cache costs).
#ifdef CA_PQ_ABLATE
#define CA_PQ_ABL(bit) (((CA_PQ_ABLATE) & (bit)) != 0)
#else
#define CA_PQ_ABL(bit) false
#endif

// EPI = 0: bias, row bias, alpha, residual (GEMM and convolution).  EPI = 1 (dense only): LayerNorm fold with finished (mean,
// rstd) per row, bias, alpha and optionally GEGLU; no residual, no row bias.  EPI = 2 (dense only): EPI = 0 + row sums of the
// stored output for the next LayerNorm (ca_gemm_args.row_sums_out), one partial sum per 80-column wave quarter.
template <int DT, int MODE, int EPI = 0>
__global__ __launch_bounds__(512, 2) void k_gemm_pq(GemmKParams p, int tiles_total, unsigned c_bytes, unsigned res_bytes) {
  constexpr int BM = 256, BN = 320, KT = 64;
  constexpr int TM = 8, TN = 5;
  constexpr int A_ROWS = 256, B0_ROWS = 128, B1_ROWS = 192;
  constexpr int OFF_A = 0, OFF_B0 = A_ROWS * KT, OFF_B1 = (A_ROWS + B0_ROWS) * KT;
  constexpr int BUF = (A_ROWS + B0_ROWS + B1_ROWS) * KT;  // elements of one K tile (72 KB)
  constexpr int PAR_BASE = 2 * BUF * 2;
  constexpr int P_BI = 0, P_RB0 = 1280, P_RB1 = 2560, P_CS = 3840, PSET = EPI == 1 ? 5120 : 4096;  // per tile: bias | row bias group 0 | group 1 | column sums, 320 floats each
  constexpr int FLAG_BASE = PAR_BASE + 2 * PSET;  // per wave: 2 flag slots of 256 B (buffer parity)
  constexpr int SMEM_BYTES = FLAG_BASE + 8 * 2 * 256;
  static_assert(SMEM_BYTES <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char smem_b[SMEM_BYTES];
  u16* const smem = reinterpret_cast<u16*>(smem_b);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int g = lane >> 4, l15 = lane & 15;

  const int tiles_n = p.n / BN;
  const int tiles_m = (p.m + BM - 1) / BM;
  const int G = gridDim.x;
  const int bslot = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int my_tiles = bslot < tiles_total ? (tiles_total - bslot + G - 1) / G : 0;
  if (my_tiles == 0) return;

  for (int i = tid; i < 8 * 2 * 64; i += 512) reinterpret_cast<unsigned*>(smem_b + FLAG_BASE)[i] = 0xFFFFFFFFu;
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, p.a2 ? p.a2_bytes : p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)p.w), 0, p.bias ? (unsigned)p.n * 4u : 0u, 0x00020000);
  const bool geglu = EPI == 1 && p.geglu != 0;
  const unsigned long long res_addr = (unsigned long long)(p.res ? (const void*)p.res : (const void*)p.c);
  const u32x4 rs_res = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)res_addr), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((res_addr >> 32) & 0xffffu)),
                        (unsigned)__builtin_amdgcn_readfirstlane((int)(p.res ? res_bytes : 0u)), 0x00020000u};
  const unsigned rb_groups = p.rowbias ? (unsigned)((p.m + p.rows_per_group - 1) / p.rows_per_group) : 0u;
  const __amdgpu_buffer_rsrc_t rs_rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.rowbias ? (const void*)p.rowbias : (const void*)p.w), 0,
                                                                         p.rowbias ? (unsigned)(((int64_t)(rb_groups - 1) * p.ld_rowbias + p.n) * 4) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_seq = __builtin_amdgcn_make_buffer_rsrc((void*)ca_seq_table.v, 0, 4096u, 0x00020000);

#ifdef CA_STAMPS  // (python -m controlanimate_amd._build --experiments --stamps: the stamp code costs registers -- the 256 x 320 kernel spills with it)
  unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(p.partial);
  int stamp_i = 0;
  auto stamp = [&](int tag) __attribute__((always_inline)) {
    if (p.dbg == 9 && blockIdx.x == 0 && (wid == 0 || wid == 4) && lane == 0 && stamp_i < 1000) {
      stamps[(wid >> 2) * 2048 + 2 * stamp_i] = __builtin_readcyclecounter();
      stamps[(wid >> 2) * 2048 + 2 * stamp_i + 1] = (unsigned long long)tag;
      ++stamp_i;
    }
  };
#else
  auto stamp = [&](int) __attribute__((always_inline)) {};
#endif
  auto swz = [](int row) { return (row >> 1) & 7; };
  const int kc = p.c1 + p.c2;
  const int kct = kc / KT;
  const unsigned wld = (unsigned)(p.taps * kc);
  const int nt = p.taps * kct;         // K tiles per output tile (>= 2)
  const int total_kt = my_tiles * nt;  // K tiles of this block's whole stream

  // ---------------------------------------------------------------- DMA side (one K tile ahead)
  constexpr unsigned OOB_V = 0x80000000u;
  unsigned b0_v[2] = {0, 0}, b1_v[3] = {0, 0, 0};
  unsigned a_v[4] = {0, 0, 0, 0};  // dense: byte offset of the lane's row + chunk; conv: tap-0 pixel index + which taps exist (dma_set_tile)
  int d_seq = 0, d_t = 0, d_tap = 0, d_c0 = 0, d_n = 0;
  bool d_live = true;
  unsigned char* const my_flags = smem_b + FLAG_BASE + wid * 2 * 256;

  auto tile_of = [&](int seq, int& tm, int& tn) __attribute__((always_inline)) -> bool {
    const int id = seq * G + bslot;
    if (id >= tiles_total) return false;
    tile_coords((unsigned)id, tiles_m, tiles_n, tm, tn);
    return true;
  };

  auto dma_set_tile = [&](int seq, int m0, int n0) __attribute__((always_inline)) {
    if (CA_PQ_ABL(16)) m0 = 0, n0 = 0;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));  // (opaque: hipcc would hoist the lane-dependent parts out of the tile loop and spill them)
    const int r8 = lane_o >> 3, cp = lane_o & 7;
    const int a_chunk0 = cp ^ swz(r8), b1_chunk0 = cp ^ swz(wid * 24 + r8);  // (A / B0 pieces start at multiples of 16 rows: swz(r8))
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wid * 32 + i * 8 + r8;  // A pieces stage rows wid*32 + 8i + r8: chunk_i = chunk_0 ^ 4(i & 1)
      const bool ok = m < p.m;
      if (MODE == 1) {
        // pixel index of the row's tap (0, 0) -- may be negative at the top / left border -- in bits 0..23 (signed), and which
        // taps exist: bit 24 / 25 = input row of kh = 0 / 2 inside the image, bit 26 / 27 = input column of kw = 0 / 2, bit 28 =
        // the output row exists.  (The middle row / column always exists: pad 1, checked by the launcher.)  Per tap the source
        // pixel is then (tap-0 pixel) + kh * W + kw, a wave-uniform delta.
        const int mm = ok ? m : p.m - 1;
        const int hw = p.hout * p.wout;
        const int img = mm / hw;
        const int rem = mm - img * hw;
        const int ho = rem / p.wout, wo = rem - ho * p.wout;
        const int hi0 = ho * p.stride - p.pad_lo, wi0 = wo * p.stride - p.pad_lo;
        const int p0 = (img * p.hin + hi0) * p.win + wi0;
        a_v[i] = ((unsigned)p0 & 0xFFFFFFu) | (hi0 >= 0 ? 1u << 24 : 0u) | (hi0 + 2 < p.hin ? 1u << 25 : 0u) | (wi0 >= 0 ? 1u << 26 : 0u) | (wi0 + 2 < p.win ? 1u << 27 : 0u) |
                 (ok ? 1u << 28 : 0u);
      } else {
        a_v[i] = ok ? (unsigned)m * (unsigned)p.lda * 2u + (unsigned)((a_chunk0 ^ (4 * (i & 1))) * 16) : OOB_V;
      }
    }
    // row-grouped weights (dense, EPI = 0 only: the Winograd convolutions' sixteen GEMMs in one launch): this row tile's weight matrix
    unsigned wgrp = 0u;
    if (MODE == 0 && EPI == 0 && p.w_group_rows > 0) wgrp = (unsigned)(m0 / p.w_group_rows) * p.w_group_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = wid * 16 + i * 8 + r8;  // B0 local row r: quarter r >> 5, MFMA tile (r >> 4) & 1, fragment row r & 15
      b0_v[i] = wgrp + (unsigned)(n0 + (r >> 5) * 80 + ca_ps_col((r >> 4) & 1, r & 15, geglu)) * wld * 2u + (unsigned)((a_chunk0 ^ (4 * i)) * 16);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int r1 = wid * 24 + i * 8 + r8;  // B1 local row r1: quarter r1 / 48, MFMA tile 2 + (r1 % 48) / 16, fragment row r1 & 15
      b1_v[i] = wgrp + (unsigned)(n0 + (r1 / 48) * 80 + ca_ps_col(2 + (r1 % 48) / 16, r1 & 15, geglu)) * wld * 2u + (unsigned)((b1_chunk0 ^ (4 * (i & 1))) * 16);
    }
    d_tap = 0;
    d_c0 = 0;
    // epilogue parameters of this tile -> parameter set (seq & 1): 320 floats = 1 KB + 256 B per operand (an absent operand
    // has a descriptor of size 0: zeros)
    unsigned char* pset = smem_b + PAR_BASE + (seq & 1) * PSET;
    if (EPI == 1 && wid == 4) {
      const __amdgpu_buffer_rsrc_t rs_cs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ln_colsum ? (const void*)p.ln_colsum : (const void*)p.w), 0, p.ln_colsum ? (unsigned)p.n * 4u : 0u, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_cs, (__attribute__((address_space(3))) void*)(pset + P_CS), 16, (unsigned)n0 * 4u + lane_o * 16u, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_cs, (__attribute__((address_space(3))) void*)(pset + P_CS + 1024), 4, (unsigned)n0 * 4u + 1024u + lane_o * 4u, 0, 0, 0);
    }
    if (wid >= 1 && wid <= 3) {
      const bool has1 = p.rowbias && m0 / p.rows_per_group + 1 < (int)rb_groups;
      const unsigned base = wid == 1 ? (unsigned)n0 * 4u
                                     : wid == 2 ? (unsigned)n0 * 4u + (unsigned)(p.rowbias ? m0 / p.rows_per_group : 0) * (unsigned)p.ld_rowbias * 4u
                                                : (has1 ? (unsigned)n0 * 4u + (unsigned)(m0 / p.rows_per_group + 1) * (unsigned)p.ld_rowbias * 4u : OOB_V);
      unsigned char* dst = pset + (wid == 1 ? P_BI : wid == 2 ? P_RB0 : P_RB1);
      if (wid == 1) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_bi, (__attribute__((address_space(3))) void*)dst, 16, base + lane_o * 16u, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_bi, (__attribute__((address_space(3))) void*)(dst + 1024), 4, base + 1024u + lane_o * 4u, 0, 0, 0);
      } else {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_rb, (__attribute__((address_space(3))) void*)dst, 16, base + lane_o * 16u, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_rb, (__attribute__((address_space(3))) void*)(dst + 1024), 4, base + 1024u + lane_o * 4u, 0, 0, 0);
      }
    }
  };

  // One unit = 9 pieces + the flag.  Both groups issue K tile cv + 1 behind their fragment reads of k half 0: that interval is
  // the short one (its reads take ~400 cycles against ~1100 of the partner group's MFMA segment), so the issue work rides in
  // the partner's shadow.  (First version: group 1 issued in front of its MFMAs -- s_memtime stamps showed 600..1500 cycles of
  // issue work in that interval with group 0 idle at the barrier; issuing piece by piece BETWEEN the MFMA rows does not hide it
  // either: instruction issue is in order, a 40-instruction piece only overlaps the last MFMA before it.)
  auto issue_tile = [&]() __attribute__((always_inline)) {
    if (!d_live) return;
    if (CA_PQ_ABL(1) && d_n >= 2) {
      ++d_n;
      ++d_t;
      return;
    }
    const int par = d_n & 1;
    u16* buf = smem + par * BUF;
    const unsigned wk = (unsigned)(d_tap * kc + d_c0) * 2u;  // weights: K runs over (tap, channel)
    const bool src2 = d_c0 >= p.c1;                          // c1 % 64 == 0: a K tile never straddles the two sources
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_B0 + (wid * 2 + 0) * 8 * KT), 16, b0_v[0], wk, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_B0 + (wid * 2 + 1) * 8 * KT), 16, b0_v[1], wk, 0, 0);
    if (MODE == 1) {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const unsigned cs2 = (unsigned)(src2 ? p.c2 : p.c1) * 2u;
      const int kh = d_tap >= 6 ? 2 : (d_tap >= 3 ? 1 : 0), kw = d_tap - kh * 3;
      const unsigned need = (1u << 28) | (kh == 0 ? 1u << 24 : kh == 2 ? 1u << 25 : 0u) | (kw == 0 ? 1u << 26 : kw == 2 ? 1u << 27 : 0u);
      const int delta = kh * p.win + kw;
      // byte offset inside a pixel: first channel of the K tile + the lane's (swizzled) 16-byte chunk; odd pieces flip chunk bit 2
      const unsigned cadd = (unsigned)(src2 ? d_c0 - p.c1 : d_c0) * 2u + (unsigned)(((lane_o & 7) ^ swz(lane_o >> 3)) * 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned av = a_v[i];
        const int pix = (((int)(av << 8)) >> 8) + delta;  // (24-bit multiplies below: v_mul_lo_u32 is quarter rate; < 2^23 pixels: launcher)
        const unsigned off = __umul24((unsigned)pix, cs2) + (cadd ^ ((i & 1) ? 64u : 0u));
        const unsigned voff = (av & need) == need ? off : OOB_V;
        void* d = buf + OFF_A + (wid * 4 + i) * 8 * KT;
        if (src2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (__attribute__((address_space(3))) void*)d, 16, voff, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)d, 16, voff, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(buf + OFF_A + (wid * 4 + i) * 8 * KT), 16, a_v[i], (unsigned)d_c0 * 2u, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_B1 + (wid * 3 + i) * 8 * KT), 16, b1_v[i], wk, 0, 0);
    // the unit's flag: its sequence number, fetched behind the nine pieces (loads return in order); the head advances
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_seq, (__attribute__((address_space(3))) void*)(my_flags + par * 256), 4, 0u, (unsigned)(d_n & 1023) * 4u, 0, 0);
    ++d_n;
    ++d_t;
    if (p.taps == 1) {
      d_c0 += KT;
    } else if (++d_tap == p.taps) {  // the nine taps of one 64-channel tile follow each other (k_tile_split, tap_inner)
      d_tap = 0;
      d_c0 += KT;
    }
  };
#define CA_PQ_SET_TILE(SEQ)                                  \
  {                                                          \
    int tm_, tn_;                                            \
    d_live = tile_of((SEQ), tm_, tn_);                       \
    if (d_live) dma_set_tile((SEQ), tm_ * BM, tn_ * BN);     \
  }
  auto advance_and_issue = [&]() __attribute__((always_inline)) {
    if (d_t == nt) {  // the stream enters the next tile
      d_t = 0;
      ++d_seq;
      CA_PQ_SET_TILE(d_seq)
    }
    issue_tile();
  };

  auto flag_begin = [&](int slot) __attribute__((always_inline)) -> unsigned {
    unsigned fv;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(my_flags + slot * 256);
    asm volatile("ds_read_b32 %0, %1" : "=v"(fv) : "v"(addr) : "memory");
    return fv;
  };
  // (bounded: a hung wave would take the whole device down; wrong results are caught by the tests, a hang is not)
  auto flag_finish = [&](int slot, int seqno, unsigned fv) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fv)::"memory");
    if (CA_PQ_ABL(1)) return;
    const unsigned want = (unsigned)(seqno & 1023);
    if ((unsigned)__builtin_amdgcn_readfirstlane(fv) == want) return;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(my_flags + slot * 256);
    for (unsigned spins = 0; spins < (1u << 20); ++spins) {
      __builtin_amdgcn_s_sleep(1);
      unsigned v;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
      if ((unsigned)__builtin_amdgcn_readfirstlane(v) == want) return;
    }
    // gave up polling (a pre-empted or very slow DMA): loads return in order, so draining the wave's VMEM counter is the
    // correct -- merely slower -- way to know the unit has landed; never continue on stale LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  // ---------------------------------------------------------------- compute side
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int fa_base, fb0_base, fb1_base;  // k half 0; the second half flips chunk bit 2 = element offset bit 5
  {
    const int ra = wr * 128 + l15, rb0 = wc * 32 + l15, rb1 = wc * 48 + l15;
    fa_base = OFF_A + ra * KT + ((g ^ swz(ra)) << 3);
    fb0_base = OFF_B0 + rb0 * KT + ((g ^ swz(rb0)) << 3);
    fb1_base = OFF_B1 + rb1 * KT + ((g ^ swz(rb1)) << 3);
  }
  u32x4 fa[TM], fb[TN];
  if (CA_PQ_ABL(2)) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = (u32x4){(unsigned)lane * 0x9E3779B9u + i, (unsigned)lane * 0x85EBCA6Bu, (unsigned)lane * 0xC2B2AE35u, (unsigned)lane * 0x27D4EB2Fu} & 0x3BFF3BFFu;
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = (u32x4){(unsigned)lane * 0x165667B1u + j, (unsigned)lane * 0xD3A2646Cu, (unsigned)lane * 0xFD7046C5u, (unsigned)lane * 0xB55A4F09u} & 0x3BFF3BFFu;
  }
  auto read_frags = [&](const u16* buf, int s) __attribute__((always_inline)) {
    if (CA_PQ_ABL(2)) {
#pragma unroll
      for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[i]));
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[j]));
      return;
    }
    const int x = s ? 32 : 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = ld16(buf + (fa_base ^ x) + i * 16 * KT);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[j] = ld16(buf + (fb0_base ^ x) + j * 16 * KT);
#pragma unroll
    for (int j = 0; j < 3; ++j) fb[2 + j] = ld16(buf + (fb1_base ^ x) + j * 16 * KT);
  };
  auto mfma_all = [&]() __attribute__((always_inline)) {
    if (CA_PQ_ABL(4)) return;
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = Elem<DT>::mfma(fb[j], fa[i], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- the epilogue of one tile, from the accumulators: round((acc + bias + rowbias) * alpha) + residual, rounded
  // (branch-free: an absent operand is an exact identity, see ca_gemm_ps.h).  The residual of a row tile is loaded by inline asm
  // one row tile ahead (the compiler would await a visible load together with the stores of the previous rows).
  auto epilogue = [&](int seq, int m0, int n0) __attribute__((always_inline)) {
    const unsigned char* pset = smem_b + PAR_BASE + (seq & 1) * PSET;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int l15 = lane_o & 15, g = lane_o >> 4;
    int rb_sel = 0;
    if (p.rowbias) rb_sel = (m0 + wr * 128) / p.rows_per_group - m0 / p.rows_per_group;  // 0 or 1 (rows_per_group % 128 == 0)
    const unsigned char* rbp = pset + (rb_sel ? P_RB1 : P_RB0);
    f32x4 bi[TN], rb[TN];  // bias / row bias of the lane's 4 columns of MFMA tile j
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c0 = wc * 80 + (j < 4 ? 32 * (j >> 1) + 8 * g + 4 * (j & 1) : 64 + 4 * g);
      bi[j] = *reinterpret_cast<const f32x4*>(pset + P_BI + c0 * 4);
      rb[j] = *reinterpret_cast<const f32x4*>(rbp + c0 * 4);
    }
    auto rowoff_of = [&](int i, int64_t ld) __attribute__((always_inline)) -> unsigned {
      const int m = m0 + wr * 128 + i * 16 + l15;
      return m < p.m ? (unsigned)m * (unsigned)ld * 2u + (unsigned)(n0 + wc * 80) * 2u : OOB_V;
    };
    u32x4 r16[2][2];
    u32x2 r8[2];
    r16[0][0] = r16[0][1] = r16[1][0] = r16[1][1] = (u32x4){0u, 0u, 0u, 0u};
    r8[0] = r8[1] = (u32x2){0u, 0u};
    auto res_load = [&](int i, int slot) __attribute__((always_inline)) {
      const unsigned ro = rowoff_of(i, p.ld_res);
      const unsigned o0 = ro + (unsigned)(8 * g) * 2u, o1 = ro + (unsigned)(32 + 8 * g) * 2u, o2 = ro + (unsigned)(64 + 4 * g) * 2u;
      // (one statement, led by s_nop 4: the descriptor may just have been restored from an SGPR spill lane by v_readlane, and a
      //  VMEM instruction needs five wait states after a VALU write of an SGPR it reads -- hipcc's hazard recogniser does not
      //  look inside inline asm: this faulted with a garbage descriptor word in one build.  Early-clobber outputs: a load's
      //  data may return before the next one has read its address register.)
      asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %3, %6, 0 offen\n\tbuffer_load_dwordx4 %1, %4, %6, 0 offen\n\tbuffer_load_dwordx2 %2, %5, %6, 0 offen"
                   : "=&v"(r16[slot][0]), "=&v"(r16[slot][1]), "=&v"(r8[slot])
                   : "v"(o0), "v"(o1), "v"(o2), "s"(rs_res)
                   : "memory");
    };
    float rsum[TM], rsq[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) rsum[i] = rsq[i] = 0.f;
    if (p.res) res_load(0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int slot = i & 1;
      if (p.res) {
        // await the three loads of row tile i: behind them are the three loads of row tile i + 1 (not for the last one) and the
        // three stores of row tile i - 1 (not for the first one)
        if (i + 1 < TM) res_load(i + 1, slot ^ 1);
        if (i == 0 || i + 1 == TM) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        asm volatile("" : "+v"(r16[slot][0]), "+v"(r16[slot][1]), "+v"(r8[slot])::"memory");
      }
      const unsigned ro = (p.dbg != 1) ? rowoff_of(i, p.ldc) : OOB_V;
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {
        const int nq = pr < 2 ? 2 : 1;
        unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u >= nq) break;
          const int j = pr * 2 + u;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ((acc[i][j][r] + bi[j][r]) + rb[j][r]) * p.alpha;  // (same association as gemm_epilogue)
          acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
          w[2 * u] = pack2<DT>(v[0], v[1]);
          w[2 * u + 1] = pack2<DT>(v[2], v[3]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k >= 2 * nq) break;
          const unsigned rr = pr < 2 ? r16[slot][pr & 1][k] : r8[slot][k & 1];
          if (DT == CA_F16) {  // fp16 + fp16 is exact in fp32: the packed add rounds exactly like the fp32 path
            unsigned s_;
            asm("v_pk_add_f16 %0, %1, %2" : "=v"(s_) : "v"(w[k]), "v"(rr));
            w[k] = s_;
          } else {
            w[k] = pack2<DT>(Elem<DT>::to_f((u16)(w[k] & 0xffffu)) + Elem<DT>::to_f((u16)(rr & 0xffffu)),
                             Elem<DT>::to_f((u16)(w[k] >> 16)) + Elem<DT>::to_f((u16)(rr >> 16)));
          }
        }
        if (EPI == 2) {  // row sums: of the values as stored (rounded)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k >= 2 * nq) break;
            const float lo = Elem<DT>::to_f((u16)(w[k] & 0xffffu)), hi = Elem<DT>::to_f((u16)(w[k] >> 16));
            rsum[i] += lo + hi;
            rsq[i] = fmaf(lo, lo, fmaf(hi, hi, rsq[i]));
          }
        }
        if (pr < 2) __builtin_amdgcn_raw_buffer_store_b128((u32x4){w[0], w[1], w[2], w[3]}, rs_c, ro + (unsigned)(32 * pr + 8 * g) * 2u, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b64((u32x2){w[0], w[1]}, rs_c, ro + (unsigned)(64 + 4 * g) * 2u, 0, 0);
      }
    }
    if (EPI == 2) {
      // ca_gemm_args.row_sums_out: (sum, sum of squares) of the wave's 80 stored columns per row -- one partial sum per wave
      // quarter, [M][4 * N / 320][2]; the consumer finishes them with ca_ln_finish_sums.  Lanes l, l^16, l^32, l^48 hold pieces
      // of one row; written after the last residual load has been awaited (the counted waits above count no other stores).
      const int tn4 = (n0 / BN) * 4 + wc, parts = (p.n / BN) * 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float a = rsum[i], b = rsq[i];
        a += __shfl_xor(a, 16);
        b += __shfl_xor(b, 16);
        a += __shfl_xor(a, 32);
        b += __shfl_xor(b, 32);
        const int m = m0 + wr * 128 + i * 16 + l15;
        if (g == 0 && m < p.m) *reinterpret_cast<float2*>(p.row_sums + ((int64_t)m * parts + tn4) * 2) = make_float2(a, b);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (parameter reads retired: the set is re-filled two tiles later)
  };

  // ---- EPI = 1: folded LayerNorm ((mean, rstd) per row, finished by the caller: ca_ln_finish_sums), bias, alpha, optional GEGLU.
  // The statistics of the lane's eight rows are ordinary loads at the start of the epilogue (16 registers); hipcc awaits them
  // with vmcnt(0) -- loads and stores share the counter and it treats the mix as unordered -- which also retires the previous
  // tile's stores and the DMA unit in flight: once per tile, ~1 us.  (A version that read the producer's partial sums with
  // inline-asm loads pipelined against the stores needed 32 more registers: hipcc spilled the asm's destination registers
  // right behind the load instruction, i.e. before the data had arrived.)  Rows are processed top to bottom and the accumulators
  // zeroed AFTER the epilogue, so finished rows free their registers: 160 accumulators + column sums and bias of five MFMA
  // tiles do not fit otherwise, and a spill anywhere puts the K loop's state into scratch as well.
  auto epilogue_ln = [&](int seq, int m0, int n0) __attribute__((always_inline)) {
    const unsigned char* pset = smem_b + PAR_BASE + (seq & 1) * PSET;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int l15 = lane_o & 15, g = lane_o >> 4;
    float2 st[TM];
    if (p.ln_stats) {
      const __amdgpu_buffer_rsrc_t rs_st = __builtin_amdgcn_make_buffer_rsrc((void*)p.ln_stats, 0, (unsigned)p.m * 8u, 0x00020000);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wr * 128 + i * 16 + l15;
        const u32x2 v = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_st, m < p.m ? (unsigned)m * 8u : OOB_V, 0, 0));
        st[i] = make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
      }
    } else {
#pragma unroll
      for (int i = 0; i < TM; ++i) st[i] = make_float2(0.f, 1.f);  // 1 * (x - 0 * 0) = x: the fold is then an exact no-op
    }
    f32x4 cs[TN], bi[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int c0 = wc * 80 + (geglu ? (j < 4 ? 16 * g + 4 * j : 64 + 4 * g) : (j < 4 ? 32 * (j >> 1) + 8 * g + 4 * (j & 1) : 64 + 4 * g));
      cs[j] = *reinterpret_cast<const f32x4*>(pset + P_CS + c0 * 4);
      bi[j] = *reinterpret_cast<const f32x4*>(pset + P_BI + c0 * 4);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wr * 128 + i * 16 + l15;
      const unsigned ro = (m < p.m && p.dbg != 1) ? (unsigned)m * (unsigned)p.ldc * 2u + (unsigned)(geglu ? (n0 >> 1) + wc * 40 : n0 + wc * 80) * 2u : OOB_V;
      unsigned w[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = st[i].y * (acc[i][j][r] - st[i].x * cs[j][r]);
          x = (x + bi[j][r]) + 0.f;  // same association as gemm_epilogue (no row bias here)
          v[r] = x * p.alpha;
        }
        if (geglu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = Elem<DT>::to_f(Elem<DT>::from_f(v[r]));  // (the Linear's output is rounded first)
          const f32x2 gg = gelu_erf_f2((f32x2){v[1], v[3]});
          w[j] = pack2<DT>(v[0] * gg[0], v[2] * gg[1]);
        } else if (j < 4) {
          const unsigned lo = pack2<DT>(v[0], v[1]), hi = pack2<DT>(v[2], v[3]);
          if ((j & 1) == 0) {
            w[0] = lo;
            w[1] = hi;
          } else {
            __builtin_amdgcn_raw_buffer_store_b128((u32x4){w[0], w[1], lo, hi}, rs_c, ro + (unsigned)(32 * (j >> 1) + 8 * g) * 2u, 0, 0);
          }
        } else {
          __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack2<DT>(v[0], v[1]), pack2<DT>(v[2], v[3])}, rs_c, ro + (unsigned)(64 + 4 * g) * 2u, 0, 0);
        }
      }
      if (geglu) {
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){w[0], w[1], w[2], w[3]}, rs_c, ro + (unsigned)(8 * g) * 2u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(w[4], rs_c, ro + (unsigned)(32 + 2 * g) * 2u, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (parameter reads retired: the set is re-filled two tiles later)
  };

  // ---------------------------------------------------------------- run
  // Convolutions: group 1 issues K tile cv + 2 in front of its MFMAs of k half 1 instead of K tile cv + 1 behind its reads of
  // k half 0 (a unit then has three intervals to land instead of two; the gather's issue takes ~1300 cycles against ~500 of a
  // dense unit).  Measured in one process (tools/ps_check.py --time, us): convolutions 32x32 latents 640->640 258 vs 294,
  // 1280->640 468 vs 545, 1280->1280 945 vs 1102; dense 32768x640x2560 135 vs 117, 131072x320x1280 160 vs 146 -- so per MODE.
  const bool g1_late = MODE == 1 && wr == 1;
  CA_PQ_SET_TILE(0)
  issue_tile();  // K tile 0 -> buffer 0
  if (g1_late) advance_and_issue();
  {
    const unsigned f0 = flag_begin(0);
    flag_finish(0, 0, f0);
  }
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0

  int cv = 0;  // K tile of the whole stream being computed
  for (int seq = 0; seq < my_tiles; ++seq) {
    int tm, tn;
    tile_of(seq, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    for (int t = 0; t < (p.dbg == 2 ? 0 : nt); ++t) {
      const int par = cv & 1;
      const u16* buf = smem + par * BUF;
      stamp(1);
      // ---- reads of k half 0, then K tile cv + 1 into the other buffer (free: its last reads were group 1's of k half 1 of
      // K tile cv - 1, one interval before group 0 gets here)
      __builtin_amdgcn_sched_barrier(0);
      read_frags(buf, 0);
      if (!g1_late) advance_and_issue();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(2);
      if (!CA_PQ_ABL(8)) __builtin_amdgcn_s_barrier();
      stamp(3);
      mfma_all();
      stamp(4);
      if (!CA_PQ_ABL(8)) __builtin_amdgcn_s_barrier();
      stamp(5);
      // ---- reads of k half 1; group 1 confirms its pieces of K tile cv + 1 behind them (group 0 reads that tile in the
      // interval after group 1's next barrier)
      __builtin_amdgcn_sched_barrier(0);
      unsigned fl = 0;
      if (wr == 1) fl = flag_begin(par ^ 1);
      read_frags(buf, 1);
      if (wr == 1 && cv + 1 < total_kt) flag_finish(par ^ 1, cv + 1, fl);
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(6);
      if (!CA_PQ_ABL(8)) __builtin_amdgcn_s_barrier();
      stamp(7);
      // ---- MFMAs of k half 1; group 0 confirms its pieces of K tile cv + 1 behind them
      if (g1_late) advance_and_issue();
      if (wr == 0) fl = flag_begin(par ^ 1);
      mfma_all();
      if (wr == 0 && cv + 1 < total_kt) flag_finish(par ^ 1, cv + 1, fl);
      stamp(8);
      if (!CA_PQ_ABL(8)) __builtin_amdgcn_s_barrier();
      ++cv;
    }
    // both groups' epilogues in the same interval (ca_gemm_ps.h): one extra barrier for group 0 before, for group 1 after
    if (wr == 0) __builtin_amdgcn_s_barrier();
    stamp(9);
    if (EPI == 1) epilogue_ln(seq, m0, n0);
    else epilogue(seq, m0, n0);  // (EPI = 0 / 2)
    stamp(10);
    if (wr == 1) __builtin_amdgcn_s_barrier();
  }
#undef CA_PQ_SET_TILE
  if (wr == 0) __builtin_amdgcn_s_barrier();
}

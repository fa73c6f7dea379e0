// Persistent ping-pong GEMM / implicit-GEMM 3x3 convolution, 128 x 320 block tile, software-pipelined epilogue.
//
// EXPERIMENT (CA_GEMM_PP=4 only; never chosen by the heuristic).  Two findings keep it out of the default path
// (DESIGN.md section 3): (1) the drain's VALU work sits inside a phase of the lock-stepped wave groups and costs more
// than the exposed epilogue it replaces; (2) its counted waits treat the drain's stores as retiring in issue order with
// the LDS-DMA loads -- the weight-resident kernel's experiments showed that stores retire OUT of order with respect to
// loads (a vmcnt that leaves "the youngest N" outstanding may leave an OLDER load outstanding once stores are among
// them), so a wait here can be satisfied before the operand unit it guards has landed.  The parity runs passed, but
// that is timing luck, not a guarantee.
//
// Main loop = ca_gemm_pp2.h (8 waves = 2 groups x 4, per-wave 64 x 80, K tile = DMA units A | B0 | B1, two phases
// per K tile, the groups one barrier apart, reads retired before the phase's first barrier).  What is new:
//   * PERSISTENT: one block per CU walks tiles b, b+G, b+2G, ...; the operand stream (LDS-DMA) runs continuously
//     across tile boundaries -- the first units of the next tile are issued during the last K tiles of the current
//     one, so a tile has no prologue bubble.
//   * PIPELINED EPILOGUE: at the end of a tile the accumulators are turned into the 40 packed registers
//     `pend` (= the value the staged epilogue of ca_gemm_core.h rounds to the activation type: (acc [LayerNorm
//     fold] + bias + rowbias) * alpha) and the next tile starts at once.  During K tiles 0..3 of the next tile each
//     wave drains one 16-row slice per K tile in its load segments: residual prefetch (phase 1), transpose through a
//     wave-private LDS patch, residual add / post scale / activation / GEGLU, 16-byte stores (phase 2).  The
//     output burst of a tile (HBM writes run at ~2.3 TB/s when all CUs store at once: 35..45% of a K = 1280 GEMM
//     with the unpipelined kernel) is spread under the MFMA work of the next tile.
//   * per-tile epilogue parameters (colsum, bias, up to two rowbias groups, LayerNorm row statistics) travel as
//     extra LDS-DMA pieces of the operand stream into a double-buffered parameter block: no VGPR-destination
//     global load in the loop except the residual prefetch, which is inline asm with its own counted wait.
//   * vmcnt bookkeeping is DYNAMIC: every wave counts the VMEM instructions it has issued (`issued`, wave-uniform)
//     and remembers the count at the issue of each unit; a wait for a unit is s_waitcnt vmcnt(issued - mark), chosen
//     from a table of immediates.  Exact for any mix of DMA / residual loads / stores; no hand-derived constants.
// Requirements (else the caller keeps k_gemm_dma): N % 320 == 0, K tiles >= 5, fp16/bf16 output, rowbias groups of a
// multiple of 64 rows, operands addressable with 32-bit byte offsets.

__device__ __forceinline__ void ca_vm_wait(int n) {
  // Waits until at most n VMEM operations of this wave are outstanding: EXACT for 0 <= n <= 23 (larger: 23, a safe
  // over-wait).  The steady-state value is tested first; the rest is a balanced decision tree of scalar compares written
  // as ONE asm statement (5 compares deep): as C++ `if` chains or a `switch` hipcc structurises the (wave-uniform!) control
  // flow into ~40 SALU instructions per call, and a table that rounds n DOWN makes the wave wait for pieces issued only one
  // phase earlier, which stalls it for the rest of their latency (both measured with s_memtime stamps).
  if (n == 7) {
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    return;
  }
  n = __builtin_amdgcn_readfirstlane(n > 23 ? 23 : (n < 0 ? 0 : n));  // (an SGPR for the asm operand)
  asm volatile(
      "s_cmp_ge_i32 %0, 12\n\t"
      "s_cbranch_scc1 30f\n\t"
      "s_cmp_ge_i32 %0, 6\n\t"
      "s_cbranch_scc1 31f\n\t"
      "s_cmp_ge_i32 %0, 3\n\t"
      "s_cbranch_scc1 32f\n\t"
      "s_cmp_ge_i32 %0, 1\n\t"
      "s_cbranch_scc1 33f\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_branch 99f\n\t"
      "33:\n"
      "s_cmp_ge_i32 %0, 2\n\t"
      "s_cbranch_scc1 34f\n\t"
      "s_waitcnt vmcnt(1)\n\t"
      "s_branch 99f\n\t"
      "34:\n"
      "s_waitcnt vmcnt(2)\n\t"
      "s_branch 99f\n\t"
      "32:\n"
      "s_cmp_ge_i32 %0, 4\n\t"
      "s_cbranch_scc1 35f\n\t"
      "s_waitcnt vmcnt(3)\n\t"
      "s_branch 99f\n\t"
      "35:\n"
      "s_cmp_ge_i32 %0, 5\n\t"
      "s_cbranch_scc1 36f\n\t"
      "s_waitcnt vmcnt(4)\n\t"
      "s_branch 99f\n\t"
      "36:\n"
      "s_waitcnt vmcnt(5)\n\t"
      "s_branch 99f\n\t"
      "31:\n"
      "s_cmp_ge_i32 %0, 9\n\t"
      "s_cbranch_scc1 37f\n\t"
      "s_cmp_ge_i32 %0, 7\n\t"
      "s_cbranch_scc1 38f\n\t"
      "s_waitcnt vmcnt(6)\n\t"
      "s_branch 99f\n\t"
      "38:\n"
      "s_cmp_ge_i32 %0, 8\n\t"
      "s_cbranch_scc1 39f\n\t"
      "s_waitcnt vmcnt(7)\n\t"
      "s_branch 99f\n\t"
      "39:\n"
      "s_waitcnt vmcnt(8)\n\t"
      "s_branch 99f\n\t"
      "37:\n"
      "s_cmp_ge_i32 %0, 10\n\t"
      "s_cbranch_scc1 40f\n\t"
      "s_waitcnt vmcnt(9)\n\t"
      "s_branch 99f\n\t"
      "40:\n"
      "s_cmp_ge_i32 %0, 11\n\t"
      "s_cbranch_scc1 41f\n\t"
      "s_waitcnt vmcnt(10)\n\t"
      "s_branch 99f\n\t"
      "41:\n"
      "s_waitcnt vmcnt(11)\n\t"
      "s_branch 99f\n\t"
      "30:\n"
      "s_cmp_ge_i32 %0, 18\n\t"
      "s_cbranch_scc1 42f\n\t"
      "s_cmp_ge_i32 %0, 15\n\t"
      "s_cbranch_scc1 43f\n\t"
      "s_cmp_ge_i32 %0, 13\n\t"
      "s_cbranch_scc1 44f\n\t"
      "s_waitcnt vmcnt(12)\n\t"
      "s_branch 99f\n\t"
      "44:\n"
      "s_cmp_ge_i32 %0, 14\n\t"
      "s_cbranch_scc1 45f\n\t"
      "s_waitcnt vmcnt(13)\n\t"
      "s_branch 99f\n\t"
      "45:\n"
      "s_waitcnt vmcnt(14)\n\t"
      "s_branch 99f\n\t"
      "43:\n"
      "s_cmp_ge_i32 %0, 16\n\t"
      "s_cbranch_scc1 46f\n\t"
      "s_waitcnt vmcnt(15)\n\t"
      "s_branch 99f\n\t"
      "46:\n"
      "s_cmp_ge_i32 %0, 17\n\t"
      "s_cbranch_scc1 47f\n\t"
      "s_waitcnt vmcnt(16)\n\t"
      "s_branch 99f\n\t"
      "47:\n"
      "s_waitcnt vmcnt(17)\n\t"
      "s_branch 99f\n\t"
      "42:\n"
      "s_cmp_ge_i32 %0, 21\n\t"
      "s_cbranch_scc1 48f\n\t"
      "s_cmp_ge_i32 %0, 19\n\t"
      "s_cbranch_scc1 49f\n\t"
      "s_waitcnt vmcnt(18)\n\t"
      "s_branch 99f\n\t"
      "49:\n"
      "s_cmp_ge_i32 %0, 20\n\t"
      "s_cbranch_scc1 50f\n\t"
      "s_waitcnt vmcnt(19)\n\t"
      "s_branch 99f\n\t"
      "50:\n"
      "s_waitcnt vmcnt(20)\n\t"
      "s_branch 99f\n\t"
      "48:\n"
      "s_cmp_ge_i32 %0, 22\n\t"
      "s_cbranch_scc1 51f\n\t"
      "s_waitcnt vmcnt(21)\n\t"
      "s_branch 99f\n\t"
      "51:\n"
      "s_cmp_ge_i32 %0, 23\n\t"
      "s_cbranch_scc1 52f\n\t"
      "s_waitcnt vmcnt(22)\n\t"
      "s_branch 99f\n\t"
      "52:\n"
      "s_waitcnt vmcnt(23)\n\t"
      "s_branch 99f\n\t"
      "99:\n"
      :
      : "s"(n)
      : "scc", "memory");
}

template <int DT, int MODE>
__global__ __launch_bounds__(512, 2) void k_gemm_pp3(GemmKParams p, int tiles_total, unsigned c_bytes, unsigned res_bytes) {
  constexpr int BM = 128, BN = 320, KT = 64;
  constexpr int TM = 4, TN = 5;
  constexpr int A_ROWS = 128, B0_ROWS = 128, B1_ROWS = 192;
  constexpr int OFF_A = 0, OFF_B0 = A_ROWS * KT, OFF_B1 = (A_ROWS + B0_ROWS) * KT;
  constexpr int BUF = (A_ROWS + B0_ROWS + B1_ROWS) * KT;  // elements of one K tile (56 KB)
  // byte layout of the single LDS array
  constexpr int PAR_BASE = 2 * BUF * 2;
  constexpr int P_CS = 0, P_BI = 2048, P_RB0 = 4096, P_RB1 = 6144, P_ST = 8192, PSET = 9216;
  constexpr int STG_BASE = PAR_BASE + 2 * PSET;
  constexpr int STG_WAVE = 3072;  // per wave: the residual of one 16-row slice as 3 LDS-DMA pieces of 1 KB
  constexpr int SMEM_BYTES = STG_BASE + 8 * STG_WAVE;
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
  // XCD-aware: block b runs on XCD b % 8; give every XCD a contiguous 1/8 of each round of G tiles
  const int bslot = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, p.a2 ? p.a2_bytes : p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ln_colsum ? (const void*)p.ln_colsum : (const void*)p.w), 0, p.ln_colsum ? (unsigned)p.n * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)p.w), 0, p.bias ? (unsigned)p.n * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_st = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ln_stats ? (const void*)p.ln_stats : (const void*)p.w), 0, p.ln_stats ? (unsigned)p.m * 8u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? (const void*)p.res : (const void*)p.c), 0, p.res ? res_bytes : 0u, 0x00020000);
  const unsigned rb_groups = p.rowbias ? (unsigned)((p.m + p.rows_per_group - 1) / p.rows_per_group) : 0u;
  const __amdgpu_buffer_rsrc_t rs_rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.rowbias ? (const void*)p.rowbias : (const void*)p.w), 0,
                                                                         p.rowbias ? (unsigned)(((int64_t)(rb_groups - 1) * p.ld_rowbias + p.n) * 4) : 0u, 0x00020000);

  // timing experiment (CA_PP_DBG=9): block 0, waves 0 and 4 (one of each group) stamp the shader clock into p.partial
  unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(p.partial);
  int stamp_i = 0;
  auto stamp = [&](int tag) __attribute__((always_inline)) {
#ifndef CA_PP3_STAMPS
    (void)tag;
    return;
#endif
    if (p.dbg == 9 && blockIdx.x == 0 && (wid == 0 || wid == 4) && lane == 0 && stamp_i < 1000) {
      stamps[(wid >> 2) * 2048 + 2 * stamp_i] = __builtin_readcyclecounter();
      stamps[(wid >> 2) * 2048 + 2 * stamp_i + 1] = (unsigned long long)tag;
      ++stamp_i;
    }
  };
  auto swz = [](int row) { return (row >> 1) & 7; };
  const int r8 = lane >> 3, cp = lane & 7;
  const int kc = p.c1 + p.c2;
  const int kct = kc / KT;
  const unsigned wld = (unsigned)(p.taps * kc);
  const int nt = p.taps * kct;  // K tiles per output tile (>= 5)

  // ---------------------------------------------------------------- DMA side (runs ~2 K tiles ahead)
  // Per-lane source offsets are kept per DMA piece as a VGPR "voffset" that already contains the lane's row and its
  // swizzled 16-byte chunk; the position along K is a wave-uniform SGPR "soffset" of the instruction, so a steady
  // K tile issues its 7 pieces with no VALU work at all.  Rows past M (and conv halo taps) get OOB_V: far beyond any
  // descriptor we accept (< 2 GB) whether or not the hardware adds soffset before the range check -> zeros.
  constexpr unsigned OOB_V = 0x80000000u;
  const int ab_chunk0 = cp ^ swz(wid * 16 + r8);   // A and B0 pieces stage rows wid*16 + 8i + r8: chunk_i = chunk_0 ^ 4i

  unsigned b0_v[2] = {0, 0}, b1_v[2] = {0, 0};     // B1 piece 2 = piece 0 + 16 rows (same chunk): soffset
  unsigned a1_v[2] = {0, 0}, a2_v[2] = {0, 0};     // dense A, source 1 / 2
  int a_img[2] = {0, 0}, a_ho[2] = {0, 0}, a_wo[2] = {0, 0};  // conv A
  bool a_ok[2] = {false, false};
  int d_seq = 0, d_t = 0;       // tile sequence number / K tile of the stream head
  int d_tap = 0, d_c0 = 0;      // the same position as (tap, first channel)
  bool d_live = false;
  int issued = 0;                 // VMEM instructions issued by this wave so far
  int mark_ab0 = 0, mark_ab1 = 0;  // `issued` right after the A/B0 unit of buffer parity 0 / 1
  int mark_b10 = 0, mark_b11 = 0;

  auto tile_of = [&](int seq, int& tm, int& tn) __attribute__((always_inline)) -> bool {
    const int id = seq * G + bslot;
    if (id >= tiles_total) return false;
    tile_coords((unsigned)(p.dbg == 6 ? bslot : id), tiles_m, tiles_n, tm, tn);  // (dbg 6: every tile of a block is its first: timing experiment)
    return true;
  };

  auto dma_set_tile = [&](int seq, int m0, int n0) __attribute__((always_inline)) {
    // everything below is recomputed from the lane id on purpose: hipcc hoists lane-dependent invariants out of the
    // tile loop and then SPILLS them; a scratch reload is followed by s_waitcnt vmcnt(0), i.e. a full drain of the
    // DMA queue -- a dozen of those cost 11..16 us per tile.  The empty asm makes the lane id opaque here.
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int r8 = lane_o >> 3, cp = lane_o & 7;
    const int ab_chunk0 = cp ^ swz(wid * 16 + r8), b1_chunk0 = cp ^ swz(wid * 24 + r8);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = wid * 16 + i * 8 + r8;
      const int m = m0 + r;
      a_ok[i] = m < p.m;
      const unsigned ch = (unsigned)((ab_chunk0 ^ (4 * i)) * 16);
      if (MODE == 1) {
        const int mm = a_ok[i] ? m : p.m - 1;
        const int hw = p.hout * p.wout;
        a_img[i] = mm / hw;
        const int rem = mm - a_img[i] * hw;
        a_ho[i] = rem / p.wout;
        a_wo[i] = rem - a_ho[i] * p.wout;
      } else {
        a1_v[i] = a_ok[i] ? (unsigned)m * (unsigned)p.lda * 2u + ch : OOB_V;   // (< 2 GB: checked by the launcher)
        a2_v[i] = a_ok[i] ? (unsigned)m * (unsigned)p.lda2 * 2u + ch : OOB_V;
      }
      b0_v[i] = (unsigned)(n0 + (r >> 5) * 80 + (r & 31)) * wld * 2u + ch;
      const int r1 = wid * 24 + i * 8 + r8;
      b1_v[i] = (unsigned)(n0 + (r1 / 48) * 80 + 32 + r1 % 48) * wld * 2u + (unsigned)((b1_chunk0 ^ (4 * i)) * 16);
    }
    d_tap = 0;
    d_c0 = 0;
    // epilogue parameters of this tile -> parameter set (seq & 1).  Pieces of 1 KB (64 lanes x 16 B):
    //   wave 0: colsum[0:256), colsum[256:512)   wave 1: bias   wave 2: rowbias group 0   wave 3: rowbias group 1
    //   wave 4: LayerNorm statistics of the 128 rows
    unsigned char* pset = smem_b + PAR_BASE + (seq & 1) * PSET;
    // (explicit branches: a `cond ? rs_x : rs_y` over captured descriptors becomes a dynamic index into the closure,
    //  which then lives in scratch together with everything it references)
#define CA_PP3_PAR2(RS, OFFS, BASE)                                                                                              \
  {                                                                                                                              \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(pset + (OFFS)), 16, (BASE) + lane_o * 16u, 0, 0, 0);          \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(pset + (OFFS) + 1024), 16, (BASE) + 1024u + lane_o * 16u, 0, 0, 0); \
    issued += 2;                                                                                                                 \
  }
    if (wid == 0) {
      if (p.ln_colsum) CA_PP3_PAR2(rs_cs, P_CS, (unsigned)n0 * 4u)
    } else if (wid == 1) {
      if (p.bias) CA_PP3_PAR2(rs_bi, P_BI, (unsigned)n0 * 4u)
    } else if (wid == 2) {
      if (p.rowbias) CA_PP3_PAR2(rs_rb, P_RB0, (unsigned)n0 * 4u + (unsigned)(m0 / p.rows_per_group) * (unsigned)p.ld_rowbias * 4u)
    } else if (wid == 3) {
      if (p.rowbias && m0 / p.rows_per_group + 1 < (int)rb_groups)
        CA_PP3_PAR2(rs_rb, P_RB1, (unsigned)n0 * 4u + (unsigned)(m0 / p.rows_per_group + 1) * (unsigned)p.ld_rowbias * 4u)
#undef CA_PP3_PAR2
    } else if (wid == 4 && p.ln_stats) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_st, (__attribute__((address_space(3))) void*)(pset + P_ST), 16, (unsigned)m0 * 8u + lane_o * 16u, 0, 0, 0);
      issued += 1;
    }
  };

  auto issue_ab0 = [&](int par) __attribute__((always_inline)) {  // A and B0 of the K tile at the stream head into buffer `par`
    if (!d_live) return;
    u16* buf = smem + par * BUF;
    const unsigned wk = (unsigned)d_t * (KT * 2u);  // weights: K is contiguous over (tap, channel)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_B0 + (wid * 2 + i) * 8 * KT), 16, b0_v[i], wk, 0, 0);
    const bool src2 = d_c0 >= p.c1;  // c1 % 64 == 0: a K tile never straddles the two sources
    if (MODE == 1) {
      const int cs = src2 ? p.c2 : p.c1;
      const int cbase = src2 ? d_c0 - p.c1 : d_c0;
      const int kh = d_tap / 3, kw = d_tap - kh * 3;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int hi = a_ho[i] * p.stride + kh - p.pad_lo;
        const int wi = a_wo[i] * p.stride + kw - p.pad_lo;
        const bool ok = a_ok[i] && hi >= 0 && wi >= 0 && hi < (p.hin << p.ups) && wi < (p.win << p.ups);
        const int pix = (a_img[i] * p.hin + (hi >> p.ups)) * p.win + (wi >> p.ups);
        const unsigned off = ok ? ((unsigned)pix * (unsigned)cs + (unsigned)(cbase + (ab_chunk0 ^ (4 * i)) * 8)) * 2u : OOB_V;
        void* d = buf + OFF_A + (wid * 2 + i) * 8 * KT;
        if (src2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (__attribute__((address_space(3))) void*)d, 16, off, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)d, 16, off, 0, 0, 0);
      }
    } else {
      const unsigned ak = (unsigned)(src2 ? d_c0 - p.c1 : d_c0) * 2u;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        void* d = buf + OFF_A + (wid * 2 + i) * 8 * KT;
        if (src2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (__attribute__((address_space(3))) void*)d, 16, a2_v[i], ak, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)d, 16, a1_v[i], ak, 0, 0);
      }
    }
    issued += 4;
  };
  auto issue_b1 = [&](int par) __attribute__((always_inline)) {  // B1 of the same K tile as the last issue_ab0, then the stream advances
    if (!d_live) return;
    u16* buf = smem + par * BUF;
    const unsigned wk = (unsigned)d_t * (KT * 2u);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_B1 + (wid * 3 + 0) * 8 * KT), 16, b1_v[0], wk, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_B1 + (wid * 3 + 1) * 8 * KT), 16, b1_v[1], wk, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_B1 + (wid * 3 + 2) * 8 * KT), 16, b1_v[0], wk + 16u * wld * 2u, 0, 0);
    issued += 3;
    ++d_t;
    d_c0 += KT;
    if (d_c0 == kc) {
      d_c0 = 0;
      ++d_tap;
    }
  };

  // ---------------------------------------------------------------- compute side
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned pend[TM][TN][2];  // packed staged values of the previous tile
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) pend[i][j][0] = pend[i][j][1] = 0u;
  bool have_pend = false;
  int pend_m0 = 0, pend_n0 = 0;

  // fragment addresses: tile i of an operand sits i * 16 rows further (same swizzle: 16 rows = 8 swizzle periods);
  // the second k half flips chunk bit 2 = element offset bit 5
  int fa_base[2], fb0_base[2], fb1_base[2];
  {
    const int ra = wr * 64 + l15, rb0 = wc * 32 + l15, rb1 = wc * 48 + l15;
    fa_base[0] = OFF_A + ra * KT + ((g ^ swz(ra)) << 3);
    fb0_base[0] = OFF_B0 + rb0 * KT + ((g ^ swz(rb0)) << 3);
    fb1_base[0] = OFF_B1 + rb1 * KT + ((g ^ swz(rb1)) << 3);
    fa_base[1] = fa_base[0] ^ 32;
    fb0_base[1] = fb0_base[0] ^ 32;
    fb1_base[1] = fb1_base[0] ^ 32;
  }
  u32x4 fa[4][2], fb0[2][2], fb1[3][2];

  // drain geometry: a 16-row slice of the wave's 64 x 80 patch = 16 rows x 10 chunks of 8 columns; lane handles row
  // lane / 4 and chunks (lane & 3) + 4u, u = 0..2 (the last one only for (lane & 3) < 2)
  unsigned char* const stg = smem_b + STG_BASE + wid * STG_WAVE;
  int mark_res = 0;
  // residual of slice d of the pending tile -> this wave's LDS patch, by LDS-DMA (no VGPR destination: the compiler
  // neither sees nor waits for these loads).  Piece u, lane l = chunk (l & 3) + 4u of row l >> 2: the store mapping.
  auto drain_prefetch = [&](int d) __attribute__((always_inline)) {
    if (!p.res) return;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));  // (no hoisting of the address arithmetic: see dma_set_tile)
    const int dr_row = lane_o >> 2, dr_c0 = lane_o & 3;
    const int m = pend_m0 + wr * 64 + d * 16 + dr_row;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const bool ok = (u < 2 || dr_c0 < 2) && m < p.m;
      const unsigned off = ok ? (unsigned)m * (unsigned)p.ld_res * 2u + (unsigned)(pend_n0 + wc * 80 + (dr_c0 + 4 * u) * 8) * 2u : OOB_V;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_res, (__attribute__((address_space(3))) void*)(stg + u * 1024), 16, off, 0, 0, 0);
    }
    issued += 3;
    mark_res = issued;
  };
  auto drain_store = [&](auto dc) __attribute__((always_inline)) {
    // slice d (16 rows of this wave's patch): the 5 packed fragments go to global IN FRAGMENT LAYOUT -- a lane holds 4
    // consecutive columns of one row, one 8-byte store per fragment (16 rows x 32 B per instruction; the four
    // instructions that complete a 128-byte line follow each other, the L2 merges them).  No LDS transposition: with
    // it a slice cost ~2900 cycles of LDS round trips in the load segment while the partner group idled at the barrier
    // (measured with s_memtime stamps: 11 us per tile).  The residual comes from the wave's LDS patch (LDS-DMA a slice
    // ahead), read back in the same fragment layout.
    constexpr int d = decltype(dc)::value;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int l15 = lane_o & 15, g = lane_o >> 4;
    const int m = pend_m0 + wr * 64 + d * 16 + l15;
    const bool ok = m < p.m && p.dbg != 1;
    if (p.res) ca_vm_wait(issued - mark_res);  // this wave's residual pieces have landed (wave-private patch: no barrier)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      __builtin_amdgcn_sched_barrier(0);  // one fragment at a time: keeps the temporaries of this block small
      u32x2 rvj = {0u, 0u};
      if (p.res) {
        const int ch = 2 * j + (g >> 1);  // 8-column chunk of the row; piece = ch / 4, lane of the piece = row * 4 + ch % 4
        rvj = *reinterpret_cast<const u32x2*>(stg + (ch >> 2) * 1024 + (l15 * 4 + (ch & 3)) * 16 + (g & 1) * 8);
      }
      // (the empty asm keeps hipcc from hoisting the fp16 -> fp32 unpacking of all 80 pending values out of the K loop:
      //  loop-invariant code motion did exactly that and the 80 extra live registers spilled)
      unsigned p0 = pend[d][j][0], p1 = pend[d][j][1];
      asm volatile("" : "+v"(p0), "+v"(p1));
      float v[4];
      v[0] = Elem<DT>::to_f((u16)(p0 & 0xffffu));
      v[1] = Elem<DT>::to_f((u16)(p0 >> 16));
      v[2] = Elem<DT>::to_f((u16)(p1 & 0xffffu));
      v[3] = Elem<DT>::to_f((u16)(p1 >> 16));
      if (p.res) {
        v[0] += Elem<DT>::to_f((u16)(rvj[0] & 0xffffu));
        v[1] += Elem<DT>::to_f((u16)(rvj[0] >> 16));
        v[2] += Elem<DT>::to_f((u16)(rvj[1] & 0xffffu));
        v[3] += Elem<DT>::to_f((u16)(rvj[1] >> 16));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] *= p.post;
      if (p.act != CA_ACT_NONE) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = act_f(v[k], p.act);
      }
      const int ncol = pend_n0 + wc * 80 + j * 16 + g * 4;
      if (p.dbg == 7 || (p.dbg == 8 && j > 0)) continue;  // (experiment knobs: no store instruction / one per slice)
      if (p.geglu) {
        const unsigned w = pack2<DT>(v[0] * gelu_erf_f(v[1]), v[2] * gelu_erf_f(v[3]));
        const unsigned off = ok ? (unsigned)m * (unsigned)p.ldc * 2u + (unsigned)(ncol >> 1) * 2u : OOB_V;
        __builtin_amdgcn_raw_buffer_store_b32(w, rs_c, off, 0, 0);
      } else {
        u32x2 w;
        w[0] = pack2<DT>(v[0], v[1]);
        w[1] = pack2<DT>(v[2], v[3]);
        const unsigned off = ok ? (unsigned)m * (unsigned)p.ldc * 2u + (unsigned)ncol * 2u : OOB_V;
        __builtin_amdgcn_raw_buffer_store_b64(w, rs_c, off, 0, 0);
      }
    }
    issued += TN;
    if (p.res) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the patch is re-filled by the next prefetch)
  };

  auto mfma_p1 = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Elem<DT>::mfma(fb0[j][s], fa[i][s], acc[i][j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfma_p2 = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][2 + j] = Elem<DT>::mfma(fb1[j][s], fa[i][s], acc[i][2 + j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // stream helpers (macros, not lambdas: a lambda that captures other lambdas keeps the closures -- and with them
  // the kernel arguments -- in scratch memory, which also makes every descriptor "divergent")
#define CA_PP3_SET_TILE(SEQ)                                   \
  {                                                            \
    int tm_, tn_;                                              \
    d_live = tile_of((SEQ), tm_, tn_);                         \
    if (d_live) dma_set_tile((SEQ), tm_ * BM, tn_ * BN);       \
  }
  // one K tile (index t of the current tile).  The four 16-row slices of the previous tile are drained at K tiles
  // 0, nt/4, nt/2, 3nt/4 -- spread over the whole tile: when every CU stored its tile within the first four K tiles
  // the HBM write burst (20 MB per round against ~2.3 TB/s) stalled the operand stream as long as an unpipelined
  // epilogue would have -- at the start of phase 1, where no fragment register is live (the lowest VGPR pressure of
  // the K tile), followed by the residual prefetch of the next slice; while K tile nt-2 is computed the stream enters the next tile (its parameter pieces and A/B0 of its
  // K tile 0 are issued there).  All waits are counted dynamically (ca_vm_wait).
#define CA_PP3_KTILE()                                                                               \
  {                                                                                                  \
    const int par = cv & 1;                                                                          \
    const u16* buf = smem + par * BUF;                                                               \
    stamp(1);                                                                                        \
    if (have_pend && t == dnext_t) { /* slice `dnext` of the previous tile */                        \
      if (dnext == 0) {                                                                              \
        drain_store(std::integral_constant<int, 0>{});                                               \
        drain_prefetch(1);                                                                           \
      } else if (dnext == 1) {                                                                       \
        drain_store(std::integral_constant<int, 1>{});                                               \
        drain_prefetch(2);                                                                           \
      } else if (dnext == 2) {                                                                       \
        drain_store(std::integral_constant<int, 2>{});                                               \
        drain_prefetch(3);                                                                           \
      } else {                                                                                       \
        drain_store(std::integral_constant<int, 3>{});                                               \
      }                                                                                              \
      ++dnext;                                                                                       \
      dnext_t = dnext < 4 ? ((dnext * nt) >> 2) + dstagger : -1;                                     \
      stamp(2);                                                                                      \
    }                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                  \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) fa[i][s] = ld16(buf + fa_base[s] + i * 16 * KT);  \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) fb0[j][s] = ld16(buf + fb0_base[s] + j * 16 * KT); \
    }                                                                                                \
    issue_b1(par ^ 1);                                                                               \
    if (par) mark_b10 = issued; else mark_b11 = issued;                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
    stamp(3);                                                                                        \
    ca_vm_wait(issued - (par ? mark_b11 : mark_b10)); /* B1 of this K tile (read in phase 2) */       \
    stamp(4);                                                                                        \
    __builtin_amdgcn_s_barrier();                                                                    \
    stamp(5);                                                                                        \
    mfma_p1();                                                                                       \
    stamp(6);                                                                                        \
    __builtin_amdgcn_s_barrier();                                                                    \
    stamp(7);                                                                                        \
    _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                    \
      _Pragma("unroll") for (int j = 0; j < 3; ++j) fb1[j][s] = ld16(buf + fb1_base[s] + j * 16 * KT); \
    if (d_t == nt) { /* the stream enters the next tile */                                           \
      d_t = 0;                                                                                       \
      ++d_seq;                                                                                       \
      CA_PP3_SET_TILE(d_seq)                                                                         \
    }                                                                                                \
    issue_ab0(par);                                                                                  \
    if (par) mark_ab1 = issued; else mark_ab0 = issued;                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
    stamp(8);                                                                                        \
    ca_vm_wait(issued - (par ? mark_ab0 : mark_ab1)); /* A/B0 of the next K tile (read next phase) */ \
    stamp(9);                                                                                        \
    __builtin_amdgcn_s_barrier();                                                                    \
    stamp(10);                                                                                       \
    mfma_p2();                                                                                       \
    stamp(11);                                                                                       \
    __builtin_amdgcn_s_barrier();                                                                    \
    stamp(12);                                                                                       \
    ++cv;                                                                                            \
  }

  // accumulators -> pend (the staged value of gemm_epilogue), parameters from the LDS parameter set of this tile
  auto convert = [&](int seq, int m0) __attribute__((always_inline)) {
    const unsigned char* pset = smem_b + PAR_BASE + (seq & 1) * PSET;
    float2 st[TM];
    if (p.ln_stats) {
#pragma unroll
      for (int i = 0; i < TM; ++i) st[i] = *reinterpret_cast<const float2*>(pset + P_ST + (wr * 64 + i * 16 + l15) * 8);
    }
    int rb_sel = 0;
    if (p.rowbias) rb_sel = (m0 + wr * 64) / p.rows_per_group - m0 / p.rows_per_group;  // 0 or 1 (rows_per_group % 64 == 0)
    const unsigned char* rbp = pset + (rb_sel ? P_RB1 : P_RB0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = wc * 80 + j * 16 + g * 4;
      f32x4 cs = {0.f, 0.f, 0.f, 0.f}, bi = {0.f, 0.f, 0.f, 0.f}, rb = {0.f, 0.f, 0.f, 0.f};
      if (p.ln_stats) cs = *reinterpret_cast<const f32x4*>(pset + P_CS + col * 4);
      if (p.bias) bi = *reinterpret_cast<const f32x4*>(pset + P_BI + col * 4);
      if (p.rowbias) rb = *reinterpret_cast<const f32x4*>(rbp + col * 4);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
        if (p.ln_stats) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = st[i].y * (v[r] - st[i].x * cs[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (v[r] + bi[r]) + rb[r];  // same association as gemm_epilogue
        pend[i][j][0] = pack2<DT>(v[0] * p.alpha, v[1] * p.alpha);
        pend[i][j][1] = pack2<DT>(v[2] * p.alpha, v[3] * p.alpha);
        acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  // ---------------------------------------------------------------- run
  int tm, tn;
  if (!tile_of(0, tm, tn)) return;  // (never: the grid is min(tiles, CUs))
  CA_PP3_SET_TILE(0)
  issue_ab0(0);
  mark_ab0 = issued;
  issue_b1(0);
  mark_b10 = issued;
  issue_ab0(1);
  mark_ab1 = issued;
  ca_vm_wait(issued - mark_b10);  // K tile 0 complete (A/B0 of K tile 1 may be in flight)
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0

  // blocks drain their slices at different K tiles (offset 0 .. nt/4-1 by block): the stores of the whole chip are spread
  // in time instead of arriving as one burst per quarter tile
  const int dstagger = p.dbg == 5 ? 0 : bslot % (nt / 4 > 0 ? nt / 4 : 1);
  int cv = 0;
  for (int seq = 0;; ++seq) {
    if (!tile_of(seq, tm, tn)) break;
    const int m0 = tm * BM, n0 = tn * BN;
    int dnext = 0, dnext_t = dstagger;
    for (int t = 0; t < nt; ++t) CA_PP3_KTILE()
    stamp(13);
    if (p.dbg != 4) convert(seq, m0);
    stamp(14);
    have_pend = p.dbg != 3;
    pend_m0 = m0;
    pend_n0 = n0;
    if (have_pend) drain_prefetch(0);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  // drain the last tile
  if (have_pend) {
    drain_store(std::integral_constant<int, 0>{});
    drain_prefetch(1);
    drain_store(std::integral_constant<int, 1>{});
    drain_prefetch(2);
    drain_store(std::integral_constant<int, 2>{});
    drain_prefetch(3);
    drain_store(std::integral_constant<int, 3>{});
  }
#undef CA_PP3_KTILE
#undef CA_PP3_SET_TILE
}

// Persistent streaming GEMM / implicit-GEMM 3x3 convolution, 128 x 320 block tile (round 3).
//
// Main loop = ca_gemm_pp2.h: 8 waves = 2 groups (wr: 64-row halves) x 4 (wc: 80-column quarters), per-wave output
// 64 x 80 = 4 x 5 MFMA tiles, K tile = three LDS-DMA units (A | B0 | B1) over two LDS buffers, two phases per K tile, the
// groups one barrier apart so that one group's MFMA segment covers the other's fragment reads and DMA issue.
//
// What is new, and why ("round 3"):
//   * PERSISTENT: one block per CU walks tiles b, b+G, b+2G, ...; the operand stream (LDS-DMA) runs continuously across
//     tile boundaries -- the first units of the next tile are issued during the last K tiles of the current one, so a tile
//     has neither a launch / prologue bubble nor a drained pipeline at its end.
//   * DMA COMPLETION WITHOUT THE VMEM COUNTER.  On gfx950 loads and stores share one counter and retire out of order with
//     respect to each other, so a wave that streams operands AND stores results can only wait for "all but the N youngest
//     loads" by also waiting for its stores -- that made every persistent variant of round 2 pay for its epilogue stores
//     inside the operand stream (ca_gemm_pp3.h).  Here every DMA unit is followed by one 4-byte LDS-DMA that fetches the
//     unit's sequence number from a global table into an LDS flag word of the issuing wave: loads return IN ORDER among
//     themselves (the property a counted vmcnt relies on), so when the flag shows the number, the unit has landed.  The
//     wave polls the flag with a ds_read in the shadow of its fragment reads (tools/probe_flag.hip: 1.01 polls per unit,
//     every word checked under a 5.4 TB/s mixed load).  No `s_waitcnt vmcnt` in the loop at all:
//   * EPILOGUE STRAIGHT FROM THE ACCUMULATORS, STORES FIRE-AND-FORGET.  At the end of a tile a wave applies the epilogue to
//     its 64 x 80 patch in registers and stores it; nothing ever waits for those stores, the next tile's MFMAs start as
//     soon as the arithmetic is done (its operands have been landing meanwhile).
//   * 16-BYTE STORES FROM THE MFMA LAYOUT.  With swapped operands a lane holds 4 consecutive output columns of one row per
//     MFMA tile (8-byte pieces, 32 B per row and instruction).  Which weight row feeds which fragment row is free, so the
//     weight rows of a PAIR of MFMA tiles are interleaved in groups of four: a lane then holds 8 consecutive columns
//     (16-byte stores and residual loads, 64 B per row and instruction); with GEGLU four tiles are interleaved and a lane
//     holds 8 consecutive OUTPUT columns.  The fifth 16-column tile of a wave keeps 8-byte (GEGLU: 4-byte) pieces.
//   * the residual is prefetched into registers two K tiles before the end of a tile by inline-asm loads the compiler does
//     not see (a visible load would be awaited with vmcnt(0) as soon as stores are pending: LLVM treats the mixed counter
//     as out of order, rightly); they are awaited with an exact counted vmcnt -- everything younger is an LDS-DMA load,
//     everything older that is a store belongs to the previous tile and is long gone.
//   * LayerNorm statistics for the consumer (ca_gemm_args.row_sums_out): (sum, sum of squares) of a wave's 80 stored
//     columns per row go through an LDS scratch, the four column quarters are added in fixed order after the next barrier.
//   * epilogue parameters (column sums, bias, up to two row-bias groups, LayerNorm row statistics / partial sums) travel
//     as extra LDS-DMA pieces ahead of a tile's first operand unit into a double-buffered parameter block.
// Requirements (else the plan keeps the other kernels): N % 320 == 0, >= 2 K tiles, fp16 / bf16 output, operands
// addressable with 32-bit byte offsets, row-bias groups of a multiple of 64 rows, ln_parts <= 4, no split-K.
//
// Epilogue arithmetic and rounding order are those of gemm_epilogue (ca_gemm_core.h): round((acc [LN fold] + bias +
// rowbias) * alpha), then + residual, * post, activation, GEGLU, round.

#include "ca_gemm_seq.h"

// exact counted wait for 0 <= n <= 23 (larger: 23 = a safe over-wait); one asm statement, a balanced tree of scalar compares
__device__ __forceinline__ void ca_ps_vm_wait(int n) {
  n = __builtin_amdgcn_readfirstlane(n > 23 ? 23 : (n < 0 ? 0 : n));
  asm volatile(
      "s_cmp_ge_i32 %0, 12\n\ts_cbranch_scc1 30f\n\t"
      "s_cmp_ge_i32 %0, 6\n\ts_cbranch_scc1 31f\n\t"
      "s_cmp_ge_i32 %0, 3\n\ts_cbranch_scc1 32f\n\t"
      "s_cmp_ge_i32 %0, 1\n\ts_cbranch_scc1 33f\n\t"
      "s_waitcnt vmcnt(0)\n\ts_branch 99f\n"
      "33:\n\ts_cmp_ge_i32 %0, 2\n\ts_cbranch_scc1 34f\n\ts_waitcnt vmcnt(1)\n\ts_branch 99f\n"
      "34:\n\ts_waitcnt vmcnt(2)\n\ts_branch 99f\n"
      "32:\n\ts_cmp_ge_i32 %0, 4\n\ts_cbranch_scc1 35f\n\ts_waitcnt vmcnt(3)\n\ts_branch 99f\n"
      "35:\n\ts_cmp_ge_i32 %0, 5\n\ts_cbranch_scc1 36f\n\ts_waitcnt vmcnt(4)\n\ts_branch 99f\n"
      "36:\n\ts_waitcnt vmcnt(5)\n\ts_branch 99f\n"
      "31:\n\ts_cmp_ge_i32 %0, 9\n\ts_cbranch_scc1 37f\n\ts_cmp_ge_i32 %0, 7\n\ts_cbranch_scc1 38f\n\ts_waitcnt vmcnt(6)\n\ts_branch 99f\n"
      "38:\n\ts_cmp_ge_i32 %0, 8\n\ts_cbranch_scc1 39f\n\ts_waitcnt vmcnt(7)\n\ts_branch 99f\n"
      "39:\n\ts_waitcnt vmcnt(8)\n\ts_branch 99f\n"
      "37:\n\ts_cmp_ge_i32 %0, 10\n\ts_cbranch_scc1 40f\n\ts_waitcnt vmcnt(9)\n\ts_branch 99f\n"
      "40:\n\ts_cmp_ge_i32 %0, 11\n\ts_cbranch_scc1 41f\n\ts_waitcnt vmcnt(10)\n\ts_branch 99f\n"
      "41:\n\ts_waitcnt vmcnt(11)\n\ts_branch 99f\n"
      "30:\n\ts_cmp_ge_i32 %0, 18\n\ts_cbranch_scc1 42f\n\ts_cmp_ge_i32 %0, 15\n\ts_cbranch_scc1 43f\n"
      "s_cmp_ge_i32 %0, 13\n\ts_cbranch_scc1 44f\n\ts_waitcnt vmcnt(12)\n\ts_branch 99f\n"
      "44:\n\ts_cmp_ge_i32 %0, 14\n\ts_cbranch_scc1 45f\n\ts_waitcnt vmcnt(13)\n\ts_branch 99f\n"
      "45:\n\ts_waitcnt vmcnt(14)\n\ts_branch 99f\n"
      "43:\n\ts_cmp_ge_i32 %0, 16\n\ts_cbranch_scc1 46f\n\ts_waitcnt vmcnt(15)\n\ts_branch 99f\n"
      "46:\n\ts_cmp_ge_i32 %0, 17\n\ts_cbranch_scc1 47f\n\ts_waitcnt vmcnt(16)\n\ts_branch 99f\n"
      "47:\n\ts_waitcnt vmcnt(17)\n\ts_branch 99f\n"
      "42:\n\ts_cmp_ge_i32 %0, 21\n\ts_cbranch_scc1 48f\n\ts_cmp_ge_i32 %0, 19\n\ts_cbranch_scc1 49f\n\ts_waitcnt vmcnt(18)\n\ts_branch 99f\n"
      "49:\n\ts_cmp_ge_i32 %0, 20\n\ts_cbranch_scc1 50f\n\ts_waitcnt vmcnt(19)\n\ts_branch 99f\n"
      "50:\n\ts_waitcnt vmcnt(20)\n\ts_branch 99f\n"
      "48:\n\ts_cmp_ge_i32 %0, 22\n\ts_cbranch_scc1 51f\n\ts_waitcnt vmcnt(21)\n\ts_branch 99f\n"
      "51:\n\ts_cmp_ge_i32 %0, 23\n\ts_cbranch_scc1 52f\n\ts_waitcnt vmcnt(22)\n\ts_branch 99f\n"
      "52:\n\ts_waitcnt vmcnt(23)\n"
      "99:\n"
      :
      : "s"(n)
      : "scc", "memory");
}

// output column (inside a wave's 80-column quarter) of fragment row i of MFMA tile j -- the interleave described above
__device__ __forceinline__ int ca_ps_col(int j, int i, bool geglu) {
  if (j == 4) return 64 + i;
  if (geglu) return 16 * (i >> 2) + 4 * j + (i & 3);
  return 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3);
}

// FLAGS = false (experiment, CA_PS_FLAGS=0): the same kernel with COUNTED vmcnt waits instead of the LDS flags -- every VMEM
// instruction of a wave is counted (`issued`), a unit remembers the count at its issue and is awaited with
// vmcnt(issued - mark), which allows the epilogue's stores to stay outstanding behind it.  That is only correct if loads and
// stores retire in issue order (what LLVM's waitcnt insertion assumes on gfx9, and what tools/probe_flag.hip mode 1 did not
// contradict in 36 million checked slots; the round-2 report of out-of-order retirement may have been a miscount).
template <int DT, int MODE, bool FLAGS = true>
__global__ __launch_bounds__(512, 2) void k_gemm_ps(GemmKParams p, int tiles_total, unsigned c_bytes, unsigned res_bytes) {
  constexpr int BM = 128, BN = 320, KT = 64;
  constexpr int TM = 4, TN = 5;
  constexpr int A_ROWS = 128, B0_ROWS = 128, B1_ROWS = 192;
  constexpr int OFF_A = 0, OFF_B0 = A_ROWS * KT, OFF_B1 = (A_ROWS + B0_ROWS) * KT;
  constexpr int BUF = (A_ROWS + B0_ROWS + B1_ROWS) * KT;  // elements of one K tile (56 KB)
  // byte layout of the single LDS array
  constexpr int PAR_BASE = 2 * BUF * 2;
  constexpr int P_CS = 0, P_BI = 2048, P_RB0 = 4096, P_RB1 = 6144, P_ST = 8192, PSET = 12288;  // (param pieces are 2 x 1 KB: regions 2 KB apart)
  constexpr int FLAG_BASE = PAR_BASE + 2 * PSET;  // per wave: 4 flag slots of 256 B (kind AB0 | B1) x (buffer parity)
  constexpr int RS_BASE = FLAG_BASE + 8 * 4 * 256;  // row-sum scratch: [128 rows][4 quarters] float2
  constexpr int SMEM_BYTES = RS_BASE + 128 * 4 * 8;
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
  // XCD-aware: block b runs on XCD b % 8; give every XCD a contiguous 1/8 of each round of G tiles
  const int bslot = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
  const int my_tiles = bslot < tiles_total ? (tiles_total - bslot + G - 1) / G : 0;
  if (my_tiles == 0) return;

  for (int i = tid; i < 8 * 4 * 64; i += 512) reinterpret_cast<unsigned*>(smem_b + FLAG_BASE)[i] = 0xFFFFFFFFu;
  __syncthreads();

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, p.a2 ? p.a2_bytes : p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(p.c, 0, c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_cs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ln_colsum ? (const void*)p.ln_colsum : (const void*)p.w), 0, p.ln_colsum ? (unsigned)p.n * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)p.w), 0, p.bias ? (unsigned)p.n * 4u : 0u, 0x00020000);
  const unsigned st_row_bytes = p.ln_parts > 0 ? (unsigned)p.ln_parts * 8u : 8u;
  const __amdgpu_buffer_rsrc_t rs_st = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ln_stats ? (const void*)p.ln_stats : (const void*)p.w), 0, p.ln_stats ? (unsigned)p.m * st_row_bytes : 0u, 0x00020000);
  // (the residual is loaded by inline asm: its descriptor as four plain SGPR words)
  const unsigned long long res_addr = (unsigned long long)(p.res ? (const void*)p.res : (const void*)p.c);
  const u32x4 rs_res = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)res_addr), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((res_addr >> 32) & 0xffffu)),
                        (unsigned)__builtin_amdgcn_readfirstlane((int)(p.res ? res_bytes : 0u)), 0x00020000u};
  const unsigned rb_groups = p.rowbias ? (unsigned)((p.m + p.rows_per_group - 1) / p.rows_per_group) : 0u;
  const __amdgpu_buffer_rsrc_t rs_rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.rowbias ? (const void*)p.rowbias : (const void*)p.w), 0,
                                                                         p.rowbias ? (unsigned)(((int64_t)(rb_groups - 1) * p.ld_rowbias + p.n) * 4) : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_seq = __builtin_amdgcn_make_buffer_rsrc((void*)ca_seq_table.v, 0, 4096u, 0x00020000);

#ifdef CA_STAMPS  // (python -m controlanimate_amd._build --experiments --stamps: the stamp code costs registers -- the 256 x 320 kernel spills with it)
  // timing experiment (CA_PP_DBG=9): block 0, waves 0 and 4 (one of each group) stamp the shader clock into p.partial
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
  const int nt = p.taps * kct;            // K tiles per output tile (>= 2)
  const int total_kt = my_tiles * nt;     // K tiles of this block's whole stream
  const bool geglu = p.geglu != 0;

  // ---------------------------------------------------------------- DMA side (runs ~2 K tiles ahead)
  // Per-lane source offsets are kept per DMA piece as a VGPR "voffset" that already contains the lane's row and its
  // swizzled 16-byte chunk; the position along K is a wave-uniform SGPR "soffset" of the instruction.  Rows past M (and
  // conv halo taps) get OOB_V: far beyond any descriptor we accept (< 2 GB) -> the hardware writes zeros.
  constexpr unsigned OOB_V = 0x80000000u;
  unsigned b0_v[2] = {0, 0}, b1_v[3] = {0, 0, 0};
  unsigned a1_v[2] = {0, 0}, a2_v[2] = {0, 0};     // dense A, source 1 / 2
  int a_img[2] = {0, 0}, a_ho[2] = {0, 0}, a_wo[2] = {0, 0};  // conv A
  bool a_ok[2] = {false, false};
  int d_seq = 0;                // tile sequence number of the stream head
  int d_t = 0;                  // K tile (within its tile) of the stream head
  int d_tap = 0, d_c0 = 0;      // the same position as (tap, first channel)
  int d_ab0 = 0, d_b1 = 0;      // K tiles (of the whole stream) whose A/B0 resp. B1 unit has been issued
  bool d_live = true;
  int issued = 0;               // VMEM instructions issued by this wave so far (for the residual wait)
  unsigned char* const my_flags = smem_b + FLAG_BASE + wid * 4 * 256;

  auto tile_of = [&](int seq, int& tm, int& tn) __attribute__((always_inline)) -> bool {
    const int id = seq * G + bslot;
    if (id >= tiles_total) return false;
    tile_coords((unsigned)id, tiles_m, tiles_n, tm, tn);
    return true;
  };

  auto dma_set_tile = [&](int seq, int m0, int n0) __attribute__((always_inline)) {
    // everything below is recomputed from the lane id on purpose: hipcc hoists lane-dependent invariants out of the
    // tile loop and then SPILLS them (ca_gemm_pp3.h).  The empty asm makes the lane id opaque here.
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int r8 = lane_o >> 3, cp = lane_o & 7;
    const int ab_chunk0 = cp ^ swz(wid * 16 + r8), b1_chunk0 = cp ^ swz(wid * 24 + r8);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = wid * 16 + i * 8 + r8;  // A and B0 pieces stage rows wid*16 + 8i + r8: chunk_i = chunk_0 ^ 4i
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
      // B0 local row r: quarter r >> 5, MFMA tile (r >> 4) & 1, fragment row r & 15
      b0_v[i] = (unsigned)(n0 + (r >> 5) * 80 + ca_ps_col((r >> 4) & 1, r & 15, geglu)) * wld * 2u + ch;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      // B1 local row r1 (pieces stage rows wid*24 + 8i + r8; chunk_i = chunk_0 ^ 4i, 16 rows = the same swizzle):
      // quarter r1 / 48, MFMA tile 2 + (r1 % 48) / 16, fragment row r1 & 15 (48 = 3 x 16)
      const int r1 = wid * 24 + i * 8 + r8;
      b1_v[i] = (unsigned)(n0 + (r1 / 48) * 80 + ca_ps_col(2 + (r1 % 48) / 16, r1 & 15, geglu)) * wld * 2u + (unsigned)((b1_chunk0 ^ (4 * (i & 1))) * 16);
    }
    d_tap = 0;
    d_c0 = 0;
    // epilogue parameters of this tile -> parameter set (seq & 1).  Pieces of 1 KB (64 lanes x 16 B):
    //   wave 0: colsum[0:256), colsum[256:512)   wave 1: bias   wave 2: rowbias group 0   wave 3: rowbias group 1
    //   waves 4..7: LayerNorm statistics / partial sums of the 128 rows, 1 KB each
    unsigned char* pset = smem_b + PAR_BASE + (seq & 1) * PSET;
#define CA_PS_PAR2(RS, OFFS, BASE)                                                                                               \
  {                                                                                                                              \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(pset + (OFFS)), 16, (BASE) + lane_o * 16u, 0, 0, 0);          \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (__attribute__((address_space(3))) void*)(pset + (OFFS) + 1024), 16, (BASE) + 1024u + lane_o * 16u, 0, 0, 0); \
    issued += 2;                                                                                                                 \
  }
    // (an absent operand has a descriptor of size 0: the DMA writes zeros -- the epilogue adds them unconditionally)
    if (wid == 0) {
      CA_PS_PAR2(rs_cs, P_CS, (unsigned)n0 * 4u)
    } else if (wid == 1) {
      CA_PS_PAR2(rs_bi, P_BI, (unsigned)n0 * 4u)
    } else if (wid == 2) {
      CA_PS_PAR2(rs_rb, P_RB0, (unsigned)n0 * 4u + (unsigned)(p.rowbias ? m0 / p.rows_per_group : 0) * (unsigned)p.ld_rowbias * 4u)
    } else if (wid == 3) {
      const bool has = p.rowbias && m0 / p.rows_per_group + 1 < (int)rb_groups;
      CA_PS_PAR2(rs_rb, P_RB1, has ? (unsigned)n0 * 4u + (unsigned)(m0 / p.rows_per_group + 1) * (unsigned)p.ld_rowbias * 4u : OOB_V)
#undef CA_PS_PAR2
    } else if (p.ln_stats) {
      const int q = wid - 4;  // KB number q of the tile's statistics block (128 rows x st_row_bytes)
      if ((unsigned)q * 1024u < 128u * st_row_bytes) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_st, (__attribute__((address_space(3))) void*)(pset + P_ST + q * 1024), 16,
                                                 (unsigned)m0 * st_row_bytes + (unsigned)q * 1024u + lane_o * 16u, 0, 0, 0);
        issued += 1;
      }
    }
  };

  int mark_u[4] = {0, 0, 0, 0};  // FLAGS == false: `issued` right after the unit in flag slot s
  auto issue_flag = [&](int slot, int seqno) __attribute__((always_inline)) {
    if (!FLAGS) {
      if (slot == 0) mark_u[0] = issued;
      else if (slot == 1) mark_u[1] = issued;
      else if (slot == 2) mark_u[2] = issued;
      else mark_u[3] = issued;
      return;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_seq, (__attribute__((address_space(3))) void*)(my_flags + slot * 256), 4, 0u, (unsigned)(seqno & 1023) * 4u, 0, 0);
    issued += 1;
  };

  auto issue_ab0 = [&]() __attribute__((always_inline)) {  // A and B0 of the K tile at the stream head into buffer (d_ab0 & 1)
    if (!d_live) return;
    const int par = d_ab0 & 1;
    u16* buf = smem + par * BUF;
    const unsigned wk = (unsigned)(d_tap * kc + d_c0) * 2u;  // weights: K runs over (tap, channel)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_B0 + (wid * 2 + i) * 8 * KT), 16, b0_v[i], wk, 0, 0);
    const bool src2 = d_c0 >= p.c1;  // c1 % 64 == 0: a K tile never straddles the two sources
    if (MODE == 1) {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int ab_chunk0 = (lane_o & 7) ^ swz(wid * 16 + (lane_o >> 3));
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
    issue_flag(par, d_ab0);
    ++d_ab0;
  };
  auto issue_b1 = [&]() __attribute__((always_inline)) {  // B1 of the same K tile as the last issue_ab0, then the stream advances
    if (!d_live) return;
    const int par = d_b1 & 1;
    u16* buf = smem + par * BUF;
    const unsigned wk = (unsigned)(d_tap * kc + d_c0) * 2u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_B1 + (wid * 3 + 0) * 8 * KT), 16, b1_v[0], wk, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_B1 + (wid * 3 + 1) * 8 * KT), 16, b1_v[1], wk, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_B1 + (wid * 3 + 2) * 8 * KT), 16, b1_v[2], wk, 0, 0);
    issued += 3;
    issue_flag(2 + par, d_b1);
    ++d_b1;
    // advance the stream head: the nine taps of one 64-channel tile follow each other (k_tile_split, tap_inner: the input
    // rows a block touches nine times stay in its XCD's L2 between the touches)
    ++d_t;
    if (p.taps == 1) {
      d_c0 += KT;
    } else if (++d_tap == p.taps) {
      d_tap = 0;
      d_c0 += KT;
    }
  };

  // the unit in flag slot `slot` carries sequence number `seqno`: poll until it has landed.  `fv` = the flag value read
  // earlier in the phase (behind the fragment reads); the slow path re-reads with a short sleep and gives up after ~2^20
  // polls (a hung wave would take the whole device down; wrong results are caught by the tests, a hang is not).
  auto flag_begin = [&](int slot) __attribute__((always_inline)) -> unsigned {
    if (!FLAGS) return 0u;
    unsigned fv;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(my_flags + slot * 256);
    asm volatile("ds_read_b32 %0, %1" : "=v"(fv) : "v"(addr) : "memory");
    return fv;
  };
  auto flag_finish = [&](int slot, int seqno, unsigned fv) __attribute__((always_inline)) {
    if (!FLAGS) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int n = issued - (slot == 0 ? mark_u[0] : slot == 1 ? mark_u[1] : slot == 2 ? mark_u[2] : mark_u[3]);
      if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");  // the steady state: one A/B0 unit (4) and one B1 unit (3) younger
      else ca_ps_vm_wait(n);
      return;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fv)::"memory");
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

  // residual of the current tile in the accumulator layout: per row tile i two 16-byte pieces (column pairs) + one
  // 8-byte piece (fifth MFMA tile); with GEGLU there is no residual (checked by the launcher)
  u32x4 res16[TM][2];
  u32x2 res8[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {  // (no residual: zeros, added unconditionally)
    res16[i][0] = res16[i][1] = (u32x4){0u, 0u, 0u, 0u};
    res8[i] = (u32x2){0u, 0u};
  }
  int mark_res = 0;
  auto res_prefetch = [&](int m0, int n0) __attribute__((always_inline)) {
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int l15o = lane_o & 15, go = lane_o >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wr * 64 + i * 16 + l15o;
      const unsigned rowoff = m < p.m ? (unsigned)m * (unsigned)p.ld_res * 2u + (unsigned)(n0 + wc * 80) * 2u : OOB_V;  // (out of range reads 0)
      const unsigned o0 = rowoff + (unsigned)(8 * go) * 2u, o1 = rowoff + (unsigned)(32 + 8 * go) * 2u, o2 = rowoff + (unsigned)(64 + 4 * go) * 2u;
      // (one statement, led by s_nop 4: the descriptor may just have been restored from an SGPR spill lane by v_readlane, and a
      //  VMEM instruction needs five wait states after a VALU write of an SGPR it reads -- hipcc's hazard recogniser does not
      //  look inside inline asm; found in ca_gemm_pq.h, where it faulted.  Early-clobber outputs: a load's data may return
      //  before the next one has read its address register.)
      asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %3, %6, 0 offen\n\tbuffer_load_dwordx4 %1, %4, %6, 0 offen\n\tbuffer_load_dwordx2 %2, %5, %6, 0 offen"
                   : "=&v"(res16[i][0]), "=&v"(res16[i][1]), "=&v"(res8[i])
                   : "v"(o0), "v"(o1), "v"(o2), "s"(rs_res)
                   : "memory");
    }
    issued += 3 * TM;
    mark_res = issued;
  };
  auto res_wait = [&]() __attribute__((always_inline)) {
    ca_ps_vm_wait(issued - mark_res);
    // (the destinations are named so that no use is scheduled above the wait: cdna_hip_programming.md 5.7 item 1, form ii)
    asm volatile("" : "+v"(res16[0][0]), "+v"(res16[0][1]), "+v"(res16[1][0]), "+v"(res16[1][1]), "+v"(res16[2][0]), "+v"(res16[2][1]), "+v"(res16[3][0]), "+v"(res16[3][1])::"memory");
    asm volatile("" : "+v"(res8[0]), "+v"(res8[1]), "+v"(res8[2]), "+v"(res8[3])::"memory");
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

  // row sums (ca_gemm_args.row_sums_out) of the tile whose epilogue ran last: written out after the next barrier
  bool rs_pending = false;
  int rs_m0 = 0, rs_tn = 0;
  auto rs_flush = [&]() __attribute__((always_inline)) {
    if (!rs_pending) return;
    rs_pending = false;
    if (wc != 0) return;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int row = wr * 64 + lane_o, m = rs_m0 + row;
    const f32x4 q0 = *reinterpret_cast<const f32x4*>(smem_b + RS_BASE + row * 32), q1 = *reinterpret_cast<const f32x4*>(smem_b + RS_BASE + row * 32 + 16);
    const float a = (((0.f + q0[0]) + q0[2]) + q1[0]) + q1[2], b = (((0.f + q0[1]) + q0[3]) + q1[1]) + q1[3];  // (quarters in order)
    if (m < p.m) *reinterpret_cast<float2*>(p.row_sums + ((int64_t)m * tiles_n + rs_tn) * 2) = make_float2(a, b);
    // (NOT counted in `issued`: the compiler may branch around the store when no lane is active; an uncounted store only makes
    //  the counted waits of the FLAGS == false variant one operation stricter, a phantom one would make them too weak)
  };

  // ---- the epilogue of one tile, from the accumulators (see the header)
  auto epilogue = [&](int seq, int m0, int n0, int tn) __attribute__((always_inline)) {
    const unsigned char* pset = smem_b + PAR_BASE + (seq & 1) * PSET;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int l15 = lane_o & 15, g = lane_o >> 4;
    // LayerNorm statistics of this lane's four rows.  (LDS reads narrower than 16 bytes of a region filled by LDS-DMA are
    // preceded by a compiler-inserted vmcnt(0) -- hipcc 7.2 treats them as possibly aliasing a DMA in flight, 16-byte reads
    // not -- which would drain the operand stream once per tile: 16-byte reads where the layout allows, else inline asm.)
    float2 st[TM];
    if (p.ln_stats) {
      int kc_o = kc;
      asm volatile("" : "+s"(kc_o));  // (recomputed per tile: hoisted out of the tile loop the quotient is spilled, and its reload drains the DMA queue)
      const float inv = 1.f / (float)kc_o;
      if (p.ln_parts == 2 || p.ln_parts == 4) {  // partial sums left by the producing GEMM's epilogue: finish (mean, rstd) here, in order
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wr * 64 + i * 16 + l15;
          float a, b;
          if (p.ln_parts == 2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(pset + P_ST + row * 16);
            a = (0.f + v[0]) + v[2];
            b = (0.f + v[1]) + v[3];
          } else {
            const f32x4 v = *reinterpret_cast<const f32x4*>(pset + P_ST + row * 32), w = *reinterpret_cast<const f32x4*>(pset + P_ST + row * 32 + 16);
            a = (((0.f + v[0]) + v[2]) + w[0]) + w[2];
            b = (((0.f + v[1]) + v[3]) + w[1]) + w[3];
          }
          const float mean = a * inv;
          st[i] = make_float2(mean, rsqrtf(fmaxf(b * inv - mean * mean, 0.f) + p.ln_eps));
        }
      } else {  // 8 bytes per row: (mean, rstd), or ONE partial sum
        u32x2 s0, s1, s2, s3;
        const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(pset + P_ST + (wr * 64 + l15) * 8);
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:128\n\tds_read_b64 %2, %4 offset:256\n\tds_read_b64 %3, %4 offset:384\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3) : "v"(addr) : "memory");
        const u32x2 sv[TM] = {s0, s1, s2, s3};
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float x = __uint_as_float(sv[i][0]), y = __uint_as_float(sv[i][1]);
          if (p.ln_parts == 1) {
            const float mean = (0.f + x) * inv;
            st[i] = make_float2(mean, rsqrtf(fmaxf((0.f + y) * inv - mean * mean, 0.f) + p.ln_eps));
          } else {
            st[i] = make_float2(x, y);
          }
        }
      }
    }
    if (!p.ln_stats) {
#pragma unroll
      for (int i = 0; i < TM; ++i) st[i] = make_float2(0.f, 1.f);  // 1 * (x - 0 * 0) = x: the LayerNorm fold is then an exact no-op
    }
    int rb_sel = 0;
    if (p.rowbias) rb_sel = (m0 + wr * 64) / p.rows_per_group - m0 / p.rows_per_group;  // 0 or 1 (rows_per_group % 64 == 0)
    const unsigned char* rbp = pset + (rb_sel ? P_RB1 : P_RB0);
    unsigned rowoff[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wr * 64 + i * 16 + l15;
      rowoff[i] = (m < p.m && p.dbg != 1) ? (unsigned)m * (unsigned)p.ldc * 2u + (unsigned)((geglu ? (n0 >> 1) + wc * 40 : n0 + wc * 80)) * 2u : OOB_V;  // (out of range: the store is dropped)
    }
    float rsum[TM], rsq[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) rsum[i] = rsq[i] = 0.f;

    // BRANCH-FREE arithmetic: an absent operand is an exact identity -- column sums / bias / row bias come out of LDS regions
    // the DMA filled with zeros (descriptor of size 0), no LayerNorm is (mean, rstd) = (0, 1), no residual is a register
    // of zeros, alpha = 1.  (The first version tested p.bias, p.ln_stats, p.res, ... inside the unrolled loops: 100 KB of
    // code for an instruction cache of 64 KB, and an epilogue of ~8 000 (GEGLU: 17 000) cycles per wave group -- measured with
    // s_memtime stamps, tools/ps_stamps.py -- most of it instruction fetch.)
    // parameters of the lane's 4 consecutive columns starting at c0 (inside the 320-column tile); staged value of an
    // accumulator quad: (acc [LN fold] + bias + rowbias) * alpha (rounded by the caller)
    struct Par {
      f32x4 cs, bi, rb;
    };
    auto load_par = [&](int c0) __attribute__((always_inline)) -> Par {
      Par q;
      q.cs = *reinterpret_cast<const f32x4*>(pset + P_CS + c0 * 4);
      q.bi = *reinterpret_cast<const f32x4*>(pset + P_BI + c0 * 4);
      q.rb = *reinterpret_cast<const f32x4*>(rbp + c0 * 4);
      return q;
    };
    auto staged = [&](int i, int j, const Par& q, float (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = st[i].y * (acc[i][j][r] - st[i].x * q.cs[r]);
        x = (x + q.bi[r]) + q.rb[r];  // same association as gemm_epilogue
        v[r] = x * p.alpha;
      }
      acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };

    if (p.res) res_wait();
    if (!geglu) {
#pragma unroll
      for (int pr = 0; pr < 3; ++pr) {  // column pairs (MFMA tiles 0|1, 2|3) and the single fifth tile
        const int nq = pr < 2 ? 2 : 1;
        Par par[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (u < nq) par[u] = load_par(wc * 80 + (pr < 2 ? 32 * pr + 8 * g + 4 * u : 64 + 4 * g));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (u >= nq) break;
            float v[4];
            staged(i, pr * 2 + u, par[u], v);
            w[2 * u] = pack2<DT>(v[0], v[1]);
            w[2 * u + 1] = pack2<DT>(v[2], v[3]);
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k >= 2 * nq) break;
            const unsigned rr = pr < 2 ? res16[i][pr & 1][k] : res8[i][k & 1];
            if (DT == CA_F16) {  // fp16 + fp16 is exact in fp32: the packed add rounds exactly like the fp32 path
              unsigned s_;
              asm("v_pk_add_f16 %0, %1, %2" : "=v"(s_) : "v"(w[k]), "v"(rr));
              w[k] = s_;
            } else {
              w[k] = pack2<DT>(Elem<DT>::to_f((u16)(w[k] & 0xffffu)) + Elem<DT>::to_f((u16)(rr & 0xffffu)),
                               Elem<DT>::to_f((u16)(w[k] >> 16)) + Elem<DT>::to_f((u16)(rr >> 16)));
            }
          }
          if (p.row_sums) {  // of the values as stored (rounded)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (k >= 2 * nq) break;
              const float lo = Elem<DT>::to_f((u16)(w[k] & 0xffffu)), hi = Elem<DT>::to_f((u16)(w[k] >> 16));
              rsum[i] += lo + hi;
              rsq[i] = fmaf(lo, lo, fmaf(hi, hi, rsq[i]));
            }
          }
          if (pr < 2) __builtin_amdgcn_raw_buffer_store_b128((u32x4){w[0], w[1], w[2], w[3]}, rs_c, rowoff[i] + (unsigned)(32 * pr + 8 * g) * 2u, 0, 0);
          else __builtin_amdgcn_raw_buffer_store_b64((u32x2){w[0], w[1]}, rs_c, rowoff[i] + (unsigned)(64 + 4 * g) * 2u, 0, 0);
        }
      }
      issued += 3 * TM;
    } else {
      // GEGLU: weight rows interleaved (h, g); MFMA tiles 0..3 interleaved so that a lane holds 16 consecutive weight rows
      // = 8 consecutive output columns; tile 4 keeps 2 outputs per lane
      Par par[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) par[j] = load_par(wc * 80 + (j < 4 ? 16 * g + 4 * j : 64 + 4 * g));
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        unsigned w[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          float v[4];
          staged(i, j, par[j], v);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = Elem<DT>::to_f(Elem<DT>::from_f(v[r]));  // (the Linear's output is rounded first)
          const f32x2 gg = gelu_erf_f2((f32x2){v[1], v[3]});
          w[j] = pack2<DT>(v[0] * gg[0], v[2] * gg[1]);
        }
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){w[0], w[1], w[2], w[3]}, rs_c, rowoff[i] + (unsigned)(8 * g) * 2u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(w[4], rs_c, rowoff[i] + (unsigned)(32 + 2 * g) * 2u, 0, 0);
      }
      issued += 2 * TM;
    }
    if (p.row_sums) {
      // sum over the four 16-lane groups of a wave (lanes l, l^16, l^32, l^48 hold pieces of one row), then one lane
      // group writes the wave's (sum, sum of squares) of its 80 columns; the quarters are added after the next barrier
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        float a = rsum[i], b = rsq[i];
        a += __shfl_xor(a, 16);
        b += __shfl_xor(b, 16);
        a += __shfl_xor(a, 32);
        b += __shfl_xor(b, 32);
        if (g == 0) {  // (inline asm: a compiler-visible 8-byte LDS store is preceded by vmcnt(0), as the narrow reads above)
          const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem_b + RS_BASE + (wr * 64 + i * 16 + l15) * 32 + wc * 8);
          const u32x2 ab = {__float_as_uint(a), __float_as_uint(b)};
          asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(ab) : "memory");
        }
      }
      rs_pending = true;
      rs_m0 = m0;
      rs_tn = tn;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (parameter reads retired: the set is re-filled two tiles later)
  };

  // ---------------------------------------------------------------- run
#define CA_PS_SET_TILE(SEQ)                                    \
  {                                                            \
    int tm_, tn_;                                              \
    d_live = tile_of((SEQ), tm_, tn_);                         \
    if (d_live) dma_set_tile((SEQ), tm_ * BM, tn_ * BN);       \
  }
  CA_PS_SET_TILE(0)
  issue_ab0();
  issue_b1();
  if (nt == 1) {  // (never: the launcher requires >= 2 K tiles)
    return;
  }
  issue_ab0();
  {  // K tile 0 complete (A/B0 of K tile 1 may be in flight)
    const unsigned f0 = flag_begin(0), f1 = flag_begin(2);
    flag_finish(0, 0, f0);
    flag_finish(2, 0, f1);
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
      // ---- phase 1: A + B0 of this K tile; issue B1 of the next one; B1 of this one must have landed for phase 2
      __builtin_amdgcn_sched_barrier(0);
      const unsigned fl1 = flag_begin(2 + par);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i][s] = ld16(buf + fa_base[s] + i * 16 * KT);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb0[j][s] = ld16(buf + fb0_base[s] + j * 16 * KT);
      }
      if (p.res && t == (nt >= 3 ? nt - 2 : 0)) res_prefetch(m0, n0);
      issue_b1();
      flag_finish(2 + par, cv, fl1);
      stamp(2);
      __builtin_amdgcn_s_barrier();
      stamp(3);
      rs_flush();
      mfma_p1();
      stamp(4);
      __builtin_amdgcn_s_barrier();
      stamp(5);
      // ---- phase 2: B1 of this K tile; issue A/B0 of the K tile after next (this buffer: its A/B0 were retired in phase 1)
      const unsigned fl2 = flag_begin(par ^ 1);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 3; ++j) fb1[j][s] = ld16(buf + fb1_base[s] + j * 16 * KT);
      if (d_t == nt) {  // the stream enters the next tile
        d_t = 0;
        ++d_seq;
        CA_PS_SET_TILE(d_seq)
      }
      issue_ab0();
      if (cv + 1 < total_kt) flag_finish(par ^ 1, cv + 1, fl2);  // A/B0 of the next K tile (read in the next phase)
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      stamp(6);
      __builtin_amdgcn_s_barrier();
      stamp(7);
      mfma_p2();
      stamp(8);
      __builtin_amdgcn_s_barrier();
      ++cv;
    }
    // The groups run one barrier apart, so group 0's epilogue would sit beside group 1's last (short) MFMA segment and group
    // 1's beside group 0's next one: the two epilogues in series (measured: group 1 idles at its barrier for as long as group
    // 0's epilogue takes).  One extra barrier for group 0 BEFORE its epilogue and one for group 1 AFTER puts both epilogues
    // into the same interval and restores the stagger behind them.
    if (wr == 0) __builtin_amdgcn_s_barrier();
    stamp(9);
    epilogue(seq, m0, n0, tn);
    stamp(10);
    if (wr == 1) __builtin_amdgcn_s_barrier();
  }
#undef CA_PS_SET_TILE
  if (wr == 0) __builtin_amdgcn_s_barrier();
  if (rs_pending) {
    __syncthreads();
    rs_flush();
  }
}

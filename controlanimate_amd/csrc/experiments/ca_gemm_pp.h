// Ping-pong MFMA GEMM / implicit-GEMM 3x3 convolution main loop for the MFMA-bound shapes (K >= 640).
// Included by ca_gemm.hip inside its anonymous namespace (shares GemmKParams, tile_coords and the
// LDS-staged epilogue).
//
// Geometry: block tile 256 x BN x 64, 8 waves = 2 groups (M halves, `wr`) x 4 (N quarters, `wc`), one
// wave of each group per SIMD, per-wave output 128 x BN/4 (BN = 256: 8 x 4 MFMA 16x16x32 tiles; 12
// fragment reads per 32 MFMAs instead of 16 with the 64x64 wave tiles of k_gemm_dma).  One block per CU:
// 2 K-tile buffers of (256 + BN) x 128 B in LDS (128 KB at BN = 256).
//
// Schedule (cdna_hip_programming.md, "256^2 8-phase template", re-derived for this operand layout):
//   * a K tile is processed in 4 phases, one C quadrant (64 x BN/8 per wave, 16 MFMAs at BN = 256) each:
//       phase 1: read B[nh0] (4 frags) then A[mh0] (8 frags)   -> C[mh0][nh0]
//       phase 2: read B[nh1] (4)                               -> C[mh0][nh1]
//       phase 3: read A[mh1] (8)                               -> C[mh1][nh1]
//       phase 4: no LDS reads (B[nh0] is still in registers)   -> C[mh1][nh0]
//     each phase = { ds_reads ; 1 half-tile of LDS-DMA prefetch ; s_barrier ; lgkmcnt(0) ; MFMAs ; s_barrier }.
//   * the two wave groups run one barrier apart (group 1 executes one extra s_barrier up front): while one
//     group's waves are in their MFMA segment, the other group's waves (same SIMDs) read fragments and issue
//     DMA -- the matrix pipe of every SIMD always has a wave in a compute segment.
//   * operands stream as "half-tiles" of 128 rows x 64 k (16 KB = 2 DMA instructions per wave), ordered
//     B[nh0], A[mh0], B[nh1], A[mh1] -- the order in which a buffer's regions are released by the phases
//     above -- one half-tile per phase, 7 half-tiles ahead of the phase that reads them.  A wave waits with a
//     COUNTED vmcnt(6) once per K tile (phase 4): the K tile after next may still have 3 half-tiles in flight.
//     No vmcnt(0) and no __syncthreads() inside the loop: LDS-DMA stays in flight across the raw barriers.
//   * hazards (barrier numbering: group 0's phase p sits between barriers 2p-2 .. 2p, group 1's between
//     2p-1 .. 2p+1):
//       RAW  a half-tile is read one phase after the vmcnt that retired it (wait before the phase's first
//            barrier, read after its second), so every wave's share has landed and been published by a barrier;
//       WAR  a region is re-staged >= 2 phases after its last read (A[mh0]: read ph.1, staged ph.3; B[nh1]: 2 -> 4;
//            A[mh1]: 3 -> 5), or 1 phase after when the reads were retired by an lgkmcnt BEFORE the reading
//            phase's first barrier (B[nh0]: read first in ph.1 and retired by lgkmcnt(8); staged in ph.2).


template <int DT, int MODE, int BN>
__global__ __launch_bounds__(512, 2) void k_gemm_pp(GemmKParams p) {
  constexpr int BM = 256;
  constexpr int KT = 64;
  constexpr int TM = 8;            // m tiles (16 rows) per wave: 2 halves of 4
  constexpr int TN = BN / 64;      // n tiles per wave: 2 halves of TN/2
  constexpr int TNH = TN / 2;
  static_assert(BN == 256 || BN == 128, "BN");
  constexpr int BROWS = BN / 2;                 // rows of one B half-tile
  constexpr int A_HALF = 128 * KT;              // elements
  constexpr int B_HALF = BROWS * KT;
  constexpr int BUF = 2 * A_HALF + 2 * B_HALF;  // one K tile: [B nh0][A mh0][B nh1][A mh1]
  constexpr int OFF_B0 = 0, OFF_A0 = B_HALF, OFF_B1 = B_HALF + A_HALF, OFF_A1 = 2 * B_HALF + A_HALF;
  constexpr int SMEM_ELEMS = 2 * BUF > BM * (BN + 8) ? 2 * BUF : BM * (BN + 8);
  constexpr int BI = BROWS / 64;                // DMA instructions per wave per B half-tile (8 rows each)
  __shared__ __attribute__((aligned(16))) u16 smem[SMEM_ELEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid >> 2, wc = wid & 3;
  const int g = lane >> 4, l15 = lane & 15;

  const int tiles_n = (p.n + BN - 1) / BN;
  const int tiles_m = (p.m + BM - 1) / BM;
  const unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  int tile_m, tile_n;
  tile_coords(bid, tiles_m, tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, p.a2 ? p.a2_bytes : p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  auto swz = [](int row) { return (row >> 1) & 7; };
  const int r8 = lane >> 3, cp = lane & 7;  // row inside an 8-row DMA group, LDS chunk position
  const int kc = p.c1 + p.c2;
  const int kct = kc / KT;
  const unsigned wld = (unsigned)(p.taps * kc);

  // ---- DMA source state: a wave stages local rows (wid*2 + i)*8 + r8 of every A half-tile (i = 0, 1) and
  // (wid*BI + i)*8 + r8 of every B half-tile.  A half `mh` local row r -> tile row (r/64)*128 + mh*64 + r%64;
  // B half `nh` local row r -> tile column (r / (BROWS/4)) * (BN/4) + nh * (BN/8) + r % (BROWS/4).
  int a_chunk[2], a_img[2][2], a_ho[2][2], a_wo[2][2];
  unsigned a_off1[2][2], a_off2[2][2];  // [mh][i]
  bool a_ok[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wid * 2 + i) * 8 + r8;
    a_chunk[i] = cp ^ swz(r);
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      const int m = m0 + (r >> 6) * 128 + mh * 64 + (r & 63);
      a_ok[mh][i] = m < p.m;
      const int mm = a_ok[mh][i] ? m : p.m - 1;
      if (MODE == 1) {
        const int hw = p.hout * p.wout;
        a_img[mh][i] = mm / hw;
        const int rem = mm - a_img[mh][i] * hw;
        a_ho[mh][i] = rem / p.wout;
        a_wo[mh][i] = rem - a_ho[mh][i] * p.wout;
        a_off1[mh][i] = a_off2[mh][i] = 0;
      } else {
        a_img[mh][i] = a_ho[mh][i] = a_wo[mh][i] = 0;
        a_off1[mh][i] = (unsigned)((int64_t)mm * p.lda * 2);
        a_off2[mh][i] = (unsigned)((int64_t)mm * p.lda2 * 2);
      }
    }
  }
  int b_chunk[BI];
  unsigned b_off[2][BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int r = (wid * BI + i) * 8 + r8;
    b_chunk[i] = cp ^ swz(r);
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      int n = n0 + (r / (BROWS / 4)) * (BN / 4) + nh * (BN / 8) + r % (BROWS / 4);
      if (n >= p.n) n = p.n - 1;  // clamped rows feed accumulators that are never stored
      b_off[nh][i] = (unsigned)n * wld * 2u;
    }
  }

  const int nt = p.taps * kct;  // K tiles (>= 2)

  // issues half-tile h of the operand stream (h / 4 = K tile, h % 4 = B nh0 | A mh0 | B nh1 | A mh1)
  auto issue = [&](int t, int kind) {
    if (t >= nt) return;
    u16* buf = smem + (t & 1) * BUF;
    int tap, cc;
    k_tile_split(p, t, kct, tap, cc);
    const int c0 = cc * KT;
    if ((kind & 1) == 0) {  // weights
      const int nh = kind >> 1;
      u16* dst = buf + (nh ? OFF_B1 : OFF_B0);
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const unsigned off = b_off[nh][i] + (unsigned)(tap * kc + c0 + b_chunk[i] * 8) * 2u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(dst + (wid * BI + i) * 8 * KT), 16, off, 0, 0, 0);
      }
    } else {
      const int mh = kind >> 1;
      u16* dst = buf + (mh ? OFF_A1 : OFF_A0);
      const bool src2 = c0 >= p.c1;  // c1 % 64 == 0: a K tile never straddles the two sources
      const int cs = src2 ? p.c2 : p.c1;
      const int cbase = src2 ? c0 - p.c1 : c0;
      if (MODE == 1) {
        const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int hi = a_ho[mh][i] * p.stride + kh - p.pad_lo;
          const int wi = a_wo[mh][i] * p.stride + kw - p.pad_lo;
          const bool ok = a_ok[mh][i] && hi >= 0 && wi >= 0 && hi < (p.hin << p.ups) && wi < (p.win << p.ups);
          const int pix = (a_img[mh][i] * p.hin + (hi >> p.ups)) * p.win + (wi >> p.ups);
          const unsigned off = ok ? ((unsigned)pix * (unsigned)cs + (unsigned)(cbase + a_chunk[i] * 8)) * 2u : DMA_OOB;
          void* d = dst + (wid * 2 + i) * 8 * KT;
          if (src2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (__attribute__((address_space(3))) void*)d, 16, off, 0, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)d, 16, off, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned off = (src2 ? a_off2[mh][i] : a_off1[mh][i]) + (unsigned)(cbase + a_chunk[i] * 8) * 2u;
          void* d = dst + (wid * 2 + i) * 8 * KT;
          if (src2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (__attribute__((address_space(3))) void*)d, 16, off, 0, 0, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)d, 16, off, 0, 0, 0);
        }
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment addresses (element offsets inside a half-tile; the same for every K tile)
  int fa_off[4][2], fb_off[TNH][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int row = wr * 64 + i * 16 + l15;
      fa_off[i][s] = row * KT + (((s * 4 + g) ^ swz(row)) << 3);
    }
#pragma unroll
  for (int j = 0; j < TNH; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int row = wc * (BROWS / 4) + j * 16 + l15;
      fb_off[j][s] = row * KT + (((s * 4 + g) ^ swz(row)) << 3);
    }

  u32x4 fa[4][2], fb0[TNH][2], fb1[TNH][2];

  // ---- prologue: half-tiles 0..6; K tile 0 (the first four) must have landed before phase 1
#pragma unroll
  for (int h = 0; h < 7; ++h) issue(h >> 2, h & 3);
  if (nt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 1 + 2 * BI) : "memory");  // leaves h = 4..6 (B, A, B)
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0

#define CA_PP_MFMA(MH, FB, NH)                                                                         \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)     \
      _Pragma("unroll") for (int j_ = 0; j_ < TNH; ++j_) acc[(MH) * 4 + i_][(NH) * TNH + j_] =          \
          Elem<DT>::mfma(FB[j_][s_], fa[i_][s_], acc[(MH) * 4 + i_][(NH) * TNH + j_]);

  for (int t = 0; t < (p.dbg == 2 ? 0 : nt); ++t) {
    const u16* buf = smem + (t & 1) * BUF;
    // ---- phase 1: B[nh0] first (retired before the barrier: its region is re-staged in phase 2), then A[mh0]
#pragma unroll
    for (int j = 0; j < TNH; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) fb0[j][s] = ld16(buf + OFF_B0 + fb_off[j][s]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) fa[i][s] = ld16(buf + OFF_A0 + fa_off[i][s]);
    issue(t + 1, 3);  // A[mh1] of K tile t+1
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    CA_PP_MFMA(0, fb0, 0)
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 2
#pragma unroll
    for (int j = 0; j < TNH; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) fb1[j][s] = ld16(buf + OFF_B1 + fb_off[j][s]);
    issue(t + 2, 0);  // B[nh0] of K tile t+2 (this buffer)
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    CA_PP_MFMA(0, fb1, 1)
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 3
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 2; ++s) fa[i][s] = ld16(buf + OFF_A1 + fa_off[i][s]);
    issue(t + 2, 1);  // A[mh0] of K tile t+2
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    CA_PP_MFMA(1, fb1, 1)
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---- phase 4: no LDS reads; the counted wait that makes K tile t+1 readable from the next phase on
    issue(t + 2, 2);  // B[nh1] of K tile t+2
    if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * 1 + 2 * BI) : "memory");  // B, A, B of K tile t+2 in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    CA_PP_MFMA(1, fb0, 0)
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
#undef CA_PP_MFMA
  if (wr == 0) __builtin_amdgcn_s_barrier();  // group 0 catches up: both groups have executed the same barriers
  __syncthreads();                             // (all DMA drained above: the last counted wait was vmcnt(0))

  if (p.dbg == 1) {
    if (acc[0][0][0] == 12345.678f) *reinterpret_cast<float*>(p.c) = acc[3][1][2] + acc[7][TN - 1][1];
    return;
  }
  gemm_epilogue<DT, BM, BN, TM, TN, 512>(p, acc, smem, m0, n0, wr, wc, l15, g, tid);
}

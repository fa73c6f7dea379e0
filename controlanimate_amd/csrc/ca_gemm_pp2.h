// Ping-pong kernel, 128 x 320 block tile (every channel count of the SD1.5 UNet / ControlNet is a multiple of
// 320: the tile divides N exactly, and the N = 320 layers of the 64x64-latent level read their A operand once).
// 8 waves = 2 groups (wr: 64-row halves) x 4 (wc: 80-column quarters); per-wave output 64 x 80 = 4 x 5 MFMA tiles.
//
// A K tile (64 deep, 56 KB) is three DMA units: A (128 rows), B0 (columns 0..31 of every quarter: 128 rows),
// B1 (columns 32..79 of every quarter: 192 rows); two LDS buffers.  Two phases per K tile:
//     phase 1: read A (8 frags) + B0 (4)   -> 16 MFMAs  C[:, n-tiles 0..1]
//     phase 2: read B1 (6)                 -> 24 MFMAs  C[:, n-tiles 2..4]
//   phase = { ds_reads ; DMA issue ; lgkmcnt(0) ; vmcnt(7) ; s_barrier ; MFMAs ; s_barrier }
// The groups run one barrier apart (see ca_gemm_pp.h).  Reads are retired BEFORE the phase's first barrier, so a
// region may be re-staged in the very next phase: phase 2 of K tile t issues A, B0 of K tile t+2 (4 DMA per wave),
// phase 1 of K tile t+1 issues B1 of K tile t+2 (3 DMA).  Every phase waits vmcnt(7): the units issued in this and
// the previous phase stay in flight, the one issued two phases ago -- read in the NEXT phase -- has landed.
template <int DT, int MODE>
__global__ __launch_bounds__(512, 2) void k_gemm_pp2(GemmKParams p) {
  constexpr int BM = 128, BN = 320, KT = 64;
  constexpr int TM = 4, TN = 5;
  constexpr int A_ROWS = 128, B0_ROWS = 128, B1_ROWS = 192;
  constexpr int OFF_A = 0, OFF_B0 = A_ROWS * KT, OFF_B1 = (A_ROWS + B0_ROWS) * KT;
  constexpr int BUF = (A_ROWS + B0_ROWS + B1_ROWS) * KT;  // elements: 56 KB
  constexpr int RS_ELEMS = BM * (BN / 8) * 4;  // row-sum scratch of the epilogue (float2 per row and 16-byte chunk), in u16 units
  constexpr int SMEM_ELEMS = 2 * BUF > BM * (BN + 8) + RS_ELEMS ? 2 * BUF : BM * (BN + 8) + RS_ELEMS;
  __shared__ __attribute__((aligned(16))) u16 smem[SMEM_ELEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid >> 2, wc = wid & 3;
  const int g = lane >> 4, l15 = lane & 15;

  const int tiles_n = p.n / BN;
  const int tiles_m = (p.m + BM - 1) / BM;
  unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  int split = 0;
  if (p.splits > 1) {  // split-K (small-M convolutions): consecutive ids = the tiles of ONE K range; fp32 slab instead of the epilogue
    split = bid / (unsigned)(tiles_m * tiles_n);
    bid -= split * (unsigned)(tiles_m * tiles_n);
  }
  int tile_m, tile_n;
  tile_coords(bid, tiles_m, tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a2 ? p.a2 : p.a), 0, p.a2 ? p.a2_bytes : p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  auto swz = [](int row) { return (row >> 1) & 7; };
  const int r8 = lane >> 3, cp = lane & 7;
  const int kc = p.c1 + p.c2;
  const int kct = kc / KT;
  const unsigned wld = (unsigned)(p.taps * kc);

  // ---- DMA source state.  A: local row = tile row, wave stages rows (wid*2 + i)*8 + r8.
  int a_chunk[2], a_img[2], a_ho[2], a_wo[2];
  unsigned a_off1[2], a_off2[2];
  bool a_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wid * 2 + i) * 8 + r8;
    a_chunk[i] = cp ^ swz(r);
    const int m = m0 + r;
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
  // B0: local row r -> column (r/32)*80 + r%32 ; B1: local row r -> column (r/48)*80 + 32 + r%48
  int b0_chunk[2], b1_chunk[3];
  unsigned b0_off[2], b1_off[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wid * 2 + i) * 8 + r8;
    b0_chunk[i] = cp ^ swz(r);
    b0_off[i] = (unsigned)(n0 + (r >> 5) * 80 + (r & 31)) * wld * 2u;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int r = (wid * 3 + i) * 8 + r8;
    b1_chunk[i] = cp ^ swz(r);
    b1_off[i] = (unsigned)(n0 + (r / 48) * 80 + 32 + r % 48) * wld * 2u;
  }

  const int nt_all = p.taps * kct;
  const int t_first = p.splits > 1 ? (int)((int64_t)nt_all * split / p.splits) : 0;
  const int nt = (p.splits > 1 ? (int)((int64_t)nt_all * (split + 1) / p.splits) : nt_all) - t_first;  // K tiles of this block: local index 0..nt-1

  auto issue_ab0 = [&](int t) {  // units A and B0 of K tile t: 4 DMA instructions
    if (t >= nt) return;
    u16* buf = smem + (t & 1) * BUF;
    int tap, cc;
    k_tile_split(p, t_first + t, kct, tap, cc);
    const int c0 = cc * KT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned off = b0_off[i] + (unsigned)(tap * kc + c0 + b0_chunk[i] * 8) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_B0 + (wid * 2 + i) * 8 * KT), 16, off, 0, 0, 0);
    }
    const bool src2 = c0 >= p.c1;
    const int cs = src2 ? p.c2 : p.c1;
    const int cbase = src2 ? c0 - p.c1 : c0;
    const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      unsigned off;
      if (MODE == 1) {
        const int hi = a_ho[i] * p.stride + kh - p.pad_lo;
        const int wi = a_wo[i] * p.stride + kw - p.pad_lo;
        const bool ok = a_ok[i] && hi >= 0 && wi >= 0 && hi < (p.hin << p.ups) && wi < (p.win << p.ups);
        const int pix = (a_img[i] * p.hin + (hi >> p.ups)) * p.win + (wi >> p.ups);
        off = ok ? ((unsigned)pix * (unsigned)cs + (unsigned)(cbase + a_chunk[i] * 8)) * 2u : DMA_OOB;
      } else {
        off = (src2 ? a_off2[i] : a_off1[i]) + (unsigned)(cbase + a_chunk[i] * 8) * 2u;
      }
      void* d = buf + OFF_A + (wid * 2 + i) * 8 * KT;
      if (src2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a2, (__attribute__((address_space(3))) void*)d, 16, off, 0, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)d, 16, off, 0, 0, 0);
    }
  };
  auto issue_b1 = [&](int t) {  // unit B1 of K tile t: 3 DMA instructions
    if (t >= nt) return;
    u16* buf = smem + (t & 1) * BUF;
    int tap, cc;
    k_tile_split(p, t_first + t, kct, tap, cc);
    const int c0 = cc * KT;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const unsigned off = b1_off[i] + (unsigned)(tap * kc + c0 + b1_chunk[i] * 8) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_B1 + (wid * 3 + i) * 8 * KT), 16, off, 0, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int fa_off[4][2], fb0_off[2][2], fb1_off[3][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = wr * 64 + i * 16 + l15;
      fa_off[i][s] = OFF_A + row * KT + (((s * 4 + g) ^ swz(row)) << 3);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wc * 32 + j * 16 + l15;
      fb0_off[j][s] = OFF_B0 + row * KT + (((s * 4 + g) ^ swz(row)) << 3);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int row = wc * 48 + j * 16 + l15;
      fb1_off[j][s] = OFF_B1 + row * KT + (((s * 4 + g) ^ swz(row)) << 3);
    }
  }
  u32x4 fa[4][2], fb0[2][2], fb1[3][2];

  // ---- prologue: K tile 0 complete, then A/B0 of K tile 1 in flight
  issue_ab0(0);
  issue_b1(0);
  issue_ab0(1);
  if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0

  for (int t = 0; t < (p.dbg == 2 ? 0 : nt); ++t) {
    const u16* buf = smem + (t & 1) * BUF;
    // ---- phase 1: A + B0 of K tile t; issue B1 of K tile t+1
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i][s] = ld16(buf + fa_off[i][s]);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb0[j][s] = ld16(buf + fb0_off[j][s]);
    }
    issue_b1(t + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // in flight afterwards: B1(t+1) [3, just issued] and A/B0(t+1) [4]; landed: B1(t) -- read in phase 2
    if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
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
    __builtin_amdgcn_s_barrier();
    // ---- phase 2: B1 of K tile t; issue A/B0 of K tile t+2 (this buffer: A/B0 were retired in phase 1)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 3; ++j) fb1[j][s] = ld16(buf + fb1_off[j][s]);
    issue_ab0(t + 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // in flight afterwards: A/B0(t+2) [4] and B1(t+1) [3]; landed: A/B0(t+1) -- read in the next phase
    if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
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
    __builtin_amdgcn_s_barrier();
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  __syncthreads();
  if (p.dbg == 1) {
    if (acc[0][0][0] == 12345.678f) *reinterpret_cast<float*>(p.c) = acc[3][1][2] + acc[2][TN - 1][1];
    return;
  }
  if (p.splits > 1) {  // raw fp32 slab; lane holds C[m = .. + l15][n = .. + 4g + (0..3)]
    float* slab = p.partial + (int64_t)split * p.m * p.n;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + wr * 64 + i * 16 + l15;
      if (m >= p.m) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(slab + (int64_t)m * p.n + n0 + wc * 80 + j * 16 + g * 4) = acc[i][j];
    }
    return;
  }
  gemm_epilogue<DT, BM, BN, TM, TN, 512, true>(p, acc, smem, m0, n0, wr, wc, l15, g, tid, reinterpret_cast<float2*>(smem + BM * (BN + 8)));
}

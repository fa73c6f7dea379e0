// The motion module's temporal self-attention of the 64x64-latent level in ONE launch (round 4):
//   o = softmax(q k^T / sqrt(d)) v  per (pixel, head) over the 16 frames,  q|k|v = (LayerNorm(x) + pe[frame]) Wqkv^T
// (reference: animatediff/models/motion_module.py:251-331 VersatileAttention.forward -- norm, positional encoding, to_q / to_k /
// to_v, attention over the frame axis; the output projection + residual stays a ca_gemm call).
//
// Why: as separate launches the 131072 x 960 q|k|v tensor is written (252 MB: the activation-resident GEMM cannot hide that store
// phase, 125 us) and read back by an attention kernel whose whole arithmetic is 16 x 16 scores per (pixel, head) (125 us of strided
// 80-byte pieces).  The sequence a (pixel, head) attends over is 16 rows -- ONE MFMA row tile -- so a block that holds the 16 frames
// of 8 pixels as its 128-row tile can finish the attention in registers and write only o (84 MB).
//
// Layout of the work:
//   * tile = 8 consecutive pixels x 16 frames: LDS row 16 i + f = row ((b 16 + f) hw + pix0 + i) of x (rows are in (b f n) order),
//     gathered by LDS-DMA into the activation-resident layout of ca_gemm_ar.h (80 KB, two blocks of four waves per CU);
//   * LayerNorm + positional encoding are applied to the tile IN PLACE: x' = gamma n(x) + (beta + pe[f]), rounded to the
//     activation type as the reference's `norm(x) + pe` is, so the projection runs on the ORIGINAL Wq / Wk / Wv, has no epilogue
//     operands at all and its accumulators ARE q, k, v;
//   * wave w owns heads w and w + 4; per head three passes of the activation-resident K loop (8 row tiles x 3 column tiles of 16:
//     d = 40 padded to 48 with zero weight rows, 96 accumulators, W fragments straight from L2 out of a per-wave contiguous
//     fragment-ordered stream, two chunks ahead):
//       q, k:  acc = mfma(W fragment, x fragment)  -> a lane holds 4 consecutive d of frame lane & 15: packed, that IS the
//              16x16x16 MFMA operand of S^T[key][query] = sum_d k[key][d] q[query][d]   (3 small MFMAs per pixel);
//       v:     acc = mfma(x fragment, W fragment)  -> a lane holds 4 consecutive frames (keys) of d_v = lane & 15: packed, the A
//              operand of O^T[d_v][query] = sum_key v[key][d_v] P^T[key][query]; P^T is S^T's own accumulator layout.
//     No value crosses a lane except the softmax's max / sum over the four 16-lane groups (two xor-shuffles each).
//   * o leaves as 8-byte pieces (4 consecutive d_v of one frame), 84 MB per launch.
//
// Fragment-ordered weights (ca_pack_w_tattn): element e of lane L's 16 bytes of column tile j of 32-deep chunk kq of pass ps
// (0 = q, 1 = k, 2 = v) of head h = w + 4 hi is W_ps[h 40 + 16 j + (L & 15)][32 kq + 8 (L >> 4) + e] (0 where 16 j + (L & 15) >= 40)
// at (((((w 2 + hi) 3 + ps) 10 + kq) 3 + j) 64 + L) 8 + e: the 60 chunks of a wave are one contiguous 180 KB stream.
template <int N>
struct IntC {
  static constexpr int value = N;
};

struct TattnParams {
  const u16* x;
  const u16* wf;
  const float* gamma;  // [320]
  const float* bp;     // [16][ld_bp]: beta + pe[frame]
  u16* o;
  int lda, ldo, ld_bp;
  int hw;              // pixels per image; groups of 16 frames = batch * hw, a multiple of 8
  float ln_eps, scale_log2;
  unsigned x_bytes, o_bytes;
  int dbg;  // (-DCA_EXPERIMENTS timing switches: 1 no LayerNorm pass, 2 no stores, 8 no tile DMA after the first)
  AttnOutParams out;  // ABI v12: the output projection + bias + residual as the kernel's last stage (k_tattn_out)
};

constexpr int CA_TATTN_WF_ELEMS = 4 * 2 * 3 * 10 * 3 * 64 * 8;

// w: row-major [960, 320] = Wq | Wk | Wv rows
__global__ __launch_bounds__(256) void k_pack_w_tattn(const u16* __restrict__ w, u16* __restrict__ dst) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // one 16-byte piece each
  if (idx >= CA_TATTN_WF_ELEMS / 8) return;
  const int L = idx & 63;
  int t = idx >> 6;
  const int j = t % 3;
  t /= 3;
  const int kq = t % 10;
  t /= 10;
  const int ps = t % 3;
  t /= 3;
  const int hi = t & 1, wv = t >> 1;
  const int head = wv + 4 * hi;
  const int dd = 16 * j + (L & 15);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  st16(dst + (int64_t)idx * 8, dd < 40 ? ld16(w + (int64_t)(ps * 320 + head * 40 + dd) * 320 + kq * 32 + (L >> 4) * 8) : zero4);
}

#ifdef CA_EXPERIMENTS
// (shader-clock stamps of block 0, wave 0, its first two tiles -- tools/tattn_stamps.py; shares ca_ff_fused.h's buffer)
#define CA_TA_STAMP(TAG)                                                      \
  if (blockIdx.x == 0 && wid == 0 && lane == 0 && stamp_i < 120) {            \
    ca_ff_stamps[0][2 * stamp_i] = __builtin_readcyclecounter();              \
    ca_ff_stamps[0][2 * stamp_i + 1] = (TAG);                                 \
    ++stamp_i;                                                                \
  }
#else
#define CA_TA_STAMP(TAG)
#endif

template <int DT>
__global__ __launch_bounds__(256, 2) void k_tattn_fused(TattnParams p, int tiles) {
#ifdef CA_EXPERIMENTS
  int stamp_i = 0;
#endif
  constexpr int K = 320, KQ = 10, TM = 8, TJ = 3, BM = 128, HD = 40;
  constexpr int ROWB = K * 2;
  constexpr unsigned CHUNKB = TJ * 1024u;  // one 32-deep chunk of one pass: three 1 KB fragments
  __shared__ __attribute__((aligned(16))) unsigned char smem[BM * ROWB];
  static_assert(2 * BM * ROWB <= 160 * 1024, "two blocks per CU");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  constexpr unsigned OOB_V = 0x80000000u;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wf, 0, (unsigned)CA_TATTN_WF_ELEMS * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)p.o, 0, p.o_bytes, 0x00020000);

  // A fragment addresses: ca_gemm_ar.h
  const int f_sw = (l15 >> 1) & 7;
  const int fa_lane = l15 * ROWB + (((f_sw >> 2) << 2) + (g ^ (f_sw & 3))) * 16;
  int fa_b[2][2] = {{fa_lane, fa_lane ^ 64}, {fa_lane + 4 * 16 * ROWB, (fa_lane ^ 64) + 4 * 16 * ROWB}};
  asm volatile("" : "+v"(fa_b[0][0]), "+v"(fa_b[0][1]), "+v"(fa_b[1][0]), "+v"(fa_b[1][1]));

  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int g0 = tile * 8;
    const int bimg = g0 / p.hw, pix0 = g0 - bimg * p.hw;  // (hw % 8 == 0: the 8 pixels of a tile belong to one batch element)
    __syncthreads();  // every wave has finished its reads of the previous tile
    CA_TA_STAMP(0)
#ifdef CA_EXPERIMENTS
    if (!(p.dbg & 8) || tile == (int)blockIdx.x)
#endif
    {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
#pragma unroll
      for (int q = 0; q < 20; ++q) {
        const unsigned idx = (unsigned)((wid * 20 + q) * 64 + lane_o);
        const unsigned r = __umulhi(idx >> 3, 0xCCCCCCCDu) >> 2;  // idx / 40: LDS row = 16 * pixel + frame
        const unsigned cp = idx - r * 40u;
        const unsigned c = cp ^ ((r >> 1) & 7u);
        const unsigned grow = (unsigned)((bimg * 16 + (int)(r & 15u)) * p.hw + pix0) + (r >> 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + (wid * 20 + q) * 1024), 16,
                                                 grow * (unsigned)p.lda * 2u + c * 16u, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CA_TA_STAMP(1)
#ifdef CA_EXPERIMENTS
    if (!(p.dbg & 1))
#endif
    {  // LayerNorm + positional encoding in place: two threads per row, 20 pieces each
      const int r = tid >> 1, h = tid & 1;
      const int fs = (r >> 1) & 7;
      unsigned char* src = smem + r * ROWB + h * 320;
      const float* bprow = p.bp + (int64_t)(r & 15) * p.ld_bp;
      float s = 0.f, ss = 0.f;
#pragma unroll 4
      for (int q = 0; q < 20; ++q) {
        const u32x4 v = ld16(src + q * 16);
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
      const float nb = -mean * rstd;
      // (rolled: fully unrolled, hipcc hoists all 80 table loads above the arithmetic and spills them; five pieces = 20 loads in flight)
#pragma unroll 5
      for (int q = 0; q < 20; ++q) {
        u32x4 v = ld16(src + q * 16);
        const int col = (((h * 20 + q) ^ fs)) * 8;  // logical chunk of physical piece h * 20 + q
        const float4 g0v = *reinterpret_cast<const float4*>(p.gamma + col), g1v = *reinterpret_cast<const float4*>(p.gamma + col + 4);
        const float4 b0v = *reinterpret_cast<const float4*>(bprow + col), b1v = *reinterpret_cast<const float4*>(bprow + col + 4);
        const float gm[8] = {g0v.x, g0v.y, g0v.z, g0v.w, g1v.x, g1v.y, g1v.z, g1v.w};
        const float bb[8] = {b0v.x, b0v.y, b0v.z, b0v.w, b1v.x, b1v.y, b1v.z, b1v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = Elem<DT>::to_f((u16)(v[e] & 0xffffu)), a1 = Elem<DT>::to_f((u16)(v[e] >> 16));
          v[e] = pack2<DT>(fmaf(fmaf(a0, rstd, nb), gm[2 * e], bb[2 * e]), fmaf(fmaf(a1, rstd, nb), gm[2 * e + 1], bb[2 * e + 1]));
        }
        st16(src + q * 16, v);
      }
    }
    __syncthreads();
    CA_TA_STAMP(2)

    int lane_k = lane;
    asm volatile("" : "+v"(lane_k));
    const unsigned wv = (unsigned)lane_k * 16u;
    const unsigned wwave = (unsigned)wid * (60u * CHUNKB);
    u32x4 fa[TM], fb[2][TJ];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < TJ; ++j) fb[c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, wwave + (unsigned)(c * TJ + j) * 1024u, 0));
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = ld16(smem + fa_b[i >> 2][0] + (i & 3) * 16 * ROWB);
    __builtin_amdgcn_sched_barrier(0);

    for (int hi = 0; hi < 2; ++hi) {
      const unsigned whead = wwave + (unsigned)hi * (30u * CHUNKB);
      // one pass of the activation-resident K loop (ca_gemm_ar.h): j-major, the W fragment of column tile j refilled two chunks
      // ahead behind its eight MFMAs, the next chunk's A fragments behind the MFMAs of the last column tile.  The stream runs on
      // across the passes (and into the wave's second head): chunk 10 of pass ps is chunk 0 of pass ps + 1.
      auto kloop = [&](auto pass_c, f32x4(&acc)[TM][TJ]) __attribute__((always_inline)) {
        constexpr int PS = decltype(pass_c)::value;
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
          const int nk = (kq + 1) % KQ;
          const int fa_off = (nk >> 1) * 128;
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              const f32x4 c0 = kq == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i][j];
              if (PS == 2) acc[i][j] = Elem<DT>::mfma(fa[i], fb[kq & 1][j], c0);
              else acc[i][j] = Elem<DT>::mfma(fb[kq & 1][j], fa[i], c0);
              if (j == TJ - 1) {
                __builtin_amdgcn_sched_barrier(0);
                fa[i] = ld16(smem + fa_b[i >> 2][nk & 1] + fa_off + (i & 3) * 16 * ROWB);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            const int sn = PS * 10 + kq + 2;  // stream position (chunks of this head) of the refill
            if (sn < 30) fb[kq & 1][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, whead + (unsigned)(sn * TJ + j) * 1024u, 0));
            else if (hi == 0) fb[kq & 1][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, whead + (unsigned)(sn * TJ + j) * 1024u, 0));
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      };

      u32x2 qp[TM][TJ];
      {
        f32x4 acc[TM][TJ];
        kloop(IntC<0>{}, acc);
        CA_TA_STAMP(3)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            qp[i][j] = (u32x2){pack2<DT>(acc[i][j][0] * p.scale_log2, acc[i][j][1] * p.scale_log2), pack2<DT>(acc[i][j][2] * p.scale_log2, acc[i][j][3] * p.scale_log2)};
      }
      CA_TA_STAMP(4)
      u32x2 pp[TM];
      {
        f32x4 acc[TM][TJ];
        kloop(IntC<1>{}, acc);
        CA_TA_STAMP(5)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          f32x4 st = {0.f, 0.f, 0.f, 0.f};  // S^T[key = 4 g + r][query = l15] in the exp2 domain
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            const u32x2 kp = {pack2<DT>(acc[i][j][0], acc[i][j][1]), pack2<DT>(acc[i][j][2], acc[i][j][3])};
            st = Elem<DT>::mfma16(kp, qp[i][j], st);
          }
          const float m = rowgroup_max(fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])));
          float e[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(st[r] - m);
          const float inv = __builtin_amdgcn_rcpf(rowgroup_sum((e[0] + e[1]) + (e[2] + e[3])));
          pp[i] = (u32x2){pack2<DT>(e[0] * inv, e[1] * inv), pack2<DT>(e[2] * inv, e[3] * inv)};
        }
      }
      {
        f32x4 acc[TM][TJ];
        CA_TA_STAMP(6)
        kloop(IntC<2>{}, acc);
        CA_TA_STAMP(7)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int l15e = lane_e & 15, ge = lane_e >> 4;
        const int head = wid + 4 * hi;
        const unsigned orow = (unsigned)((bimg * 16 + l15e) * p.hw + pix0) * (unsigned)p.ldo * 2u;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            const u32x2 vp = {pack2<DT>(acc[i][j][0], acc[i][j][1]), pack2<DT>(acc[i][j][2], acc[i][j][3])};
            const f32x4 ot = Elem<DT>::mfma16(vp, pp[i], (f32x4){0.f, 0.f, 0.f, 0.f});  // O^T[d_v = 16 j + 4 g + r][query = l15]
            const int dv = 16 * j + 4 * ge;
#ifdef CA_EXPERIMENTS
            if (p.dbg & 2) {
              asm volatile("" ::"v"(ot[0]), "v"(ot[1]), "v"(ot[2]), "v"(ot[3]));
              continue;
            }
#endif
            const unsigned off = dv < HD ? orow + (unsigned)(head * HD + dv) * 2u : OOB_V;
            __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack2<DT>(ot[0], ot[1]), pack2<DT>(ot[2], ot[3])}, rs_o, off, (unsigned)(i * p.ldo * 2), 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        CA_TA_STAMP(8)
      }
    }
  }
}


// ---- round 5 (ABI v12): the same attention WITH its output projection, bias and residual (ca_attn_out.h) ---------------------------
//   y = softmax(q k^T / sqrt(d)) v Wout^T + b_out + x        (motion_module.py:212-224: `attention_block(norm(x)) + x`)
// One block of EIGHT waves per CU (wave w = head w: one head each instead of two), the tile in one of two 80 KB buffers:
//   top     x(t) has landed in buf[t & 1]                                                        barrier
//   LN      LayerNorm + positional encoding in place, four threads per row                        barrier
//   attn    waves 0..3 issue the DMA of tile t + 1 into the OTHER buffer (their first W refill then waits for it -- loads return in
//           order -- while waves 4..7, the second wave of each SIMD, have the matrix pipes to themselves: the stall is covered);
//           three passes of the activation-resident K loop + the attention, exactly as k_tattn_fused; o stays in registers (48)
//           Wout chunks 0, 1 and the residual rows are requested                                  barrier (every wave has read x')
//   o       o -> buf[t & 1] in place of x' (8-byte pieces, the tile's swizzle)                    barrier
//   out     y = o Wout^T + b + residual, 64 x 80 per wave (ca_attn_out.h), stores
// o never reaches HBM (84 MB written + read per launch) and the k_gemm_wres launch behind it is gone.
// FR = frames per sequence: 16 (one MFMA row tile per pixel, 8 pixels per tile), 32 (two row tiles per pixel, 4 pixels: S^T and
// O^T in 2 x 2 / 2 blocks of 16 x 16) or 8 (two pixels per row tile, 16 pixels: the cross-pixel quarter of S^T masked out).
template <int DT, int FR>
__global__ __launch_bounds__(512, 2) void k_tattn_out(TattnParams p, int tiles) {
  static_assert(FR == 8 || FR == 16 || FR == 32, "frames per sequence");
  constexpr int PXT = 128 / FR;  // pixels per 128-row tile; LDS row = FR * pixel + frame
  constexpr int K = 320, KQ = 10, TM = 8, TJ = 3, BM = 128, HD = 40;
  constexpr int ROWB = K * 2, TILEB = BM * ROWB;
  constexpr unsigned CHUNKB = TJ * 1024u;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILEB];
  static_assert(2 * TILEB <= 160 * 1024, "two tile buffers");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wf, 0, (unsigned)CA_TATTN_WF_ELEMS * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)p.o, 0, p.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wo = __builtin_amdgcn_make_buffer_rsrc((void*)p.out.wof, 0, (unsigned)CA_WOUT_ELEMS * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out.bias ? (const void*)p.out.bias : (const void*)p.wf), 0, p.out.bias ? 320u * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out.res ? (const void*)p.out.res : (const void*)p.wf), 0, p.out.res ? p.out.res_bytes : 0u, 0x00020000);

  const int f_sw = (l15 >> 1) & 7;
  const int fa_lane = l15 * ROWB + (((f_sw >> 2) << 2) + (g ^ (f_sw & 3))) * 16;
  int fa_b[2][2] = {{fa_lane, fa_lane ^ 64}, {fa_lane + 4 * 16 * ROWB, (fa_lane ^ 64) + 4 * 16 * ROWB}};
  asm volatile("" : "+v"(fa_b[0][0]), "+v"(fa_b[0][1]), "+v"(fa_b[1][0]), "+v"(fa_b[1][1]));

  // tile DMA by waves 0..3, 20 wave instructions (1 KB of the tile image) each
  auto issue_tile = [&](int tile, int bufsel) __attribute__((always_inline)) {
    if (wid >= 4) return;
    const int g0 = tile * PXT;
    const int bimg = g0 / p.hw, pix0 = g0 - bimg * p.hw;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
#pragma unroll
    for (int q = 0; q < 20; ++q) {
      const unsigned idx = (unsigned)((wid * 20 + q) * 64 + lane_o);
      const unsigned r = __umulhi(idx >> 3, 0xCCCCCCCDu) >> 2;  // idx / 40: LDS row = FR * pixel + frame
      const unsigned cp = idx - r * 40u;
      const unsigned c = cp ^ ((r >> 1) & 7u);
      const unsigned grow = (unsigned)((bimg * FR + (int)(r & (unsigned)(FR - 1))) * p.hw + pix0) + r / (unsigned)FR;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + bufsel * TILEB + (wid * 20 + q) * 1024), 16,
                                               grow * (unsigned)p.lda * 2u + c * 16u, 0, 0, 0);
    }
  };

  if ((int)blockIdx.x < tiles) issue_tile(blockIdx.x, 0);
  int it = 0;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x, ++it) {
    const int cur = it & 1;
    unsigned char* const xb = smem + cur * TILEB;
    const int g0 = tile * PXT;
    const int bimg = g0 / p.hw, pix0 = g0 - bimg * p.hw;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // x(tile) has landed; every wave is past the previous tile's output stage
    {  // LayerNorm + positional encoding in place: four threads per row, 10 pieces each
      const int r = tid >> 2, h = tid & 3;
      const int fs = (r >> 1) & 7;
      unsigned char* src = xb + r * ROWB + h * 160;
      const float* bprow = p.bp + (int64_t)(r & (FR - 1)) * p.ld_bp;
      float s = 0.f, ss = 0.f;
#pragma unroll 5
      for (int q = 0; q < 10; ++q) {
        const u32x4 v = ld16(src + q * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = Elem<DT>::to_f((u16)(v[e] & 0xffffu)), a1 = Elem<DT>::to_f((u16)(v[e] >> 16));
          s += a0 + a1;
          ss = fmaf(a0, a0, fmaf(a1, a1, ss));
        }
      }
      s += __shfl_xor(s, 1);
      ss += __shfl_xor(ss, 1);
      s += __shfl_xor(s, 2);
      ss += __shfl_xor(ss, 2);
      const float mean = s * (1.f / K);
      const float rstd = rsqrtf(fmaxf(ss * (1.f / K) - mean * mean, 0.f) + p.ln_eps);
      const float nb = -mean * rstd;
#pragma unroll 5
      for (int q = 0; q < 10; ++q) {
        u32x4 v = ld16(src + q * 16);
        const int col = (((h * 10 + q) ^ fs)) * 8;  // logical chunk of physical piece h * 10 + q
        const float4 g0v = *reinterpret_cast<const float4*>(p.gamma + col), g1v = *reinterpret_cast<const float4*>(p.gamma + col + 4);
        const float4 b0v = *reinterpret_cast<const float4*>(bprow + col), b1v = *reinterpret_cast<const float4*>(bprow + col + 4);
        const float gm[8] = {g0v.x, g0v.y, g0v.z, g0v.w, g1v.x, g1v.y, g1v.z, g1v.w};
        const float bb[8] = {b0v.x, b0v.y, b0v.z, b0v.w, b1v.x, b1v.y, b1v.z, b1v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = Elem<DT>::to_f((u16)(v[e] & 0xffffu)), a1 = Elem<DT>::to_f((u16)(v[e] >> 16));
          v[e] = pack2<DT>(fmaf(fmaf(a0, rstd, nb), gm[2 * e], bb[2 * e]), fmaf(fmaf(a1, rstd, nb), gm[2 * e + 1], bb[2 * e + 1]));
        }
        st16(src + q * 16, v);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the table loads of the pass: nothing of this wave is in flight when the DMA is queued)
    __syncthreads();

    int lane_k = lane;
    asm volatile("" : "+v"(lane_k));
    const unsigned wv = (unsigned)lane_k * 16u;
    const unsigned whead = (unsigned)((wid & 3) * 2 + (wid >> 2)) * (30u * CHUNKB);  // head = wid: ca_pack_w_tattn keeps heads w, w + 4 adjacent
    u32x4 fa[TM], fb[2][TJ];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < TJ; ++j) fb[c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, whead + (unsigned)(c * TJ + j) * 1024u, 0));
    {
      const int next = tile + (int)gridDim.x;
      if (next < tiles) issue_tile(next, cur ^ 1);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = ld16(xb + fa_b[i >> 2][0] + (i & 3) * 16 * ROWB);
    __builtin_amdgcn_sched_barrier(0);

    auto kloop = [&](auto pass_c, f32x4(&acc)[TM][TJ]) __attribute__((always_inline)) {
      constexpr int PS = decltype(pass_c)::value;
#pragma unroll
      for (int kq = 0; kq < KQ; ++kq) {
        const int nk = (kq + 1) % KQ;
        const int fa_off = (nk >> 1) * 128;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const f32x4 c0 = kq == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i][j];
            if (PS == 2) acc[i][j] = Elem<DT>::mfma(fa[i], fb[kq & 1][j], c0);
            else acc[i][j] = Elem<DT>::mfma(fb[kq & 1][j], fa[i], c0);
            if (j == TJ - 1) {
              __builtin_amdgcn_sched_barrier(0);
              fa[i] = ld16(xb + fa_b[i >> 2][nk & 1] + fa_off + (i & 3) * 16 * ROWB);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          const int sn = PS * 10 + kq + 2;  // stream position (chunks of this head) of the refill
          if (sn < 30) fb[kq & 1][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, whead + (unsigned)(sn * TJ + j) * 1024u, 0));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };

    u32x2 qp[TM][TJ];
    {
      f32x4 acc[TM][TJ];
      kloop(IntC<0>{}, acc);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
          qp[i][j] = (u32x2){pack2<DT>(acc[i][j][0] * p.scale_log2, acc[i][j][1] * p.scale_log2), pack2<DT>(acc[i][j][2] * p.scale_log2, acc[i][j][3] * p.scale_log2)};
    }
    // P^T as the B operand of O^T = V^T P^T.  FR <= 16: pp[i][0] for row tile i; FR = 32: pp[2 px + kt][qt] for the (key tile, query tile) block
    u32x2 pp[TM][FR == 32 ? 2 : 1];
    {
      f32x4 acc[TM][TJ];
      kloop(IntC<1>{}, acc);
      int lane_s = lane;
      asm volatile("" : "+v"(lane_s));
      const int l15s = lane_s & 15, gs = lane_s >> 4;
      if constexpr (FR != 32) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          f32x4 st = {0.f, 0.f, 0.f, 0.f};  // S^T[key = 4 g + r][query = l15] in the exp2 domain
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            const u32x2 kp = {pack2<DT>(acc[i][j][0], acc[i][j][1]), pack2<DT>(acc[i][j][2], acc[i][j][3])};
            st = Elem<DT>::mfma16(kp, qp[i][j], st);
          }
          if constexpr (FR == 8) {  // two pixels share the row tile: a key of the other pixel is not a key
            if ((gs >> 1) != (l15s >> 3)) st = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          }
          const float m = rowgroup_max(fmaxf(fmaxf(st[0], st[1]), fmaxf(st[2], st[3])));
          float e[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(st[r] - m);
          const float inv = __builtin_amdgcn_rcpf(rowgroup_sum((e[0] + e[1]) + (e[2] + e[3])));
          pp[i][0] = (u32x2){pack2<DT>(e[0] * inv, e[1] * inv), pack2<DT>(e[2] * inv, e[3] * inv)};
        }
      } else {
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          u32x2 kp[2][TJ];
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
              kp[kt][j] = (u32x2){pack2<DT>(acc[2 * px + kt][j][0], acc[2 * px + kt][j][1]), pack2<DT>(acc[2 * px + kt][j][2], acc[2 * px + kt][j][3])};
#pragma unroll
          for (int qt = 0; qt < 2; ++qt) {  // queries: frames 16 qt + l15; keys: frames 16 kt + 4 g + r
            f32x4 st[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
              st[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int j = 0; j < TJ; ++j) st[kt] = Elem<DT>::mfma16(kp[kt][j], qp[2 * px + qt][j], st[kt]);
            }
            const float m = rowgroup_max(fmaxf(fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3])), fmaxf(fmaxf(st[1][0], st[1][1]), fmaxf(st[1][2], st[1][3]))));
            float e[2][4], l = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                e[kt][r] = __builtin_amdgcn_exp2f(st[kt][r] - m);
                l += e[kt][r];
              }
            const float inv = __builtin_amdgcn_rcpf(rowgroup_sum(l));
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) pp[2 * px + kt][qt] = (u32x2){pack2<DT>(e[kt][0] * inv, e[kt][1] * inv), pack2<DT>(e[kt][2] * inv, e[kt][3] * inv)};
          }
        }
      }
    }
    u32x2 op[TM][TJ];  // O^T[d_v = 16 j + 4 g + r][query = frame] of row tile i, rounded to the activation type
    {
      f32x4 acc[TM][TJ];
      kloop(IntC<2>{}, acc);
      if constexpr (FR != 32) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            const u32x2 vp = {pack2<DT>(acc[i][j][0], acc[i][j][1]), pack2<DT>(acc[i][j][2], acc[i][j][3])};
            const f32x4 ot = Elem<DT>::mfma16(vp, pp[i][0], (f32x4){0.f, 0.f, 0.f, 0.f});
            op[i][j] = (u32x2){pack2<DT>(ot[0], ot[1]), pack2<DT>(ot[2], ot[3])};
          }
      } else {
#pragma unroll
        for (int px = 0; px < 4; ++px)
#pragma unroll
          for (int j = 0; j < TJ; ++j) {
            u32x2 vp[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) vp[kt] = (u32x2){pack2<DT>(acc[2 * px + kt][j][0], acc[2 * px + kt][j][1]), pack2<DT>(acc[2 * px + kt][j][2], acc[2 * px + kt][j][3])};
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
              f32x4 ot = Elem<DT>::mfma16(vp[0], pp[2 * px][qt], (f32x4){0.f, 0.f, 0.f, 0.f});
              ot = Elem<DT>::mfma16(vp[1], pp[2 * px + 1][qt], ot);
              op[2 * px + qt][j] = (u32x2){pack2<DT>(ot[0], ot[1]), pack2<DT>(ot[2], ot[3])};
            }
          }
      }
    }
    // output stage: its first W chunks and the residual rows are requested before the barriers
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int l15e = lane_e & 15, ge = lane_e >> 4;
    // global row of LDS row 64 rh + 16 i + l15 (= FR * pixel + frame) = row_base + (i & 1) * step_a + (i >> 1) * step_b:
    //   FR 16: pixel 4 rh + i, frame l15;  FR 32: pixel 2 rh + (i >> 1), frame 16 (i & 1) + l15;  FR 8: pixel 8 rh + 2 i + (l15 >> 3), frame l15 & 7
    const int rh_ = wid >> 2;
    const unsigned row_base = FR == 16 ? (unsigned)((bimg * 16 + l15e) * p.hw + pix0 + 4 * rh_)
                                       : FR == 32 ? (unsigned)((bimg * 32 + l15e) * p.hw + pix0 + 2 * rh_) : (unsigned)((bimg * 8 + (l15e & 7)) * p.hw + pix0 + 8 * rh_ + (l15e >> 3));
    const unsigned step_a = FR == 16 ? 1u : FR == 32 ? 16u * (unsigned)p.hw : 2u, step_b = FR == 16 ? 2u : FR == 32 ? 1u : 4u;
    AttnOutRegs R;
    attn_out_prefetch<DT>(R, p.out, rs_wo, rs_res, wid, lane_e, row_base, step_a, step_b);
    __syncthreads();  // every wave has finished its K loops on x'
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int dv = 16 * j + 4 * ge;
        if (dv < HD) {
          const int col = wid * HD + dv;
          ca_lds_store8(xb + (16 * i + l15e) * ROWB + (((col >> 3) ^ ((l15e >> 1) & 7)) << 4) + (col & 7) * 2, op[i][j]);
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // the o tile is complete
    attn_out_run<DT>(R, xb, fa_b, p.out, rs_wo, rs_bo, rs_o, wid, lane_e, row_base, step_a, step_b, (unsigned)p.ldo);
  }
}

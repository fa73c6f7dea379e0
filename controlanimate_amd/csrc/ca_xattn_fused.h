// The text cross-attention of the 64x64-latent level up to its output projection in ONE launch (round 4):
//   o = softmax(q K^T / sqrt(d)) V per (row, head),  q = LayerNorm(x) Wq^T   (K, V: the 77 projected text tokens of the row's batch element)
// (reference: animatediff/models/attention.py:253-262 BasicTransformerBlock -- norm2, attn2: to_q, attention over encoder_hidden_states;
// the output projection + residual stays a ca_gemm call; K / V are projected once per window and cached, as before).
//
// As separate launches q is written and read back (2 x 84 MB) between a 131072x320x320 GEMM and an attention kernel that are both
// memory-side bound (56 + 50 us).  Here the block that holds 128 rows of x as its activation-resident tile (ca_gemm_ar.h / ca_tattn_fused.h:
// 80 KB, LDS-DMA, two blocks of four waves per CU) normalises them in place, and wave w computes for heads w and w + 4:
//   * the q pass of the activation-resident K loop in the transposed orientation, acc = mfma(W fragment, x fragment): 8 row tiles x 3
//     column tiles (d = 40 padded to 48 with zero weight rows), so that a lane holds 4 consecutive d of row lane & 15 -- packed, the B
//     operand of S^T[key][row] = sum_d K[key][d] q[row][d] on 16x16x16 MFMAs;
//   * per row tile: 5 key tiles x 3 d-chunks of S^T, one max / exp2 / sum over the lane's 20 scores, O^T[d_v][row] = sum_key
//     V^T[d_v][key] P^T[key][row] (P^T is S^T's own accumulator layout), 8-byte stores of 4 consecutive d_v.
// K and V reach the lanes as ready-made fragments: ca_xattn_pack_kv writes, once per window and layer, for every (text batch, head) the
// 15 K fragments (pre-multiplied by scale * log2 e, zero beyond key nk and d 40) and the 15 V^T fragments in lane order -- 30 coalesced
// 8-byte loads per head and tile, requested under the last chunks of the q pass.
//
// Fragment-ordered weights (ca_xattn_pack_w): as ca_tattn_fused.h with one pass: element e of lane L's 16 bytes of column tile j of chunk kq
// of head h = w + 4 hi is Wq[h 40 + 16 j + (L & 15)][32 kq + 8 (L >> 4) + e] (0 where 16 j + (L & 15) >= 40) at ((((w 2 + hi) 10 + kq) 3 + j) 64 + L) 8 + e.
struct XattnParams {
  const u16* x;
  const u16* wf;
  const float* bias;  // [320] or null
  const u16* kvf;     // [kv batch][head][30 fragments][64 lanes][4]
  u16* o;
  int lda, ldo;
  int m, tokens, frames_per_kv, kv_mod, nk;
  float ln_eps;
  unsigned x_bytes, o_bytes, kvf_bytes;
  AttnOutParams out;  // ABI v12: the output projection + bias + residual as the kernel's last stage (k_xattn_out)
  // ABI v13: the IP-Adapter's image-prompt tokens (k_xattn_out<DT, true>): a second K / V set in the same fragment format (its first key tile
  // only: nk_ip <= 16), its own softmax, o = o_text + ip_scale * o_ip before the output projection (modules/attention_processor.py:433-477)
  const u16* kvf_ip;
  int nk_ip;
  float ip_scale;
};

constexpr int CA_XATTN_WF_ELEMS = 4 * 2 * 10 * 3 * 64 * 8;  // 122880
constexpr int CA_XATTN_KVF_ELEMS = 30 * 64 * 4;             // per (text batch, head)

__global__ __launch_bounds__(256) void k_xattn_pack_w(const u16* __restrict__ w, u16* __restrict__ dst) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // one 16-byte piece each
  if (idx >= CA_XATTN_WF_ELEMS / 8) return;
  const int L = idx & 63;
  int t = idx >> 6;
  const int j = t % 3;
  t /= 3;
  const int kq = t % 10;
  t /= 10;
  const int hi = t & 1, wv = t >> 1;
  const int dd = 16 * j + (L & 15);
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  st16(dst + (int64_t)idx * 8, dd < 40 ? ld16(w + (int64_t)((wv + 4 * hi) * 40 + dd) * 320 + kq * 32 + (L >> 4) * 8) : zero4);
}

// kv: [kv_batches * rows_per_batch, ld] rows, K at column 0, V at column 320; keys row_offset .. row_offset + nk of every batch.
// grid = kv_batches * 8 blocks of 64 threads: lane L writes its 30 fragments (8 bytes each).
template <int DT>
__global__ __launch_bounds__(64) void k_xattn_pack_kv(const u16* __restrict__ kv, int64_t ld, int rows_per_batch, int row_offset, int nk, float scale_log2,
                                                      u16* __restrict__ dst) {
  const int head = blockIdx.x & 7, zk = blockIdx.x >> 3;
  const int L = threadIdx.x, g = L >> 4, l15 = L & 15;
  const u16* kp = kv + ((int64_t)zk * rows_per_batch + row_offset) * ld + head * 40;
  const u16* vp = kp + 320;
  u16* out = dst + ((int64_t)blockIdx.x * 30 * 64 + L) * 4;
#pragma unroll 1
  for (int f = 0; f < 15; ++f) {  // K fragment (kt, c): K[16 kt + l15][16 c + 4 g + r] * scale
    const int kt = f / 3, c = f - kt * 3;
    const int key = 16 * kt + l15, d0 = 16 * c + 4 * g;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (key < nk && d0 + r < 40) ? Elem<DT>::to_f(kp[(int64_t)key * ld + d0 + r]) * scale_log2 : 0.f;
    *reinterpret_cast<u32x2*>(out + (int64_t)f * 64 * 4) = (u32x2){pack2<DT>(v[0], v[1]), pack2<DT>(v[2], v[3])};
  }
#pragma unroll 1
  for (int f = 0; f < 15; ++f) {  // V^T fragment (j, kt): V[16 kt + 4 g + r][16 j + l15]
    const int j = f / 5, kt = f - j * 5;
    const int dv = 16 * j + l15, k0 = 16 * kt + 4 * g;
    u16 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = (k0 + r < nk && dv < 40) ? vp[(int64_t)(k0 + r) * ld + dv] : (u16)0;
    *reinterpret_cast<u32x2*>(out + (int64_t)(15 + f) * 64 * 4) = (u32x2){(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16)};
  }
}

template <int DT>
__global__ __launch_bounds__(256, 2) void k_xattn_fused(XattnParams p, int tiles) {
  constexpr int K = 320, KQ = 10, TM = 8, TJ = 3, BM = 128, HD = 40, KT = 5;
  constexpr int ROWB = K * 2;
  constexpr unsigned CHUNKB = TJ * 1024u;
  __shared__ __attribute__((aligned(16))) unsigned char smem[BM * ROWB];
  static_assert(2 * BM * ROWB <= 160 * 1024, "two blocks per CU");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  constexpr unsigned OOB_V = 0x80000000u;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wf, 0, (unsigned)CA_XATTN_WF_ELEMS * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)p.o, 0, p.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_kv = __builtin_amdgcn_make_buffer_rsrc((void*)p.kvf, 0, p.kvf_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)p.wf), 0, p.bias ? 320u * 4u : 0u, 0x00020000);

  const int f_sw = (l15 >> 1) & 7;
  const int fa_lane = l15 * ROWB + (((f_sw >> 2) << 2) + (g ^ (f_sw & 3))) * 16;
  int fa_b[2][2] = {{fa_lane, fa_lane ^ 64}, {fa_lane + 4 * 16 * ROWB, (fa_lane ^ 64) + 4 * 16 * ROWB}};
  asm volatile("" : "+v"(fa_b[0][0]), "+v"(fa_b[0][1]), "+v"(fa_b[1][0]), "+v"(fa_b[1][1]));

  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int m0 = tile * BM;
    const int zk = ((m0 / p.tokens) / p.frames_per_kv) % p.kv_mod;  // (tokens % 128 == 0: a tile belongs to one image)
    __syncthreads();  // every wave has finished its reads of the previous tile
    {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
#pragma unroll
      for (int q = 0; q < 20; ++q) {
        const unsigned idx = (unsigned)((wid * 20 + q) * 64 + lane_o);
        const unsigned r = __umulhi(idx >> 3, 0xCCCCCCCDu) >> 2;  // idx / 40
        const unsigned cp = idx - r * 40u;
        const unsigned c = cp ^ ((r >> 1) & 7u);
        const unsigned off = (m0 + (int)r) < p.m ? (unsigned)(m0 + (int)r) * (unsigned)p.lda * 2u + c * 16u : OOB_V;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + (wid * 20 + q) * 1024), 16, off, 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {  // LayerNorm in place (gamma / beta live in Wq and the bias): two threads per row, 20 pieces each
      const int r = tid >> 1, h = tid & 1;
      unsigned char* src = smem + r * ROWB + h * 320;
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
#pragma unroll 4
      for (int q = 0; q < 20; ++q) {
        u32x4 v = ld16(src + q * 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = Elem<DT>::to_f((u16)(v[e] & 0xffffu)), a1 = Elem<DT>::to_f((u16)(v[e] >> 16));
          v[e] = pack2<DT>(fmaf(a0, rstd, nb), fmaf(a1, rstd, nb));
        }
        st16(src + q * 16, v);
      }
    }
    __syncthreads();

    int lane_k = lane;
    asm volatile("" : "+v"(lane_k));
    const unsigned wv = (unsigned)lane_k * 16u;
    const unsigned wwave = (unsigned)wid * (20u * CHUNKB);
    u32x4 fa[TM], fb[2][TJ];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < TJ; ++j) fb[c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, wwave + (unsigned)(c * TJ + j) * 1024u, 0));
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = ld16(smem + fa_b[i >> 2][0] + (i & 3) * 16 * ROWB);
    __builtin_amdgcn_sched_barrier(0);

    for (int hi = 0; hi < 2; ++hi) {
      const int head = wid + 4 * hi;
      const unsigned whead = wwave + (unsigned)hi * (10u * CHUNKB);
      const unsigned kvb = (unsigned)(zk * 8 + head) * (unsigned)(CA_XATTN_KVF_ELEMS * 2) + (unsigned)lane_k * 8u;
      u32x2 kf[KT][TJ], vf[TJ][KT];
      f32x4 acc[TM][TJ];
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
            acc[i][j] = Elem<DT>::mfma(fb[kq & 1][j], fa[i], c0);
            if (j == TJ - 1) {
              __builtin_amdgcn_sched_barrier(0);
              fa[i] = ld16(smem + fa_b[i >> 2][nk & 1] + fa_off + (i & 3) * 16 * ROWB);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          const int sn = kq + 2;  // stream position (chunks of this head) of the refill; 10, 11 = the second head's first chunks
          if (sn < 10 || hi == 0) fb[kq & 1][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, whead + (unsigned)(sn * TJ + j) * 1024u, 0));
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kq == KQ - 2) {  // this head's K fragments, under the last two chunks
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int c = 0; c < TJ; ++c) kf[kt][c] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_kv, kvb, (unsigned)((kt * 3 + c) * 512), 0));
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int l15e = lane_e & 15, ge = lane_e >> 4;
      u32x2 qp[TM][TJ];
      {
        f32x4 bi[TJ];
        // (V^T fragments only now: beside 96 accumulators they did not fit -- hipcc spilled ten of them inside the K loop)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) vf[j][kt] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_kv, kvb, (unsigned)((15 + j * 5 + kt) * 512), 0));
#pragma unroll
        for (int j = 0; j < TJ; ++j) bi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bi, (unsigned)(head * HD + 16 * j + 4 * ge) * 4u, 0, 0));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            qp[i][j] = (u32x2){pack2<DT>(acc[i][j][0] + bi[j][0], acc[i][j][1] + bi[j][1]), pack2<DT>(acc[i][j][2] + bi[j][2], acc[i][j][3] + bi[j][3])};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        f32x4 st[KT];  // S^T[key = 16 kt + 4 g + r][row = l15] in the exp2 domain (K carries scale * log2 e)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          st[kt] = Elem<DT>::mfma16(kf[kt][0], qp[i][0], (f32x4){0.f, 0.f, 0.f, 0.f});
          st[kt] = Elem<DT>::mfma16(kf[kt][1], qp[i][1], st[kt]);
          st[kt] = Elem<DT>::mfma16(kf[kt][2], qp[i][2], st[kt]);
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kt >= 4 && 16 * kt + 4 * ge + r >= p.nk) st[kt][r] = -INFINITY;  // (64 < nk <= 80: only the last tile is ragged)
        float m = fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3]));
#pragma unroll
        for (int kt = 1; kt < KT; ++kt) m = fmaxf(m, fmaxf(fmaxf(st[kt][0], st[kt][1]), fmaxf(st[kt][2], st[kt][3])));
        m = rowgroup_max(m);
        float l = 0.f;
        u32x2 pp[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          float e[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            e[r] = __builtin_amdgcn_exp2f(st[kt][r] - m);
            l += e[r];
          }
          pp[kt] = (u32x2){pack2_prob<DT>(e[0], e[1]), pack2_prob<DT>(e[2], e[3])};
        }
        const float inv = __builtin_amdgcn_rcpf(rowgroup_sum(l));
        const int row = m0 + 16 * i + l15e;
        const unsigned ro = row < p.m ? (unsigned)row * (unsigned)p.ldo * 2u : OOB_V;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          f32x4 ot = Elem<DT>::mfma16(vf[j][0], pp[0], (f32x4){0.f, 0.f, 0.f, 0.f});  // O^T[d_v = 16 j + 4 g + r][row = l15]
#pragma unroll
          for (int kt = 1; kt < KT; ++kt) ot = Elem<DT>::mfma16(vf[j][kt], pp[kt], ot);
          const int dv = 16 * j + 4 * ge;
          __builtin_amdgcn_raw_buffer_store_b64((u32x2){pack2<DT>(ot[0] * inv, ot[1] * inv), pack2<DT>(ot[2] * inv, ot[3] * inv)}, rs_o,
                                                dv < HD ? ro + (unsigned)(head * HD + dv) * 2u : OOB_V, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}


// ---- round 5 (ABI v12): the same attention WITH its output projection, bias and residual (ca_attn_out.h) ---------------------------
//   y = softmax(q K^T / sqrt(d)) V Wout^T + b_out + x        (animatediff/models/attention.py:253-262: `attn2(norm2(x), ehs) + x`)
// The structure of k_tattn_out (ca_tattn_fused.h): one block of eight waves per CU, wave w = head w, the tile in one of two 80 KB
// buffers (the next tile's DMA issued by waves 0..3 at the start of the K loops), o through LDS into the 64 x 80-per-wave output stage.
template <int DT, bool IP>
__global__ __launch_bounds__(512, 2) void k_xattn_out(XattnParams p, int tiles) {
  constexpr int K = 320, KQ = 10, TM = 8, TJ = 3, BM = 128, HD = 40, KT = 5;
  constexpr int ROWB = K * 2, TILEB = BM * ROWB;
  constexpr unsigned CHUNKB = TJ * 1024u;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILEB];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;

  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wf, 0, (unsigned)CA_XATTN_WF_ELEMS * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)p.o, 0, p.o_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_kv = __builtin_amdgcn_make_buffer_rsrc((void*)p.kvf, 0, p.kvf_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bi = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)p.wf), 0, p.bias ? 320u * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_kvi = __builtin_amdgcn_make_buffer_rsrc((void*)(IP ? (const void*)p.kvf_ip : (const void*)p.kvf), 0, IP ? p.kvf_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wo = __builtin_amdgcn_make_buffer_rsrc((void*)p.out.wof, 0, (unsigned)CA_WOUT_ELEMS * 2u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bo = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out.bias ? (const void*)p.out.bias : (const void*)p.wf), 0, p.out.bias ? 320u * 4u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out.res ? (const void*)p.out.res : (const void*)p.wf), 0, p.out.res ? p.out.res_bytes : 0u, 0x00020000);

  const int f_sw = (l15 >> 1) & 7;
  const int fa_lane = l15 * ROWB + (((f_sw >> 2) << 2) + (g ^ (f_sw & 3))) * 16;
  int fa_b[2][2] = {{fa_lane, fa_lane ^ 64}, {fa_lane + 4 * 16 * ROWB, (fa_lane ^ 64) + 4 * 16 * ROWB}};
  asm volatile("" : "+v"(fa_b[0][0]), "+v"(fa_b[0][1]), "+v"(fa_b[1][0]), "+v"(fa_b[1][1]));

  auto issue_tile = [&](int tile, int bufsel) __attribute__((always_inline)) {
    if (wid >= 4) return;
    const int m0 = tile * BM;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
#pragma unroll
    for (int q = 0; q < 20; ++q) {
      const unsigned idx = (unsigned)((wid * 20 + q) * 64 + lane_o);
      const unsigned r = __umulhi(idx >> 3, 0xCCCCCCCDu) >> 2;  // idx / 40
      const unsigned cp = idx - r * 40u;
      const unsigned c = cp ^ ((r >> 1) & 7u);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + bufsel * TILEB + (wid * 20 + q) * 1024), 16,
                                               (unsigned)(m0 + (int)r) * (unsigned)p.lda * 2u + c * 16u, 0, 0, 0);  // (m % 128 == 0: every tile is whole)
    }
  };

  if ((int)blockIdx.x < tiles) issue_tile(blockIdx.x, 0);
  int it = 0;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x, ++it) {
    const int cur = it & 1;
    unsigned char* const xb = smem + cur * TILEB;
    const int m0 = tile * BM;
    const int zk = ((m0 / p.tokens) / p.frames_per_kv) % p.kv_mod;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {  // LayerNorm in place (gamma / beta live in Wq and the bias): four threads per row, 10 pieces each
      const int r = tid >> 2, h = tid & 3;
      unsigned char* src = xb + r * ROWB + h * 160;
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
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a0 = Elem<DT>::to_f((u16)(v[e] & 0xffffu)), a1 = Elem<DT>::to_f((u16)(v[e] >> 16));
          v[e] = pack2<DT>(fmaf(a0, rstd, nb), fmaf(a1, rstd, nb));
        }
        st16(src + q * 16, v);
      }
    }
    __syncthreads();

    int lane_k = lane;
    asm volatile("" : "+v"(lane_k));
    const unsigned wv = (unsigned)lane_k * 16u;
    const int head = wid;
    const unsigned whead = (unsigned)((wid & 3) * 2 + (wid >> 2)) * (10u * CHUNKB);  // ca_xattn_pack_w keeps heads w, w + 4 adjacent
    const unsigned kvb = (unsigned)(zk * 8 + head) * (unsigned)(CA_XATTN_KVF_ELEMS * 2) + (unsigned)lane_k * 8u;
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

    u32x2 kf[KT][TJ], vf[TJ][KT];
    u32x2 kfi[TJ], vfi[TJ];
    u32x2 op[TM][TJ];
    {
      f32x4 acc[TM][TJ];
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
            acc[i][j] = Elem<DT>::mfma(fb[kq & 1][j], fa[i], c0);
            if (j == TJ - 1 && kq + 1 < KQ) {
              __builtin_amdgcn_sched_barrier(0);
              fa[i] = ld16(xb + fa_b[i >> 2][nk & 1] + fa_off + (i & 3) * 16 * ROWB);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          if (kq + 2 < KQ) fb[kq & 1][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv, whead + (unsigned)((kq + 2) * TJ + j) * 1024u, 0));
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kq == KQ - 2) {  // the head's K fragments, under the last two chunks
#pragma unroll
          for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int c = 0; c < TJ; ++c) kf[kt][c] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_kv, kvb, (unsigned)((kt * 3 + c) * 512), 0));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      int lane_q = lane;
      asm volatile("" : "+v"(lane_q));
      const int gq = lane_q >> 4;
      u32x2 qp[TM][TJ];
      {
        f32x4 bi[TJ];
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) vf[j][kt] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_kv, kvb, (unsigned)((15 + j * 5 + kt) * 512), 0));
        if (IP) {  // key tile 0 of the image-prompt set: K fragments (0, c), V^T fragments (j, 0)
#pragma unroll
          for (int c = 0; c < TJ; ++c) kfi[c] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_kvi, kvb, (unsigned)(c * 512), 0));
#pragma unroll
          for (int j = 0; j < TJ; ++j) vfi[j] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_kvi, kvb, (unsigned)((15 + j * 5) * 512), 0));
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) bi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bi, (unsigned)(head * HD + 16 * j + 4 * gq) * 4u, 0, 0));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            qp[i][j] = (u32x2){pack2<DT>(acc[i][j][0] + bi[j][0], acc[i][j][1] + bi[j][1]), pack2<DT>(acc[i][j][2] + bi[j][2], acc[i][j][3] + bi[j][3])};
      }
      __builtin_amdgcn_sched_barrier(0);
      // IP: o goes into the tile as each row tile finishes (its 48 registers are what the second K / V set needs), so every wave must be
      // through with its q pass on the normalised tile here already -- the barrier the plain variant has behind the attention
      const int l15q = lane_q & 15;
      if (IP) __syncthreads();
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        f32x4 st[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          st[kt] = Elem<DT>::mfma16(kf[kt][0], qp[i][0], (f32x4){0.f, 0.f, 0.f, 0.f});
          st[kt] = Elem<DT>::mfma16(kf[kt][1], qp[i][1], st[kt]);
          st[kt] = Elem<DT>::mfma16(kf[kt][2], qp[i][2], st[kt]);
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kt >= 4 && 16 * kt + 4 * gq + r >= p.nk) st[kt][r] = -INFINITY;
        float m = fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3]));
#pragma unroll
        for (int kt = 1; kt < KT; ++kt) m = fmaxf(m, fmaxf(fmaxf(st[kt][0], st[kt][1]), fmaxf(st[kt][2], st[kt][3])));
        m = rowgroup_max(m);
        float l = 0.f;
        u32x2 pp[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
          float e[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            e[r] = __builtin_amdgcn_exp2f(st[kt][r] - m);
            l += e[r];
          }
          pp[kt] = (u32x2){pack2_prob<DT>(e[0], e[1]), pack2_prob<DT>(e[2], e[3])};
        }
        const float inv = __builtin_amdgcn_rcpf(rowgroup_sum(l));
        u32x2 ppi;
        float invi = 0.f;
        if (IP) {  // the image-prompt tokens: their own softmax over <= 16 keys (only the lanes of key group 0..3 hold real scores)
          f32x4 si = Elem<DT>::mfma16(kfi[0], qp[i][0], (f32x4){0.f, 0.f, 0.f, 0.f});
          si = Elem<DT>::mfma16(kfi[1], qp[i][1], si);
          si = Elem<DT>::mfma16(kfi[2], qp[i][2], si);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * gq + r >= p.nk_ip) si[r] = -INFINITY;
          const float mi = rowgroup_max(fmaxf(fmaxf(si[0], si[1]), fmaxf(si[2], si[3])));
          float ei[4], li = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            ei[r] = __builtin_amdgcn_exp2f(si[r] - mi);
            li += ei[r];
          }
          ppi = (u32x2){pack2_prob<DT>(ei[0], ei[1]), pack2_prob<DT>(ei[2], ei[3])};
          invi = p.ip_scale * __builtin_amdgcn_rcpf(rowgroup_sum(li));
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          f32x4 ot = Elem<DT>::mfma16(vf[j][0], pp[0], (f32x4){0.f, 0.f, 0.f, 0.f});  // O^T[d_v = 16 j + 4 g + r][row = l15]
#pragma unroll
          for (int kt = 1; kt < KT; ++kt) ot = Elem<DT>::mfma16(vf[j][kt], pp[kt], ot);
          if (IP) {
            const f32x4 oi = Elem<DT>::mfma16(vfi[j], ppi, (f32x4){0.f, 0.f, 0.f, 0.f});
            const u32x2 ov = {pack2<DT>(fmaf(oi[0], invi, ot[0] * inv), fmaf(oi[1], invi, ot[1] * inv)), pack2<DT>(fmaf(oi[2], invi, ot[2] * inv), fmaf(oi[3], invi, ot[3] * inv))};
            const int dv = 16 * j + 4 * gq;
            if (dv < HD) {
              const int col = head * HD + dv;
              ca_lds_store8(xb + (16 * i + l15q) * ROWB + (((col >> 3) ^ ((l15q >> 1) & 7)) << 4) + (col & 7) * 2, ov);
            }
          } else {
            op[i][j] = (u32x2){pack2<DT>(ot[0] * inv, ot[1] * inv), pack2<DT>(ot[2] * inv, ot[3] * inv)};
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int l15e = lane_e & 15, ge = lane_e >> 4;
    const unsigned row_base = (unsigned)(m0 + (wid >> 2) * 64 + l15e);
    AttnOutRegs R;
    attn_out_prefetch<DT>(R, p.out, rs_wo, rs_res, wid, lane_e, row_base, 16u, 32u);
    if (!IP) {
      __syncthreads();  // every wave has finished its K loop on the normalised tile
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const int dv = 16 * j + 4 * ge;
          if (dv < HD) {
            const int col = head * HD + dv;
            ca_lds_store8(xb + (16 * i + l15e) * ROWB + (((col >> 3) ^ ((l15e >> 1) & 7)) << 4) + (col & 7) * 2, op[i][j]);
          }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // the o tile is complete
    attn_out_run<DT>(R, xb, fa_b, p.out, rs_wo, rs_bo, rs_o, wid, lane_e, row_base, 16u, 32u, (unsigned)p.ldo);
  }
}

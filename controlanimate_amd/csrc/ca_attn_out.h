// The output projection of an attention, `y = o Wout^T + bias + residual` (reference: modules/attention_processor.py:258-270 --
// `attn.to_out[0]`, dropout = identity, `+ residual` in BasicTransformerBlock / animatediff/models/motion_module.py:212-224), as the
// LAST STAGE of the one-launch attention kernels of the 64x64-latent level (round 5, ABI v12): the block that computed the attention of
// a 128-row tile already owns all eight heads of those rows, so o goes into LDS instead of HBM and the 320 x 320 projection runs on it.
//
// Stage layout (8 waves, one block per CU): the o tile sits in the activation-resident layout of ca_gemm_ar.h (128 rows x 640 bytes,
// chunk c of row r at c ^ ((r >> 1) & 7)); wave (rh = wid >> 2, cg = wid & 3) computes rows 64 rh .. + 64 (four MFMA row tiles) of
// columns 80 cg .. + 80 (five column tiles, 80 accumulators): per 32-deep chunk four A fragments from LDS and five Wout fragments
// straight from L2 out of a fragment-ordered copy (ca_pack_w_out), two chunks ahead.  A lane holds 8 consecutive output columns of
// a pair of column tiles (the weight-row interleave of ca_gemm_ps.h): 16-byte residual loads and stores, 8-byte for the fifth tile.
//
// Fragment-ordered Wout: element e of lane L's 16 bytes of column tile j (0..4) of chunk kq (0..9) of column group cg (0..3) is
//   Wout[80 cg + ca_wout_col(j, L & 15)][32 kq + 8 (L >> 4) + e]      at ((((cg 10 + kq) 5 + j) 64 + L) 8 + e.
__device__ __forceinline__ int ca_wout_col(int j, int i) { return j < 4 ? 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3) : 64 + i; }

constexpr int CA_WOUT_ELEMS = 320 * 320;

__global__ __launch_bounds__(256) void k_pack_w_out(const u16* __restrict__ w, u16* __restrict__ dst) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // one 16-byte piece each
  if (idx >= CA_WOUT_ELEMS / 8) return;
  const int L = idx & 63;
  int t = idx >> 6;
  const int j = t % 5;
  t /= 5;
  const int kq = t % 10;
  const int cg = t / 10;
  const int row = cg * 80 + ca_wout_col(j, L & 15);
  st16(dst + (int64_t)idx * 8, ld16(w + (int64_t)row * 320 + kq * 32 + (L >> 4) * 8));
}

// 8-byte LDS store that the compiler does not fence: a compiler-visible LDS access narrower than 16 bytes into a region an LDS-DMA may
// fill is preceded by s_waitcnt vmcnt(0) -- here that would drain the stage's prefetch.  The caller retires it with
// s_waitcnt lgkmcnt(0) in front of its barrier.
__device__ __forceinline__ void ca_lds_store8(unsigned char* p, u32x2 v) {
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)p;
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

struct AttnOutParams {
  const u16* wof;      // fragment-ordered Wout, or NULL: no output stage
  const float* bias;   // [320] or NULL
  const u16* res;      // residual rows (same row indexing as the output) or NULL
  int ld_res;
  unsigned res_bytes;
};

// The registers a wave carries from `prefetch` (issued before the stage's barriers, so that the first chunks and the residual are on
// their way while o is written to LDS) to `run`.
struct AttnOutRegs {
  u32x4 fb[2][5];
  u32x4 rr[4][2];
  u32x2 r4[4];
};

// row_base: ROW index (not bytes) of this lane's row in row tile 0 of the wave's half; row tile i of the half is the row
// row_base + (i & 1) * step_a + (i >> 1) * step_b  (consecutive rows: 16, 32; the temporal kernels' (pixel, frame) tiles: see their callers).
__device__ __forceinline__ unsigned attn_out_row(unsigned row_base, unsigned step_a, unsigned step_b, int i) { return row_base + (unsigned)(i & 1) * step_a + (unsigned)(i >> 1) * step_b; }

template <int DT>
__device__ __forceinline__ void attn_out_prefetch(AttnOutRegs& R, const AttnOutParams& a, __amdgpu_buffer_rsrc_t rs_wo, __amdgpu_buffer_rsrc_t rs_res, int wid,
                                                  int lane, unsigned row_base, unsigned step_a, unsigned step_b) {
  const int cg = wid & 3, g = lane >> 4;
  const unsigned wv = (unsigned)lane * 16u;
  const unsigned wbase = (unsigned)cg * (10u * 5u * 1024u);
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int j = 0; j < 5; ++j) R.fb[c][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wo, wv, wbase + (unsigned)(c * 5 + j) * 1024u, 0));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned ro = attn_out_row(row_base, step_a, step_b, i) * (unsigned)a.ld_res * 2u + (unsigned)(cg * 80) * 2u;  // (size-0 descriptor without a residual: zeros)
    R.rr[i][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro + (unsigned)(8 * g) * 2u, 0, 0));
    R.rr[i][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_res, ro + (unsigned)(32 + 8 * g) * 2u, 0, 0));
    R.r4[i] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_res, ro + (unsigned)(64 + 4 * g) * 2u, 0, 0));
  }
}

template <int DT>
__device__ __forceinline__ unsigned attn_out_add(unsigned w, unsigned r_) {
  if (DT == CA_F16) {  // fp16 + fp16 is exact in fp32: the packed add rounds exactly like the fp32 path (ca_gemm_ar.h)
    unsigned s_;
    asm("v_pk_add_f16 %0, %1, %2" : "=v"(s_) : "v"(w), "v"(r_));
    return s_;
  }
  return pack2<DT>(Elem<DT>::to_f((u16)(w & 0xffffu)) + Elem<DT>::to_f((u16)(r_ & 0xffffu)), Elem<DT>::to_f((u16)(w >> 16)) + Elem<DT>::to_f((u16)(r_ >> 16)));
}

// tile: the o tile in LDS (behind a barrier).  fa_b: the A fragment base addresses of ca_gemm_ar.h ([row half][chunk parity]).
template <int DT>
__device__ __forceinline__ void attn_out_run(AttnOutRegs& R, const unsigned char* tile, const int (&fa_b)[2][2], const AttnOutParams& a,
                                             __amdgpu_buffer_rsrc_t rs_wo, __amdgpu_buffer_rsrc_t rs_bo, __amdgpu_buffer_rsrc_t rs_y, int wid, int lane,
                                             unsigned row_base, unsigned step_a, unsigned step_b, unsigned ldo) {
  constexpr int KQ = 10, ROWB = 640;
  const int rh = wid >> 2, cg = wid & 3;
  int lane_k = lane;
  asm volatile("" : "+v"(lane_k));
  const unsigned wv = (unsigned)lane_k * 16u;
  const unsigned wbase = (unsigned)cg * (10u * 5u * 1024u);
  f32x4 acc[4][5];
  u32x4 fa[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = ld16(tile + fa_b[rh][0] + i * 16 * ROWB);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kq = 0; kq < KQ; ++kq) {
    const int nk = kq + 1;
    const int fa_off = (nk >> 1) * 128;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (kq == 0) acc[i][j] = Elem<DT>::mfma(R.fb[0][j], fa[i], (f32x4){0.f, 0.f, 0.f, 0.f});
        else acc[i][j] = Elem<DT>::mfma(R.fb[kq & 1][j], fa[i], acc[i][j]);
        if (j == 4 && nk < KQ) {
          __builtin_amdgcn_sched_barrier(0);
          fa[i] = ld16(tile + fa_b[rh][nk & 1] + fa_off + i * 16 * ROWB);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (kq + 2 < KQ) R.fb[kq & 1][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_wo, wv, wbase + (unsigned)(((kq + 2) * 5 + j) * 1024), 0));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // epilogue: bias, round, + residual, stores (every load of the stage was issued before its first store)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int g = lane_e >> 4;
  f32x4 bi[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const unsigned c4 = (unsigned)(cg * 80 + (j < 4 ? 32 * (j >> 1) + 8 * g + 4 * (j & 1) : 64 + 4 * g)) * 4u;
    bi[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_bo, c4, 0, 0));
  }
  unsigned w[4][10];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      w[i][2 * j] = pack2<DT>(acc[i][j][0] + bi[j][0], acc[i][j][1] + bi[j][1]);
      w[i][2 * j + 1] = pack2<DT>(acc[i][j][2] + bi[j][2], acc[i][j][3] + bi[j][3]);
    }
    if (a.res) {
#pragma unroll
      for (int k = 0; k < 8; ++k) w[i][k] = attn_out_add<DT>(w[i][k], R.rr[i][k >> 2][k & 3]);
      w[i][8] = attn_out_add<DT>(w[i][8], R.r4[i][0]);
      w[i][9] = attn_out_add<DT>(w[i][9], R.r4[i][1]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned ro = attn_out_row(row_base, step_a, step_b, i) * ldo * 2u + (unsigned)(cg * 80) * 2u;
    __builtin_amdgcn_raw_buffer_store_b128((u32x4){w[i][0], w[i][1], w[i][2], w[i][3]}, rs_y, ro + (unsigned)(8 * g) * 2u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128((u32x4){w[i][4], w[i][5], w[i][6], w[i][7]}, rs_y, ro + (unsigned)(32 + 8 * g) * 2u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b64((u32x2){w[i][8], w[i][9]}, rs_y, ro + (unsigned)(64 + 4 * g) * 2u, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}

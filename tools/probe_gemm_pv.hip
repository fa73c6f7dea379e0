// Probe (round 6, measurement aid -- not part of the library): the 256 x 256 / four-wave / AGPR-accumulator tile of tools/probe_gemm_pr.hip made
// PERSISTENT, with what its ablations said was missing: (1) no per-tile prologue / epilogue bubble -- the global -> LDS stream runs two K tiles
// ahead of the MFMAs ACROSS tile boundaries, the accumulators are initialised by the first MFMA of a tile (C = 0 form), the epilogue's stores are
// fire-and-forget; (2) 64-deep K tiles = whole 128-byte lines per row and DMA instruction; (3) 16-byte output stores through interleaved weight rows
// (the trick of ca_gemm_ps.h); (4) XCD-compact tile order.  C[M, N] = A[M, K] W[N, K]^T, fp16, fp32 accumulate; M, N multiples of 256, K of 64, >= 128.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_gemm_pv.hip -o tools/probe_gemm_pv.bin && tools/probe_gemm_pv.bin [M N K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, KT = 64;
constexpr int STAGE_B = (BM + BN) * KT * 2;  // 64 KB
constexpr int OFF_W = BM * KT * 2;           // 32 KB

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ void k_fill(_Float16* p, size_t n, unsigned seed, float scale, int zeros) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < 4; ++k) s += (float)(hash(seed + (unsigned)i * 4u + k) & 0xffff) / 65536.f - 0.5f;
  p[i] = zeros ? (_Float16)0.f : (_Float16)(s * 1.732f * scale);
}
// output column of fragment row r (0..15) of MFMA tile j (0..7) inside a wave's 128 columns: tiles 2m, 2m+1 interleave in groups of four, so that
// a lane's 4 + 4 accumulator columns of the pair are 8 consecutive output columns
__device__ __forceinline__ int pv_col(int j, int r) { return 32 * (j >> 1) + 8 * (r >> 2) + 4 * (j & 1) + (r & 3); }

#define PV_MFMA(ACC, WA, XA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(WA), "v"(XA))
#define PV_MFMA0(ACC, WA, XA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "+a"(ACC) : "v"(WA), "v"(XA))  // ("+a": the same registers as the accumulating form -- no copies at the join)

template <int ABL>  // timing-only ablations (results wrong): 1 = no DMA after the first two K tiles, 2 = no stores
__global__ __launch_bounds__(256) void k_gemm_pv(const _Float16* __restrict__ A, const _Float16* __restrict__ W, _Float16* __restrict__ C, int M, int N, int K, int tiles_m,
                                                 int tiles_n, const unsigned* __restrict__ seq_table) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_B + 4 * 2 * 256];  // two stages + per wave two flag slots
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int g = lane >> 4, l15 = lane & 15;
  const int G = gridDim.x;
  const int tiles_total = tiles_m * tiles_n;
  const int bslot = (G % 8 == 0) ? (int)(blockIdx.x % 8) * (G / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;  // consecutive slots on one XCD
  const int my_tiles = bslot < tiles_total ? (tiles_total - bslot + G - 1) / G : 0;
  if (my_tiles == 0) return;
  const int nk = K / KT;
  const int total = my_tiles * nk;
  auto tile_of = [&](int seq, int& m0, int& n0) __attribute__((always_inline)) {
    const int id = seq * G + bslot;
    // panels of 4 row tiles, column-major inside a panel: 32 consecutive ids = a 4 x 8 patch
    const int per_panel = 4 * tiles_n;
    const int panel = id / per_panel, r = id - panel * per_panel;
    const int rows_here = tiles_m - panel * 4 < 4 ? tiles_m - panel * 4 : 4;
    m0 = (panel * 4 + r % rows_here) * BM;
    n0 = (r / rows_here) * BN;
  };
  const unsigned a_bytes = (unsigned)((size_t)M * K * 2), w_bytes = (unsigned)((size_t)N * K * 2), c_bytes = (unsigned)((size_t)M * N * 2);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, c_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_seq = __builtin_amdgcn_make_buffer_rsrc((void*)seq_table, 0, 4096u, 0x00020000);
  unsigned char* const my_flags = smem + 2 * STAGE_B + wid * 512;
  if (lane < 2) *reinterpret_cast<unsigned*>(my_flags + lane * 256) = 0xFFFFFFFFu;

  // ---- DMA side: a K tile is 32 + 32 pieces of 1 KB (8 rows x 128 B); wave w issues pieces 8w .. 8w + 7 of A and of W.  Lane: row lane >> 3 of the
  // piece, position lane & 7, which holds source chunk (lane & 7) ^ ((row >> 1) & 7) (conflict-free ds_read_b128, as ca_gemm_ps.h).
  unsigned va[8], vw[8];
  int d_T = 0, d_seq = -1;
  auto dma_set_tile = [&](int seq) __attribute__((always_inline)) {
    int m0, n0;
    tile_of(seq, m0, n0);
    if (ABL & 4) m0 = 0, n0 = 0;  // (every tile streams the operands of tile (0, 0): an L2-resident stream)
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int r8 = lane_o >> 3, cp = lane_o & 7;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = (wid * 8 + q) * 8 + r8;  // local row 0..255
      const unsigned ch = (unsigned)((cp ^ ((row >> 1) & 7)) * 16);
      va[q] = (unsigned)(m0 + row) * (unsigned)K * 2u + ch;
      const int wq = row >> 7, j = (row >> 4) & 7, r = row & 15;  // W stage row -> wave column half, MFMA tile, fragment row
      vw[q] = (unsigned)(n0 + wq * 128 + pv_col(j, r)) * (unsigned)K * 2u + ch;
    }
  };
  auto issue_piece = [&](int T, int q) __attribute__((always_inline)) {  // piece q (0..15) of K tile T of this block's stream: 0..7 A, 8..15 W
    unsigned char* buf = smem + (T & 1) * STAGE_B;
    const int kt = T % nk;
    const bool live = T < total && !((ABL & 1) && T >= 2);
    const unsigned koff = (unsigned)kt * (KT * 2);
    if (q < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(buf + (wid * 8 + q) * 1024), 16, live ? va[q] : 0x80000000u, koff, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_W + (wid * 8 + q - 8) * 1024), 16, live ? vw[q - 8] : 0x80000000u, koff, 0, 0);
  };
  auto dma_advance = [&](int T) __attribute__((always_inline)) {  // before the first piece of K tile T: the stream may enter the next tile
    if (T < total && T / nk != d_seq) {
      d_seq = T / nk;
      dma_set_tile(d_seq);
    }
  };

  // ---- compute side
  const int sw = (l15 >> 1) & 7;
  const int fx0 = (wr * 128 + l15) * 128 + ((g ^ sw) * 16), fw0 = OFF_W + (wc * 128 + l15) * 128 + ((g ^ sw) * 16);  // k half 0; half 1: ^ 64
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 xa[2][8], wa[2][8];

  dma_advance(0);
#pragma unroll
  for (int q = 0; q < 16; ++q) issue_piece(0, q);
  dma_advance(1);
#pragma unroll
  for (int q = 0; q < 16; ++q) issue_piece(1, q);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // K tile 0 landed (loads return in order)
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    xa[0][i] = *reinterpret_cast<const f16x8*>(smem + fx0 + i * 2048);
    wa[0][i] = *reinterpret_cast<const f16x8*>(smem + fw0 + i * 2048);
  }

  int T = 0;
  for (int seq = 0; seq < my_tiles; ++seq) {
    int m0, n0;
    tile_of(seq, m0, n0);
    for (int kt = 0; kt < nk; ++kt, ++T) {
      unsigned char* const sb = smem + (T & 1) * STAGE_B;        // this K tile
      unsigned char* const nb = smem + ((T + 1) & 1) * STAGE_B;  // the next one
      // ---- phase 1: MFMAs of k half 0 (register set 0); the fragments of k half 1 (set 1) are read behind them
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int idx = 0; idx < 64; ++idx) {
        const int i = (ABL & 8) ? (idx & 7) : (idx >> 3), j = (ABL & 8) ? (idx >> 3) : (idx & 7);
        PV_MFMA(acc[i][j], wa[0][j], xa[0][i]);
        if ((idx & 1) == 1 && (idx >> 1) < 16) {
          const int r = idx >> 1;
          if (r < 8) xa[1][r] = *reinterpret_cast<const f16x8*>(sb + (fx0 ^ 64) + r * 2048);
          else wa[1][r - 8] = *reinterpret_cast<const f16x8*>(sb + (fw0 ^ 64) + (r - 8) * 2048);
        }
      }
      // every wave has read this stage for the last time, and its own pieces of K tile T + 1 have landed: after the barrier the stage is free and
      // the next K tile is complete
      // (its own pieces: confirmed by the LDS flag, not by vmcnt -- the previous tile's output stores share that counter and must not be awaited)
      if (T + 1 < total && T + 1 >= 2) {
        const unsigned want = (unsigned)((T + 1) & 1023);
        const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(my_flags + ((T + 1) & 1) * 256);
        unsigned v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        for (unsigned spins = 0; (unsigned)__builtin_amdgcn_readfirstlane(v) != want && spins < (1u << 20); ++spins) {
          __builtin_amdgcn_s_sleep(1);
          asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        }
        if ((unsigned)__builtin_amdgcn_readfirstlane(v) != want) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      // ---- phase 2: MFMAs of k half 1 (set 1); K tile T + 2 is requested into this stage, the fragments of k half 0 of K tile T + 1 are read
      dma_advance(T + 2);
#pragma unroll
      for (int idx = 0; idx < 64; ++idx) {
        const int i = (ABL & 8) ? (idx & 7) : (idx >> 3), j = (ABL & 8) ? (idx >> 3) : (idx & 7);
        PV_MFMA(acc[i][j], wa[1][j], xa[1][i]);
        if (!(ABL & 16) && (idx & 1) == 0 && (idx >> 1) < 16) issue_piece(T + 2, idx >> 1);
        if ((ABL & 16) && (idx & 3) == 0) issue_piece(T + 2, idx >> 2);
        if (idx == ((ABL & 16) ? 62 : 32))  // the unit's flag: its sequence number, fetched BEHIND the sixteen pieces (loads return in order) into this wave's slot
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_seq, (__attribute__((address_space(3))) void*)(my_flags + ((T + 2) & 1) * 256), 4, 0u, (unsigned)((T + 2) & 1023) * 4u, 0, 0);
        if ((idx & 1) == 1 && idx >= 32) {
          const int r = (idx - 32) >> 1;
          if (r < 8) xa[0][r] = *reinterpret_cast<const f16x8*>(nb + fx0 + r * 2048);
          else wa[0][r - 8] = *reinterpret_cast<const f16x8*>(nb + fw0 + (r - 8) * 2048);
        }
      }
    }
    // ---- epilogue of the tile: lane holds rows m0 + wr 128 + 16 i + l15, columns n0 + wc 128 + 32 m + 8 g .. + 7 of tile pair (2m, 2m + 1)
    if (!(ABL & 2)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned ro = ((unsigned)(m0 + wr * 128 + 16 * i + l15) * (unsigned)N + (unsigned)(n0 + wc * 128 + 8 * g)) * 2u;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const f32x4 lo = acc[i][2 * m], hi = acc[i][2 * m + 1];
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          const h2 p0 = {(_Float16)lo[0], (_Float16)lo[1]}, p1 = {(_Float16)lo[2], (_Float16)lo[3]};
          const h2 p2 = {(_Float16)hi[0], (_Float16)hi[1]}, p3 = {(_Float16)hi[2], (_Float16)hi[3]};
          const u32x4 v = {__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1), __builtin_bit_cast(unsigned, p2), __builtin_bit_cast(unsigned, p3)};
          __builtin_amdgcn_raw_buffer_store_b128(v, rs_c, ro + (unsigned)(32 * m) * 2u, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);  // (row tile by row tile: the accumulators leave the AGPR half 32 at a time, not all 256 at once)
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};  // (256 v_accvgpr_write per tile, ~0.4 us)
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the block
}

__global__ void k_check(const _Float16* A, const _Float16* W, const _Float16* C, int M, int N, int K, int samples, float* maxerr) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= samples) return;
  const int m = hash(t * 2 + 1) % M, n = hash(t * 2 + 2) % N;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += (float)A[(size_t)m * K + k] * (float)W[(size_t)n * K + k];
  const float e = fabsf((float)C[(size_t)m * N + n] - s) / (1.f + fabsf(s));
  atomicMax(reinterpret_cast<int*>(maxerr), __float_as_int(e));
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef void (*kern_t)(const _Float16*, const _Float16*, _Float16*, int, int, int, int, int, const unsigned*);

int main(int argc, char** argv) {
  const int abl = getenv("PV_ABL") ? atoi(getenv("PV_ABL")) : 0;
  kern_t kern = abl == 1 ? k_gemm_pv<1> : abl == 2 ? k_gemm_pv<2> : abl == 3 ? k_gemm_pv<3> : abl == 4 ? k_gemm_pv<4> : abl == 6 ? k_gemm_pv<6> : abl == 8 ? k_gemm_pv<8> : abl == 16 ? k_gemm_pv<16> : abl == 24 ? k_gemm_pv<24> : abl == 10 ? k_gemm_pv<10> : abl == 26 ? k_gemm_pv<26> : k_gemm_pv<0>;
  unsigned* seq_table;
  {
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = (unsigned)i;
    CK(hipMalloc(&seq_table, 4096));
    CK(hipMemcpy(seq_table, h.data(), 4096, hipMemcpyHostToDevice));
  }
  if (abl) printf("ablation %d (results wrong)\n", abl);
  std::vector<int> shapes = {8192, 10240, 1280, 32768, 5120, 640, 8192, 3840, 1280, 8192, 1280, 5120, 2048, 10240, 1280, 8192, 1280, 1280};
  if (argc == 4) shapes = {atoi(argv[1]), atoi(argv[2]), atoi(argv[3])};
  for (size_t si = 0; si + 2 < shapes.size(); si += 3) {
    const int M = shapes[si], N = shapes[si + 1], K = shapes[si + 2];
    if (M % 256 || N % 256 || K % 64 || K < 128) { printf("%d x %d x %d: not a multiple of the tile\n", M, N, K); continue; }
    _Float16 *A, *W, *C;
    float* err;
    CK(hipMalloc(&A, (size_t)M * K * 2));
    CK(hipMalloc(&W, (size_t)N * K * 2));
    CK(hipMalloc(&C, (size_t)M * N * 2));
    CK(hipMalloc(&err, 4));
    for (int zeros = 0; zeros < 2; ++zeros) {
      k_fill<<<(unsigned)(((size_t)M * K + 255) / 256), 256>>>(A, (size_t)M * K, 1u, 1.f, zeros);
      k_fill<<<(unsigned)(((size_t)N * K + 255) / 256), 256>>>(W, (size_t)N * K, 77u, 1.f / sqrtf((float)K), zeros);
      CK(hipMemset(C, 0xff, (size_t)M * N * 2));
      const int tiles_m = M / BM, tiles_n = N / BN, tiles = tiles_m * tiles_n;
      const int grid = tiles < 256 ? tiles : 256;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, A, W, C, M, N, K, tiles_m, tiles_n, seq_table);
      CK(hipDeviceSynchronize());
      CK(hipMemset(err, 0, 4));
      k_check<<<64, 256>>>(A, W, C, M, N, K, 64 * 256, err);
      float e = 0.f;
      CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, A, W, C, M, N, K, tiles_m, tiles_n, seq_table);
      CK(hipEventRecord(e0));
      const int iters = 20;
      for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, A, W, C, M, N, K, tiles_m, tiles_n, seq_table);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters;
      printf("%6d x %6d x %5d  %s: %8.1f us  %7.1f TFLOP/s   (%d tiles of 256 x 256 on %d blocks; max rel err of 16384 sampled outputs %.2e)\n", M, N, K, zeros ? "zeros " : "N(0,1)", us,
             2.0 * M * N * K / us * 1e-6, tiles, grid, e);
    }
    hipFree(A); hipFree(W); hipFree(C); hipFree(err);
  }
  return 0;
}

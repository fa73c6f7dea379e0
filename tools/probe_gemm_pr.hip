// Probe (round 5, measurement aid -- not part of the library): a 256 x 256 block tile with FOUR waves of 128 x 128 (one wave per SIMD,
// 256 accumulator registers each), the tile shape of the vendor's fastest fp16 kernels on this chip, written with the library's own
// building blocks -- LDS-DMA units into a swizzled four-stage ring of 32-deep K steps, fragment reads of the next stage issued in front of
// the current stage's 64 MFMAs -- to price what the 128 x 80 wave tiles of ca_gemm_pq.h (two waves per SIMD) leave on the table on the
// wide-N projections (8192 x 10240 x 1280: 191 us there, 165 us vendor).  C[M, N] = A[M, K] W[N, K]^T, fp16, fp32 accumulate, no epilogue
// operands.  M, N multiples of 256, K of 32.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_gemm_pr.hip -o tools/probe_gemm_pr.bin && tools/probe_gemm_pr.bin [M N K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, KS = 32, NST = 4;
constexpr int STAGE_B = (BM + BN) * KS * 2;  // 32 KB
constexpr int OFF_W = BM * KS * 2;           // 16 KB

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ void k_fill(_Float16* p, size_t n, unsigned seed, float scale, int zeros) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // sum of four uniforms: close enough to N(0, 1) in its bit statistics (what the matrix pipes' power draw depends on)
  float s = 0.f;
  for (int k = 0; k < 4; ++k) s += (float)(hash(seed + (unsigned)i * 4u + k) & 0xffff) / 65536.f - 0.5f;
  p[i] = zeros ? (_Float16)0.f : (_Float16)(s * 1.732f * scale);
}

template <int ABL, int UNIT>  // UNIT = 1: the two 32-deep halves of a 128-byte line are requested back to back (units of two steps)
// timing-only ablations (results wrong): 1 = no DMA inside the loop, 2 = no fragment reads, 4 = no barriers, 8 = no MFMAs
__global__ __launch_bounds__(256) void k_gemm_pr(const _Float16* __restrict__ A, const _Float16* __restrict__ W, _Float16* __restrict__ C, int M, int N, int K, int tiles_n, int xmap) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;
  const int g = lane >> 4, l15 = lane & 15;
  int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  if (xmap) {  // XCD-compact: the 32 blocks an XCD runs at a time are a 4 x 8 patch of tiles (round-robin dispatch: XCD = blockIdx % 8)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3, per = (int)gridDim.x >> 3;  // tiles per XCD = 4 tile rows x tiles_n
    const int grp = idx >> 5, r = idx & 31;
    tm = xcd * (per / tiles_n) + (r & 3);
    tn = grp * 8 + (r >> 2);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const unsigned a_bytes = (unsigned)((size_t)M * K * 2), w_bytes = (unsigned)((size_t)N * K * 2);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, w_bytes, 0x00020000);

  // DMA: a stage is 16 + 16 pieces of 1 KB (16 rows x 64 B); wave w issues pieces 4w .. 4w + 3 of A and of W.  Lane i of a piece: row
  // i >> 2, position i & 3 -- which holds the 16-byte chunk (i & 3) ^ ((row >> 1) & 3) of that row (fragment reads of 8 consecutive rows
  // then hit 8 different 16-byte bank groups).
  unsigned va[4], vw[4];
  {
    const int r16 = lane >> 2, c = lane & 3;
    const unsigned ch = (unsigned)((c ^ ((r16 >> 1) & 3)) * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = (wid * 4 + q) * 16 + r16;
      va[q] = (unsigned)(m0 + row) * (unsigned)K * 2u + ch;
      vw[q] = (unsigned)(n0 + row) * (unsigned)K * 2u + ch;
    }
  }
  auto issue = [&](int s) __attribute__((always_inline)) {  // (beyond the last K step: offsets outside the descriptors -- zeros into a buffer nobody reads)
    unsigned char* buf = smem + (s & (NST - 1)) * STAGE_B;
    const unsigned koff = (unsigned)s * (KS * 2);
    const bool live = s * KS < K;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(buf + (wid * 4 + q) * 1024), 16, live ? va[q] : 0x80000000u, koff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(buf + OFF_W + (wid * 4 + q) * 1024), 16, live ? vw[q] : 0x80000000u, koff, 0, 0);
    }
  };
  // fragment addresses inside a stage: x rows (B operand) and W rows (A operand); row tile i adds i * 16 * 64 bytes
  const int fsw = (g ^ ((l15 >> 1) & 3)) * 16;
  const int fx = (wr * 128 + l15) * 64 + fsw, fw = OFF_W + (wc * 128 + l15) * 64 + fsw;

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 xa[2][8], wa[2][8];

  const int ns = K / KS;  // (even, >= 2: K % 64 == 0)
  auto issue_unit = [&](int u) __attribute__((always_inline)) {  // stages 2u, 2u + 1: piece by piece, the two halves of its lines back to back
    unsigned char* b0 = smem + ((2 * u) & (NST - 1)) * STAGE_B;
    unsigned char* b1 = smem + ((2 * u + 1) & (NST - 1)) * STAGE_B;
    const unsigned k0 = (unsigned)(2 * u) * (KS * 2), k1 = k0 + KS * 2;
    const bool live = 2 * u * KS < K;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(b0 + (wid * 4 + q) * 1024), 16, live ? va[q] : 0x80000000u, k0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(b1 + (wid * 4 + q) * 1024), 16, live ? va[q] : 0x80000000u, k1, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(b0 + OFF_W + (wid * 4 + q) * 1024), 16, live ? vw[q] : 0x80000000u, k0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(b1 + OFF_W + (wid * 4 + q) * 1024), 16, live ? vw[q] : 0x80000000u, k1, 0, 0);
    }
  };
  if (UNIT) {
    issue_unit(0);
    issue_unit(1);
  } else {
    issue(0);
    issue(1);
    issue(2);
    issue(3);
  }
  if (UNIT) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // unit 0 landed
  else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // stage 0 landed (loads return in order)
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    xa[0][i] = *reinterpret_cast<const f16x8*>(smem + fx + i * 1024);
    wa[0][i] = *reinterpret_cast<const f16x8*>(smem + fw + i * 1024);
  }
  // One K step (s + 1 < ns): stage s is in register set CUR.  Behind the barrier that publishes stage s + 1: the 64 MFMAs of stage s in a
  // fixed order (inline asm, accumulators pinned to the AGPR half of the register file) with the fragment reads of stage s + 1 (one per two
  // MFMAs) and the request for stage s + 4 (one piece per four MFMAs, into the buffer stage s leaves) placed between them by hand, so
  // that the matrix pipe never waits for the other instructions' issue slots.
#define PR_MFMA(ACC, WA, XA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(WA), "v"(XA) : "memory")
  auto step = [&](int s, auto CUR) __attribute__((always_inline)) {
    constexpr int cur = decltype(CUR)::value;
    if ((ABL & 1) || (UNIT && cur == 1)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (UNIT, odd step: the next unit, whole)
    else asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");  // stage s + 1 landed (s + 2, s + 3 may be outstanding); set CUR complete
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
    unsigned char* const nb = smem + ((s + 1) & (NST - 1)) * STAGE_B;
    unsigned char* const db = smem + (s & (NST - 1)) * STAGE_B;
    const unsigned koff = (unsigned)(s + 4) * (KS * 2);
    const bool live = (s + 4) * KS < K;
#pragma unroll
    for (int idx = 0; idx < 64; ++idx) {
      const int i = idx >> 3, j = idx & 7;
      if (!(ABL & 8)) PR_MFMA(acc[i][j], wa[cur][j], xa[cur][i]);
      if (!(ABL & 2) && (idx & 1) == 1 && (idx >> 1) < 16) {
        const int r = idx >> 1;
        if (r < 8) xa[cur ^ 1][r] = *reinterpret_cast<const f16x8*>(nb + fx + r * 1024);
        else wa[cur ^ 1][r - 8] = *reinterpret_cast<const f16x8*>(nb + fw + (r - 8) * 1024);
      }
      if (UNIT && !(ABL & 1) && cur == 1 && (idx & 3) == 3) {  // odd step: unit (s + 3) / 2 into the buffers of steps s - 1 and s, one request per four MFMAs
        const int q = idx >> 2;  // 0 .. 15: A piece q >> 1 half q & 1 (0 .. 7), then W
        const int pc = (q & 7) >> 1, hf = q & 1;
        unsigned char* ub = smem + ((s + 3 + hf) & (NST - 1)) * STAGE_B;
        const unsigned uk = (unsigned)(s + 3 + hf) * (KS * 2);
        const bool ulive = (s + 3) * KS < K;
        if (q < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(ub + (wid * 4 + pc) * 1024), 16, ulive ? va[pc] : 0x80000000u, uk, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(ub + OFF_W + (wid * 4 + pc) * 1024), 16, ulive ? vw[pc] : 0x80000000u, uk, 0, 0);
      }
      if (!UNIT && !(ABL & 1) && idx >= 32 && (idx & 3) == 3) {
        const int q = (idx - 32) >> 2;  // 0 .. 7: A pieces, then W pieces
        if (q < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_a, (__attribute__((address_space(3))) void*)(db + (wid * 4 + q) * 1024), 16, live ? va[q] : 0x80000000u, koff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(db + OFF_W + (wid * 4 + q - 4) * 1024), 16, live ? vw[q - 4] : 0x80000000u, koff, 0, 0);
      }
    }
  };
  for (int s = 0; s < ns; s += 2) {  // (the last step reads a stage that does not exist -- zeros -- into a set nobody uses)
    step(s, std::integral_constant<int, 0>{});
    step(s + 1, std::integral_constant<int, 1>{});
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the block
  // epilogue: lane holds C[m0 + wr 128 + 16 i + l15][n0 + wc 128 + 16 j + 4 g .. + 4]
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    _Float16* row = C + (size_t)(m0 + wr * 128 + 16 * i + l15) * N + n0 + wc * 128 + 4 * g;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f16x4 v = {(_Float16)acc[i][j][0], (_Float16)acc[i][j][1], (_Float16)acc[i][j][2], (_Float16)acc[i][j][3]};
      *reinterpret_cast<f16x4*>(row + 16 * j) = v;
    }
  }
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

typedef void (*kern_t)(const _Float16*, const _Float16*, _Float16*, int, int, int, int, int);
static kern_t pick(int abl, int unit) {
  if (unit) {
    switch (abl) {
      case 8: return k_gemm_pr<8, 1>;
      case 10: return k_gemm_pr<10, 1>;
      default: return k_gemm_pr<0, 1>;
    }
  }
  switch (abl) {
    case 1: return k_gemm_pr<1, 0>;
    case 2: return k_gemm_pr<2, 0>;
    case 3: return k_gemm_pr<3, 0>;
    case 4: return k_gemm_pr<4, 0>;
    case 7: return k_gemm_pr<7, 0>;
    case 8: return k_gemm_pr<8, 0>;
    case 9: return k_gemm_pr<9, 0>;
    case 10: return k_gemm_pr<10, 0>;
    default: return k_gemm_pr<0, 0>;
  }
}

int main(int argc, char** argv) {
  const int abl = getenv("PR_ABL") ? atoi(getenv("PR_ABL")) : 0;
  const int unit = getenv("PR_UNIT") ? atoi(getenv("PR_UNIT")) : 0;
  kern_t kern = pick(abl, unit);
  const int xmap_req = getenv("PR_MAP") ? atoi(getenv("PR_MAP")) : 0;
  if (unit) printf("units of two steps (whole 128-byte lines requested back to back)\n");
  if (abl) printf("ablation %d (results wrong)\n", abl);
  std::vector<int> shapes = {8192, 10240, 1280, 32768, 5120, 640, 8192, 3840, 1280, 8192, 1280, 5120, 8192, 1280, 1280, 32768, 1920 + 128, 640};
  if (argc == 4) shapes = {atoi(argv[1]), atoi(argv[2]), atoi(argv[3])};
  for (size_t si = 0; si + 2 < shapes.size(); si += 3) {
    const int M = shapes[si], N = shapes[si + 1], K = shapes[si + 2];
    if (M % 256 || N % 256 || K % 64) { printf("%d x %d x %d: not a multiple of the tile\n", M, N, K); continue; }
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
      const int tiles_n = N / BN, tiles = (M / BM) * tiles_n;
      const int xmap = xmap_req && (M / BM) == 32 && tiles_n % 8 == 0;  // (the mapping above: 8 XCDs x 4 tile rows)
      hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), 0, 0, A, W, C, M, N, K, tiles_n, xmap);
      CK(hipDeviceSynchronize());
      CK(hipMemset(err, 0, 4));
      k_check<<<64, 256>>>(A, W, C, M, N, K, 64 * 256, err);
      float e = 0.f;
      CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), 0, 0, A, W, C, M, N, K, tiles_n, xmap);
      CK(hipEventRecord(e0));
      const int iters = 20;
      for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), 0, 0, A, W, C, M, N, K, tiles_n, xmap);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters;
      printf("%6d x %6d x %5d  %s: %8.1f us  %7.1f TFLOP/s   (%d tiles of 256 x 256; max rel err of 16384 sampled outputs %.2e)\n", M, N, K, zeros ? "zeros " : "N(0,1)", us,
             2.0 * M * N * K / us * 1e-6, tiles, e);
    }
    hipFree(A); hipFree(W); hipFree(C); hipFree(err);
  }
  return 0;
}

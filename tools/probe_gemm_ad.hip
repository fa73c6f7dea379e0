// Feasibility probe: GEMM with the A fragments loaded global -> VGPR directly (no LDS), W through LDS-DMA
// (double-buffered).  C[M,N] = A[M,K] W[N,K]^T, fp16, fp32 accumulate, plain stores (no fused epilogue).
//   hipcc --offload-arch=gfx950 -O3 -w -mllvm -amdgpu-mfma-vgpr-form tools/probe_gemm_ad.hip -o tools/probe_gemm_ad.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
constexpr int BM = 128, BN = 128, KT = 64, TM = 4, TN = 4;

__device__ __forceinline__ f32x4 mfma(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  unsigned xcd = bid % 8, idx = bid / 8, q = nwg / 8, r = nwg % 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int OCC>
__global__ __launch_bounds__(256, OCC) void k_gemm_ad(const u16* A, const u16* W, u16* C, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) u16 smem[2 * BN * KT];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1, g = lane >> 4, l15 = lane & 15;
  const int tiles_n = N / BN, tiles_m = M / BM;
  unsigned bid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned gsz = 8 * tiles_n, group = bid / gsz, in = bid - group * gsz;
  const int first_m = group * 8, gm = tiles_m - first_m < 8 ? tiles_m - first_m : 8;
  const int m0 = (first_m + in % gm) * BM, n0 = (in / gm) * BN;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (unsigned)((size_t)M * K * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (unsigned)((size_t)N * K * 2), 0x00020000);
  // W staging: 8 rows x 8 chunks per wave instruction, 4 instructions per wave per stage
  const int r8 = lane >> 3, cp = lane & 7;
  unsigned b_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wid * 4 + j) * 8 + r8;
    b_off[j] = (unsigned)(n0 + row) * (unsigned)K * 2u + (unsigned)((cp ^ ((row >> 1) & 7)) * 16);
  }
  auto stage_w = [&](int t, int buf) {
    u16* sb = smem + buf * BN * KT;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(sb + (wid * 4 + j) * 8 * KT), 16,
                                               b_off[j] + (unsigned)t * KT * 2u, 0, 0, 0);
  };
  // A fragments: lane (row l15 of 16-row tile i, k chunk g of k32 step s) -> one 16-byte load
  unsigned a_off[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) a_off[i] = (unsigned)(m0 + wm * 64 + i * 16 + l15) * (unsigned)K * 2u + (unsigned)g * 16u;
  auto load_a = [&](int t, u32x4 (&fa)[2][TM]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[s][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, a_off[i] + (unsigned)(t * KT + s * 32) * 2u, 0, 0);
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto compute = [&](int buf, u32x4 (&fa)[2][TM]) {
    const u16* sb = smem + buf * BN * KT;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 fb[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * 64 + j * 16 + l15;
        fb[j] = *reinterpret_cast<const u32x4*>(sb + row * KT + (((s * 4 + g) ^ ((row >> 1) & 7)) << 3));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma(fb[j], fa[s][i], acc[i][j]);
    }
  };
  const int nt = K / KT;
  u32x4 fa0[2][TM], fa1[2][TM];
  load_a(0, fa0);
  stage_w(0, 0);
  __syncthreads();
  for (int t = 0; t < nt; t += 2) {
    if (t + 1 < nt) {
      stage_w(t + 1, 1);
      load_a(t + 1, fa1);
    }
    compute(0, fa0);
    __syncthreads();
    if (t + 1 < nt) {
      if (t + 2 < nt) {
        stage_w(t + 2, 0);
        load_a(t + 2, fa0);
      }
      compute(1, fa1);
      __syncthreads();
    }
  }
  // lane holds C[m = .. + l15][n = .. + 4g + (0..3)]
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * 64 + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 64 + j * 16 + g * 4;
      typedef _Float16 h4 __attribute__((ext_vector_type(4)));
      h4 o = {(_Float16)acc[i][j][0], (_Float16)acc[i][j][1], (_Float16)acc[i][j][2], (_Float16)acc[i][j][3]};
      *reinterpret_cast<h4*>(C + (size_t)m * N + n) = o;
    }
  }
}

template <int OCC>
float run(const u16* A, const u16* W, u16* C, int M, int N, int K) {
  dim3 grid((M / BM) * (N / BN));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_gemm_ad<OCC>, grid, dim3(256), 0, 0, A, W, C, M, N, K);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_gemm_ad<OCC>, grid, dim3(256), 0, 0, A, W, C, M, N, K);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}

int main() {
  const int shapes[][3] = {{8192, 10240, 1280}, {8192, 1280, 5120}, {32768, 5120, 640}, {8192, 1280, 11520}, {32768, 640, 5760}, {131072, 2560, 320}};
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    std::vector<u16> ha((size_t)M * K), hw((size_t)N * K);
    srand(1);
    for (auto& v : ha) { _Float16 f = (_Float16)((rand() % 2001 - 1000) * 1e-3f); v = __builtin_bit_cast(u16, f); }
    for (auto& v : hw) { _Float16 f = (_Float16)((rand() % 2001 - 1000) * 1e-3f * 0.03f); v = __builtin_bit_cast(u16, f); }
    u16 *A, *W, *C;
    hipMalloc(&A, ha.size() * 2); hipMalloc(&W, hw.size() * 2); hipMalloc(&C, (size_t)M * N * 2);
    hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    const float ms3 = run<3>(A, W, C, M, N, K), ms2 = run<2>(A, W, C, M, N, K);
    // spot check a few outputs
    std::vector<u16> hc(1024);
    hipMemcpy(hc.data(), C + (size_t)(M - 1) * N + (N - 1024), 2048, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int n = N - 1024; n < N; n += 97) {
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)(float)__builtin_bit_cast(_Float16, ha[(size_t)(M - 1) * K + k]) * (double)(float)__builtin_bit_cast(_Float16, hw[(size_t)n * K + k]);
      double got = (double)(float)__builtin_bit_cast(_Float16, hc[n - (N - 1024)]);
      if (fabs(got - ref) > maxerr) maxerr = fabs(got - ref);
    }
    printf("A-direct %6dx%5dx%5d: occ3 %7.1f us %7.1f TF | occ2 %7.1f us %7.1f TF | spot max abs err %.3g\n", M, N, K, ms3 * 1e3, 2.0 * M * N * K / ms3 / 1e9,
           ms2 * 1e3, 2.0 * M * N * K / ms2 / 1e9, maxerr);
    hipFree(A); hipFree(W); hipFree(C);
  }
  return 0;
}

// MFMA issue-rate probe on gfx950: cycles per v_mfma_f32_16x16x32_f16 for ONE wave per SIMD and for 2 / 4 waves per SIMD,
// with the accumulators in ArchVGPRs (what hipcc emits for the GEMM kernels) and in AccVGPRs (inline asm), 40 independent
// accumulators (the 128 x 80 wave tile of ca_gemm_pq.h) or 20 (ca_gemm_ps.h).  s_memtime cycles AND wall clock (hipEvent):
// the ratio gives the shader clock under this load.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma.hip -o tools/probe_mfma.bin && tools/probe_mfma.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
  f16x8 a, b[5];
  for (int i = 0; i < 8; ++i) a[i] = (_Float16)(seed + threadIdx.x * 1e-3f);
  for (int j = 0; j < 5; ++j)
    for (int i = 0; i < 8; ++i) b[j][i] = (_Float16)(seed * 0.5f + j);
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[i % 5], a, acc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(b[i % 5]), "v"(a));
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

template <int NACC, int MODE>
void run(const char* name, int waves_per_simd) {
  float* d;
  const int blocks = 256, threads = 256 * waves_per_simd, iters = 2000;
  hipMalloc(&d, sizeof(float) * blocks * threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, d, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0, cyc = 0;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
  const double n = (double)iters * NACC;  // MFMAs per wave
  printf("%-44s waves/SIMD %d: %6.2f s_memtime ticks per MFMA per wave, %6.2f ns per MFMA per SIMD, %7.1f TFLOP/s chip, ticks/ns %.3f\n", name, waves_per_simd, cyc / n,
         ms * 1e6 / (n * waves_per_simd), 16384.0 * n * waves_per_simd * 1024 / (ms * 1e-3) / 1e12, cyc / (ms * 1e6));
  hipFree(d);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<40, 0>("40 accumulators, ArchVGPR (hipcc builtin)", w);
    if (w <= 2) run<40, 1>("40 accumulators, AccVGPR (inline asm)", w);
    run<20, 0>("20 accumulators, ArchVGPR (hipcc builtin)", w);
    run<20, 1>("20 accumulators, AccVGPR (inline asm)", w);
  }
  return 0;
}

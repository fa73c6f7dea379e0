// Issue-rate probe: wave64 VALU op classes on gfx950 (cycles per instruction per SIMD).
// hipcc --offload-arch=gfx950 -O3 tools/probe_rates.hip -o /tmp/probe_rates && /tmp/probe_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void k(float* out, int iters, float seed) {
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) a[i] = __builtin_amdgcn_exp2f(a[i]);
      if (MODE == 1) a[i] = fmaf(a[i], 1.0001f, 0.5f);
      if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a[i & 6]) : "v"(*(double*)&a[(i + 2) & 6]));
      if (MODE == 3) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
      if (MODE == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
      if (MODE == 5) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
      if (MODE == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
template <int MODE>
void run(const char* name, int waves_per_simd) {
  float* d;
  hipMalloc(&d, 1 << 22);
  const int iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  dim3 grid(256), block(256 * waves_per_simd);
  hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, iters, 0.1f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, d, iters, 0.1f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  float clk;
  hipMemcpy(&clk, d, 4, hipMemcpyDeviceToHost);
  // per SIMD: waves_per_simd waves x iters x 8 instructions
  double n = (double)waves_per_simd * iters * 8;
  printf("%-22s waves/SIMD=%d  %.2f clock64-ticks/instr (one wave)  %.3f ns/instr/SIMD\n", name, waves_per_simd, clk / (iters * 8.0), ms * 1e6 / n);
  hipFree(d);
}
int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_exp_f32", w);
    run<1>("v_fma_f32", w);
    run<2>("v_pk_fma_f32", w);
    run<3>("v_cvt_pkrtz_f16_f32", w);
    run<4>("v_max3_f32", w);
    run<5>("v_exp_f16", w);
    run<6>("v_rcp_f32", w);
  }
  return 0;
}

// Does VALU/transcendental work overlap with MFMA on one SIMD (same wave / different waves)?
// hipcc --offload-arch=gfx950 -O3 -w tools/probe_overlap.hip -o tools/probe_overlap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// what: bit0 = MFMA, bit1 = exp, bit2 = fma ; split: waves with odd id do only the VALU part, even only MFMA
template <int NM, int NE, int NF>
__global__ void k(float* out, int iters, int split) {
  f32x4 acc[8];
  float e[16], f[16];
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f); b[i] = (_Float16)0.5f; }
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 16; ++i) { e[i] = threadIdx.x * 1e-3f - i; f[i] = i; }
  const int wid = threadIdx.x >> 6;
  const bool do_m = !split || ((wid >> 2) & 1) == 0;
  const bool do_v = !split || ((wid >> 2) & 1) == 1;
  for (int it = 0; it < iters; ++it) {
    if (do_m && do_v) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < NM) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (i * 2 + j < NE) e[i * 2 + j] = __builtin_amdgcn_exp2f(e[i * 2 + j]);
          if (i * 2 + j < NF) f[i * 2 + j] = fmaf(f[i * 2 + j], 1.0001f, 0.5f);
        }
      }
    } else if (do_m) {
#pragma unroll
      for (int i = 0; i < NM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i < NE) e[i] = __builtin_amdgcn_exp2f(e[i]);
        if (i < NF) f[i] = fmaf(f[i], 1.0001f, 0.5f);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += e[i] + f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NM, int NE, int NF>
void run(const char* name, int waves_per_simd, int split) {
  float* d;
  hipMalloc(&d, 1 << 22);
  const int iters = 8192;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  dim3 grid(256), block(256 * waves_per_simd);
  hipLaunchKernelGGL((k<NM, NE, NF>), grid, block, 0, 0, d, iters, split);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NM, NE, NF>), grid, block, 0, 0, d, iters, split);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-34s waves/SIMD=%d split=%d  %.1f ns per loop iteration\n", name, waves_per_simd, split, ms * 1e6 / iters);
  hipFree(d);
}
int main() {
  run<8, 0, 0>("8 MFMA", 1, 0);
  run<0, 16, 0>("16 exp", 1, 0);
  run<0, 0, 16>("16 fma", 1, 0);
  run<0, 16, 16>("16 exp + 16 fma", 1, 0);
  run<8, 16, 0>("8 MFMA + 16 exp (one stream)", 1, 0);
  run<8, 0, 16>("8 MFMA + 16 fma (one stream)", 1, 0);
  run<8, 16, 16>("8 MFMA + 16 exp + 16 fma (one)", 1, 0);
  run<8, 16, 16>("8 MFMA | 16 exp + 16 fma (2 waves)", 2, 1);
  run<8, 16, 0>("8 MFMA | 16 exp (2 waves)", 2, 1);
  run<8, 16, 16>("2 x (8 MFMA + 16 exp + 16 fma)", 2, 0);
  run<8, 0, 0>("2 x 8 MFMA", 2, 0);
  return 0;
}

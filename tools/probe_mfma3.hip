// Probe: what the matrix pipes sustain as a function of the OPERAND DATA (power throttling) and of the MFMA shape:
// v_mfma_f32_16x16x32_f16 (40 accumulators: the 128 x 80 wave tile of ca_gemm_pq.h) against v_mfma_f32_32x32x16_f16
// (10 accumulators of 16 registers: a 64 x 160 wave tile) -- equal FLOPs per iteration, equal accumulator registers; the
// 32x32 shape reads half the operand registers per FLOP.  One or two waves per SIMD, operands cycled as a GEMM does.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma3.hip -o tools/probe_mfma3.bin && tools/probe_mfma3.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// mode 0: zeros; 1: values like activations (uniform in [-2, 2) rounded to fp16); 2: random bit patterns (finite)
__device__ __forceinline__ f16x8 operand(int mode, unsigned id) {
  f16x8 v;
  for (int i = 0; i < 8; ++i) {
    const unsigned h = hash(id * 8u + i + 1u);
    if (mode == 0) v[i] = (_Float16)0.f;
    else if (mode == 1) v[i] = (_Float16)(((float)(h & 0xffff) / 65536.f - 0.5f) * 4.f);
    else {
      unsigned short bits = (unsigned short)(h & 0xBBFFu);  // exponent field < 0x1f: finite
      v[i] = __builtin_bit_cast(_Float16, bits);
    }
  }
  return v;
}

template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  if (SHAPE == 16) {
    f16x8 fa[8], fb[5];
    for (int i = 0; i < 8; ++i) fa[i] = operand(mode, tid * 16u + i);
    for (int j = 0; j < 5; ++j) fb[j] = operand(mode, tid * 16u + 8 + j);
    f32x4 acc[8][5];
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 5; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(fa[i]));
    }
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 5; ++j) s += acc[i][j][0] + acc[i][j][3];
  } else {
    f16x8 fa[2][2], fb[2][5];
    for (int kk = 0; kk < 2; ++kk) {
      for (int i = 0; i < 2; ++i) fa[kk][i] = operand(mode, tid * 16u + kk * 8 + i);
      for (int j = 0; j < 5; ++j) fb[kk][j] = operand(mode, tid * 16u + kk * 8 + 2 + j);
    }
    f32x16 acc[2][5];
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 5; ++j)
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(fa[0][i]), "+v"(fa[1][i]));
    }
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 5; ++j) s += acc[i][j][0] + acc[i][j][15];
  }
  out[tid] = s;
}

template <int SHAPE>
void run(int waves_per_simd, int mode) {
  float* d;
  const int blocks = 256, threads = 256 * waves_per_simd, iters = 4000;
  hipMalloc(&d, sizeof(float) * blocks * threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(threads), 0, 0, d, 50, mode);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE>), dim3(blocks), dim3(threads), 0, 0, d, iters, mode);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flop = (double)iters * 40.0 * 16384.0 * waves_per_simd * 1024.0;  // 40 x (16x16x32) == 20 x (32x32x16) per wave and iteration
  static const char* names[3] = {"zero operands", "activation-like operands", "random bit patterns"};
  printf("v_mfma_f32_%s, %d wave(s) per SIMD, %-26s: %7.1f TFLOP/s chip\n", SHAPE == 16 ? "16x16x32_f16" : "32x32x16_f16", waves_per_simd, names[mode], flop / (best * 1e-3) / 1e12);
  hipFree(d);
}

int main() {
  for (int w = 1; w <= 2; ++w)
    for (int mode = 0; mode < 3; ++mode) {
      run<16>(w, mode);
      run<32>(w, mode);
    }
  return 0;
}

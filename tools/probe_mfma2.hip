// What slows a wave's MFMAs inside the ping-pong kernels (26 instead of 16.8 cycles per 16x16x32 MFMA)?  Waves 0..3 of a block
// (one per SIMD) issue independent MFMAs back to back; waves 4..7 (their SIMD partners) run one of:
//   0 nothing (exit)   1 ds_read_b128 bursts (13 reads, wait)   2 full-rate VALU (v_fma)   3 LDS-DMA (buffer_load ... lds) bursts
//   4 s_sleep loop     5 ds_read bursts + VALU
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma2.hip -o tools/probe_mfma2.bin && tools/probe_mfma2.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PARTNER>
__global__ __launch_bounds__(512) void k(float* out, const unsigned* src, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 1024];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16 * 1024; i += 512) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  if (wid < 4) {
    f16x8 a, b[5];
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(seed + lane * 1e-3f);
    for (int j = 0; j < 5; ++j)
      for (int i = 0; i < 8; ++i) b[j][i] = (_Float16)(seed * 0.5f + j);
    f32x4 acc[20];
    for (int i = 0; i < 20; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 20; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[i % 5], a, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 20; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
  } else {
    if (PARTNER == 0) return;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    u32x4 r[13];
    unsigned acc = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1u << 20, 0x00020000);
    for (int it = 0; it < iters * 2; ++it) {
      if (PARTNER == 1 || PARTNER == 5) {
#pragma unroll
        for (int q = 0; q < 13; ++q) r[q] = *reinterpret_cast<const u32x4*>(lds + ((lane * 16 + q * 1024 + it * 64) & 0xFFF0));
#pragma unroll
        for (int q = 0; q < 13; ++q) acc += r[q][0];
      }
      if (PARTNER == 2 || PARTNER == 5) {
#pragma unroll
        for (int q = 0; q < 64; ++q) v[q & 7] = fmaf(v[q & 7], 1.0001f, 0.5f);
      }
      if (PARTNER == 3) {
#pragma unroll
        for (int q = 0; q < 9; ++q)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 32768 + (wid - 4) * 1024 * 8 + (q & 7) * 1024), 16, (unsigned)(lane * 16 + ((it * 9 + q) & 1023) * 1024), 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (PARTNER == 4) __builtin_amdgcn_s_sleep(20);
    }
    float s = acc;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

template <int PARTNER>
void run(const char* name) {
  float* d;
  unsigned* src;
  const int blocks = 256, iters = 4000;
  hipMalloc(&d, sizeof(float) * blocks * 512);
  hipMalloc(&src, 2u << 20);
  hipMemset(src, 1, 2u << 20);
  hipLaunchKernelGGL((k<PARTNER>), dim3(blocks), dim3(512), 0, 0, d, src, 10, 1.0f);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<PARTNER>), dim3(blocks), dim3(512), 0, 0, d, src, iters, 1.0f);
  hipDeviceSynchronize();
  float cyc = 0;
  hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
  printf("partner wave: %-40s %6.2f cycles per MFMA\n", name, cyc / ((double)iters * 20));
  hipFree(d);
  hipFree(src);
}

int main() {
  run<0>("none");
  run<4>("s_sleep");
  run<1>("13 x ds_read_b128 bursts");
  run<2>("full-rate VALU (v_fma)");
  run<5>("ds_read bursts + VALU");
  run<3>("9 x LDS-DMA 1 KB bursts + vmcnt(0)");
  return 0;
}

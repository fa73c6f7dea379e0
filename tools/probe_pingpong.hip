// Ping-pong skeleton of ca_gemm_pq.h without the GEMM: 8 waves = 2 groups one barrier apart, each iteration
//   [13 x ds_read_b128 ; wait] barrier [40 independent MFMAs] barrier
// MODE 0: that alone.  MODE 1: + every wave issues 10 LDS-DMA pieces (1 KB each, from a 64 MB global buffer) behind its reads,
// never waited for inside the loop (the operand stream of the real kernel).  Reports s_memtime ticks of an MFMA segment and of a
// whole iteration for wave 0, and ticks per ns (wall clock from hipEvents): is the MFMA segment slower beside the stream, and
// does the shader clock drop?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_pingpong.hip -o tools/probe_pingpong.bin && tools/probe_pingpong.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, const unsigned* src, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[144 * 1024];
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wr = wid >> 2;
  for (int i = threadIdx.x; i < 36 * 1024; i += 512) reinterpret_cast<unsigned*>(lds)[i] = i;
  __syncthreads();
  f32x4 acc[40];
  for (int i = 0; i < 40; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 64u << 20, 0x00020000);
  long long seg = 0, t_begin = 0;
  if (wr == 1) __builtin_amdgcn_s_barrier();
  t_begin = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    u32x4 fa[8], fb[5];
    const unsigned char* base = lds + ((it & 1) * 72 * 1024) + lane * 16;
#pragma unroll
    for (int q = 0; q < 8; ++q) fa[q] = *reinterpret_cast<const u32x4*>(base + q * 2048 + wr * 16384);
#pragma unroll
    for (int q = 0; q < 5; ++q) fb[q] = *reinterpret_cast<const u32x4*>(base + 32768 + q * 2048 + (wid & 3) * 10240);
    if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < 10; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + ((it + 1) & 1) * 72 * 1024 + (wid * 9 + (q % 9)) * 1024), 16,
                                                 (unsigned)(lane * 16), (unsigned)((((blockIdx.x * 977 + it * 8 + wid) * 10 + q) & 0xFFFF) * 1024), 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const long long t0 = __builtin_readcyclecounter();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j)
        acc[i * 5 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fb[j]), __builtin_bit_cast(f16x8, fa[i]), acc[i * 5 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    seg += __builtin_readcyclecounter() - t0;
    __builtin_amdgcn_s_barrier();
  }
  const long long t_end = __builtin_readcyclecounter();
  if (wr == 0) __builtin_amdgcn_s_barrier();
  if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0;
  for (int i = 0; i < 40; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x + 4] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = (float)seg;
    out[1] = (float)(t_end - t_begin);
  }
}

template <int MODE>
void run(const char* name) {
  float* d;
  unsigned* src;
  const int blocks = 256, iters = 3000;
  hipMalloc(&d, sizeof(float) * (blocks * 512 + 8));
  hipMalloc(&src, 65u << 20);
  hipMemset(src, 1, 65u << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 0, 0, d, src, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(512), 0, 0, d, src, iters, 1.0f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0, v[2] = {0, 0};
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(v, d, 8, hipMemcpyDeviceToHost);
  printf("%-46s MFMA segment %7.1f ticks (%5.2f per MFMA), iteration %7.1f ticks, %5.3f ticks/ns, %6.1f TFLOP/s chip\n", name, v[0] / iters, v[0] / iters / 40, v[1] / iters, v[1] / (ms * 1e6),
         16384.0 * 40 * iters * 8 * 256 / (ms * 1e-3) / 1e12);
  hipFree(d);
  hipFree(src);
}

int main() {
  run<0>("reads | MFMA ping-pong");
  run<1>("reads | MFMA ping-pong + 10 KB LDS-DMA per wave");
  run<0>("reads | MFMA ping-pong (again)");
  return 0;
}

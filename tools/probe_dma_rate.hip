// Probe: sustained global -> LDS rate of `buffer_load_dwordx4 ... lds` (1 KiB per wave instruction) per CU, all CUs busy, as a
// function of the footprint (L2-resident / MALL / HBM) and of the pieces each wave keeps in flight.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_dma_rate.hip -o /tmp/probe_dma_rate && /tmp/probe_dma_rate
// Layout mimics a K tile of k_gemm_pq: a piece = 8 rows x 128 bytes, rows `ld` bytes apart.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int INFLIGHT>
__global__ __launch_bounds__(512) void k(const unsigned char* src, unsigned bytes, unsigned ld, int iters, unsigned region_mask, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  // the block walks its own stream of pieces; consecutive blocks start 9 pieces * 8 waves apart
  unsigned piece = (blockIdx.x * 8u + wid) * 9u;
  const unsigned lane_off = (unsigned)(lane >> 3) * ld + (unsigned)(lane & 7) * 16u;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const unsigned p = (piece + i);
      // piece p: rows 8p..8p+7 of a [rows][ld] matrix, column block (p >> 16) & ... kept simple: 128-byte column 0
      const unsigned off = ((p * 8u * ld) & region_mask) + lane_off;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + ((it & 1) * 72 + wid * 9 + i) * 1024), 16, off, 0, 0, 0);
    }
    piece += 8u * 9u * gridDim.x;
    if (INFLIGHT == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if (INFLIGHT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // INFLIGHT == 18: two units in flight
    else if (it & 1) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const size_t total = 1ull << 31;  // 2 GiB
  unsigned char* d;
  unsigned long long* dc;
  CK(hipMalloc(&d, total));
  CK(hipMemset(d, 1, total));
  CK(hipMalloc(&dc, 256 * 8));
  const int iters = 400;
  CK(hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456));
  CK(hipFuncSetAttribute((const void*)k<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456));
  CK(hipFuncSetAttribute((const void*)k<18>, hipFuncAttributeMaxDynamicSharedMemorySize, 147456));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int fl : {0, 9, 18}) {
    for (unsigned region_log2 : {21u, 24u, 27u, 30u}) {
      const unsigned mask = (1u << region_log2) - 1u;
      for (unsigned ld : {128u, 5120u}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0));
          if (fl == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 147456, 0, d, (unsigned)(total - 1), ld, iters, mask, dc);
          else if (fl == 9) hipLaunchKernelGGL(k<9>, dim3(256), dim3(512), 147456, 0, d, (unsigned)(total - 1), ld, iters, mask, dc);
          else hipLaunchKernelGGL(k<18>, dim3(256), dim3(512), 147456, 0, d, (unsigned)(total - 1), ld, iters, mask, dc);
          CK(hipGetLastError());
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
        }
        unsigned long long hc[256];
        CK(hipMemcpy(hc, dc, sizeof(hc), hipMemcpyDeviceToHost));
        double avg = 0;
        for (int i = 0; i < 256; ++i) avg += (double)hc[i];
        avg /= 256;
        const double bytes_per_block = (double)iters * 72.0 * 1024.0;
        printf("in flight per wave %2d pieces, footprint 2^%u B, row pitch %5u B: %7.1f us, %6.2f TB/s chip, %5.1f B/clk/CU (s_memtime-class counter %.0f ticks per 72 KB unit)\n", fl == 0 ? 9 : fl == 9 ? 18 : 27,
               region_log2, ld, best * 1e3, bytes_per_block * 256 / best / 1e9, bytes_per_block / avg, avg / iters);
      }
    }
  }
  return 0;
}

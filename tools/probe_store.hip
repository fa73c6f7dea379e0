// Store rate of the chip against the cache-policy bits of buffer_store (gfx950: aux 1 = sc0, 2 = nt, 16 = sc1) and the access shape.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_store tools/probe_store.hip && /tmp/probe_store
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__global__ __launch_bounds__(256) void k_store(unsigned char* dst, size_t bytes, int rows_per_block) {
  // row-major [rows][ROW bytes]; a block writes `rows_per_block` consecutive rows of 2560 B (the N = 1280 output of the K = 320 GEMMs)
  const unsigned ROW = 2560;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (unsigned)(bytes > 0xFFFFFF00ull ? 0xFFFFFF00ull : bytes), 0x00020000);
  const u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  const size_t rows = bytes / ROW;
  for (size_t r0 = (size_t)blockIdx.x * rows_per_block; r0 < rows; r0 += (size_t)gridDim.x * rows_per_block)
    for (unsigned i = threadIdx.x; i < (unsigned)rows_per_block * (ROW / 16); i += 256) {
      const unsigned off = (unsigned)(r0 * ROW) + i * 16u;
      __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, AUX);
    }
}

__global__ __launch_bounds__(256) void k_read(const u32x4* src, size_t n, unsigned* out) {
  u32x4 a = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += src[i];
  if (a[0] + a[1] + a[2] + a[3] == 0x12345678u) out[0] = 1;
}

int main() {
  const size_t bytes = 335544320;  // 131072 x 1280 x 2
  unsigned char* d;
  unsigned* o;
  hipMalloc(&d, bytes);
  hipMalloc(&o, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.1f us  %5.2f TB/s\n", name, ms * 100, bytes / (ms * 1e-4) * 1e-12);
  };
  for (int rep = 0; rep < 2; ++rep) {
    run("store aux 0", [&] { hipLaunchKernelGGL(k_store<0>, dim3(2048), dim3(256), 0, 0, d, bytes, 128); });
    run("store aux 1 (sc0)", [&] { hipLaunchKernelGGL(k_store<1>, dim3(2048), dim3(256), 0, 0, d, bytes, 128); });
    run("store aux 2 (nt)", [&] { hipLaunchKernelGGL(k_store<2>, dim3(2048), dim3(256), 0, 0, d, bytes, 128); });
    run("store aux 3 (sc0 nt)", [&] { hipLaunchKernelGGL(k_store<3>, dim3(2048), dim3(256), 0, 0, d, bytes, 128); });
    run("store aux 16 (sc1)", [&] { hipLaunchKernelGGL(k_store<16>, dim3(2048), dim3(256), 0, 0, d, bytes, 128); });
    run("store aux 17 (sc0 sc1)", [&] { hipLaunchKernelGGL(k_store<17>, dim3(2048), dim3(256), 0, 0, d, bytes, 128); });
    run("store aux 18 (sc1 nt)", [&] { hipLaunchKernelGGL(k_store<18>, dim3(2048), dim3(256), 0, 0, d, bytes, 128); });
    run("store aux 19 (sc0 sc1 nt)", [&] { hipLaunchKernelGGL(k_store<19>, dim3(2048), dim3(256), 0, 0, d, bytes, 128); });
    run("store aux 0, 512 blocks", [&] { hipLaunchKernelGGL(k_store<0>, dim3(512), dim3(256), 0, 0, d, bytes, 128); });
    run("store aux 0, 8 rows per block", [&] { hipLaunchKernelGGL(k_store<0>, dim3(2048), dim3(256), 0, 0, d, bytes, 8); });
    run("read", [&] { hipLaunchKernelGGL(k_read, dim3(4096), dim3(256), 0, 0, (const u32x4*)d, bytes / 16, o); });
    run("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, bytes, 0); });
  }
  return 0;
}

// Probe: does `buffer_load_dwordx4 ... lds` write ZEROS into LDS for out-of-range lanes?
// (needed to use LDS-DMA for the zero halo of the implicit-GEMM conv).  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/probe_glds.hip -o /tmp/probe_glds && /tmp/probe_glds
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* a, unsigned short* out, int ld, unsigned nbytes) {
  __shared__ __attribute__((aligned(16))) unsigned short sm[128 * 64];
  const int tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 128 * 64; i += 256) sm[i] = 0xABCD;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, nbytes, 0x00020000);
  for (int i = 0; i < 4; ++i) {
    int row = wid * 32 + i * 8 + (lane >> 3);
    int chunk = (lane & 7) ^ ((row >> 1) & 7);
    unsigned off = (row % 3 == 1) ? 0xFFFFFFF0u : (unsigned)((row * ld + chunk * 8) * 2);
    unsigned short* l = sm + (wid * 32 + i * 8) * 64;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)l, 16, off, 0, 0, 0);
  }
  __syncthreads();
  for (int i = tid; i < 128 * 64; i += 256) out[i] = sm[i];
}
int main() {
  const int rows = 128, ld = 64;
  std::vector<unsigned short> h(rows * ld), o(rows * ld);
  for (int i = 0; i < rows * ld; ++i) h[i] = (unsigned short)(i & 0x7fff);
  unsigned short *da, *dout;
  hipMalloc(&da, h.size() * 2); hipMalloc(&dout, o.size() * 2);
  hipMemcpy(da, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, da, dout, ld, (unsigned)(h.size() * 2));
  hipMemcpy(o.data(), dout, o.size() * 2, hipMemcpyDeviceToHost);
  int bad_zero = 0, bad_data = 0, stale = 0;
  for (int r = 0; r < rows; ++r) for (int cp = 0; cp < 8; ++cp) for (int j = 0; j < 8; ++j) {
    int c = cp ^ ((r >> 1) & 7);                      // LDS chunk position cp holds global chunk c
    unsigned short got = o[r * 64 + cp * 8 + j];
    if (r % 3 == 1) { if (got == 0xABCD) ++stale; else if (got != 0) ++bad_zero; }
    else if (got != h[r * ld + c * 8 + j]) ++bad_data;
  }
  printf("glds probe: oob_stale=%d oob_nonzero=%d data_mismatch=%d -> %s\n", stale, bad_zero, bad_data,
         (stale || bad_zero || bad_data) ? "FAIL" : "OK (OOB lanes write zeros; swizzled source lands lane-linear)");
  return 0;
}

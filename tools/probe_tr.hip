// Probe the semantics of ds_read_b64_tr_b16 (gfx950 LDS transpose read).
// Hypothesis: within each 16-lane group, lane i supplies the address of the i-th 8-byte piece of a
// [4 rows][16 cols] b16 block (piece p = row p/4, cols 4*(p%4)..+3); lane i receives column i:
// result[i][j] = block[row j][col i].
//   hipcc --offload-arch=gfx950 -O3 tools/probe_tr.hip -o /tmp/probe_tr && /tmp/probe_tr
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned short* out, int row_stride /*elements*/) {
  __shared__ __attribute__((aligned(16))) unsigned short sm[64 * 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 64; i += 64) sm[i] = (unsigned short)i;  // value = element index
  __syncthreads();
  const int grp = lane >> 4, p = lane & 15;
  // group g reads the block rows 4g..4g+3, cols 0..15 of a [rows][row_stride] matrix
  const unsigned base = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned short*)sm;  // LDS byte offset (keeps sm alive)
  const unsigned addr = base + (unsigned)(((4 * grp + p / 4) * row_stride + 4 * (p % 4)) * 2);
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
  out[lane * 4 + 0] = r[0] & 0xffff;
  out[lane * 4 + 1] = r[0] >> 16;
  out[lane * 4 + 2] = r[1] & 0xffff;
  out[lane * 4 + 3] = r[1] >> 16;
}
int main() {
  unsigned short* d; unsigned short h[256];
  hipMalloc(&d, sizeof(h));
  for (int rs : {16, 64, 72}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, rs);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 4; ++j) {
      int grp = lane >> 4, i = lane & 15;
      int expect = (4 * grp + j) * rs + i;  // block[row j][col i]
      if (h[lane * 4 + j] != expect) ++bad;
    }
    printf("row_stride=%d: mismatches=%d  lane0={%d,%d,%d,%d} lane1={%d,%d,%d,%d} lane17={%d,%d,%d,%d}\n", rs, bad, h[0], h[1], h[2], h[3],
           h[4], h[5], h[6], h[7], h[68], h[69], h[70], h[71]);
  }
  return 0;
}

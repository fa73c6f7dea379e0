// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, in the access shapes this package's
// kernels use (MI355X_MICROARCH.md, HBM section: only the 16-byte coalesced read is calibrated -- x2 -- everything else is not).
// Every kernel moves exactly 1 GiB (4x the 256 MiB Infinity Cache) once per launch, so bytes / counter is the calibration factor.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_pmc_calib tools/probe_pmc_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -f csv -d out/f -o run -- /tmp/probe_pmc_calib      (and the same with WRITE_SIZE)
//   python tools/pmc_calib_summary.py out/f/.../run_counter_collection.csv out/w/.../run_counter_collection.csv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr size_t BYTES = 1ull << 30;

template <typename T>
__global__ __launch_bounds__(256) void k_calib_read(const T* __restrict__ src, size_t n, unsigned* out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const T v = src[i];
    const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
    acc += b[0] + b[sizeof(T) - 1];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// global -> LDS DMA, 16 bytes per lane (the operand path of every GEMM / attention kernel here)
__global__ __launch_bounds__(256) void k_calib_read_lds_dma16(const unsigned char* src, size_t bytes, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 1024];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)bytes, 0x00020000);
  const unsigned wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned acc = 0;
  for (size_t base = (size_t)blockIdx.x * 4096; base < bytes; base += (size_t)gridDim.x * 4096) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wid * 1024), 16,
                                             (unsigned)base + wid * 1024u + lane * 16u, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += *reinterpret_cast<const unsigned*>(smem + threadIdx.x * 16);
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// rows of 640 B (C = 320 activations) of which a (wave, head) touches 80 B: the q / o pieces of the attention kernels
__global__ __launch_bounds__(256) void k_calib_read_pieces80(const unsigned char* src, size_t bytes, unsigned* out) {
  // a block reads whole 640-byte rows, but as 8 passes of 80-byte pieces (8-byte lanes, 10 lanes per piece)
  const size_t rows = bytes / 640;
  unsigned acc = 0;
  for (size_t r0 = (size_t)blockIdx.x * 25; r0 < rows; r0 += (size_t)gridDim.x * 25)
    for (int h = 0; h < 8; ++h) {
      const unsigned t = threadIdx.x;
      if (t < 250) {
        const size_t r = r0 + t / 10;
        if (r < rows) acc += reinterpret_cast<const u32x2*>(src + r * 640 + h * 80)[t % 10][0];
      }
    }
  if (acc == 0x12345678u) out[0] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void k_calib_write(T* __restrict__ dst, size_t n) {
  T v;
  unsigned char* b = reinterpret_cast<unsigned char*>(&v);
  for (unsigned i = 0; i < sizeof(T); ++i) b[i] = (unsigned char)(threadIdx.x + i);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}

// 8-byte stores of 80-byte pieces per (row, head): the o stores of k_attn_short / k_tattn_fused / k_xattn_fused
__global__ __launch_bounds__(256) void k_calib_write_pieces80(unsigned char* dst, size_t bytes) {
  const size_t rows = bytes / 640;
  const u32x2 v = {threadIdx.x, blockIdx.x};
  for (size_t r0 = (size_t)blockIdx.x * 25; r0 < rows; r0 += (size_t)gridDim.x * 25)
    for (int h = 0; h < 8; ++h) {
      const unsigned t = threadIdx.x;
      if (t < 250) {
        const size_t r = r0 + t / 10;
        if (r < rows) reinterpret_cast<u32x2*>(dst + r * 640 + h * 80)[t % 10] = v;
      }
    }
}

// 64-byte half lines: a wave's 16-byte stores cover 64 consecutive bytes of each of 16 rows (the epilogue stores of k_gemm_ar)
__global__ __launch_bounds__(256) void k_calib_write_half_lines(unsigned char* dst, size_t bytes) {
  const size_t rows = bytes / 2560;  // [rows][2560 B]
  const u32x4 v = {threadIdx.x, blockIdx.x, 1u, 2u};
  const unsigned lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (size_t r0 = (size_t)blockIdx.x * 16; r0 < rows; r0 += (size_t)gridDim.x * 16)
    for (unsigned c = wid; c < 40; c += 4) {  // 40 half lines per row
      const size_t r = r0 + (lane & 15);
      if (r < rows) *reinterpret_cast<u32x4*>(dst + r * 2560 + c * 64 + (lane >> 4) * 16) = v;
    }
}

int main() {
  unsigned char* d;
  unsigned* o;
  if (hipMalloc(&d, BYTES) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
  hipMemset(d, 1, BYTES);
  hipDeviceSynchronize();
  const dim3 g(4096), b(256);
  // each kernel once, cold with respect to the previous one (1 GiB >> 256 MiB of Infinity Cache)
  hipLaunchKernelGGL(k_calib_read<u32x4>, g, b, 0, 0, (const u32x4*)d, BYTES / 16, o);
  hipLaunchKernelGGL(k_calib_read<u32x2>, g, b, 0, 0, (const u32x2*)d, BYTES / 8, o);
  hipLaunchKernelGGL(k_calib_read<unsigned>, g, b, 0, 0, (const unsigned*)d, BYTES / 4, o);
  hipLaunchKernelGGL(k_calib_read<unsigned short>, g, b, 0, 0, (const unsigned short*)d, BYTES / 2, o);
  hipLaunchKernelGGL(k_calib_read_lds_dma16, g, b, 0, 0, d, BYTES, o);
  hipLaunchKernelGGL(k_calib_read_pieces80, g, b, 0, 0, d, BYTES / 640 * 640, o);
  hipLaunchKernelGGL(k_calib_write<u32x4>, g, b, 0, 0, (u32x4*)d, BYTES / 16);
  hipLaunchKernelGGL(k_calib_write<u32x2>, g, b, 0, 0, (u32x2*)d, BYTES / 8);
  hipLaunchKernelGGL(k_calib_write<unsigned>, g, b, 0, 0, (unsigned*)d, BYTES / 4);
  hipLaunchKernelGGL(k_calib_write<unsigned short>, g, b, 0, 0, (unsigned short*)d, BYTES / 2);
  hipLaunchKernelGGL(k_calib_write_pieces80, g, b, 0, 0, d, BYTES / 640 * 640);
  hipLaunchKernelGGL(k_calib_write_half_lines, g, b, 0, 0, d, BYTES / 2560 * 2560);
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  printf("probe_pmc_calib: 12 kernels, %zu bytes each\n", BYTES);
  return 0;
}

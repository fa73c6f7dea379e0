// Probe: can a wave learn that its LDS-DMA pieces have LANDED without `s_waitcnt vmcnt`?
//
// Why: on gfx950 stores and loads share the VMEM counter and retire out of order with respect to each other, so a wave
// that streams operands by LDS-DMA and also stores results can only wait for "everything but the N youngest LOADS" by
// also waiting for its stores -- which is what makes a persistent GEMM pay for its epilogue stores.
// Idea: loads return IN ORDER among themselves (that is what a counted vmcnt relies on), so a 4-byte LDS-DMA issued
// AFTER the data pieces, fetching a sequence number from a small global table into an LDS flag word, lands after them;
// the wave polls that LDS word (ds_read) instead of waiting on the counter, and its stores stay fire-and-forget.
//
// The probe streams 4 KB slots (4 data pieces + 1 flag piece per slot, two slots in flight) from a buffer larger than
// the Infinity Cache in every wave of a full grid, with 8 un-waited 16-byte stores per slot in between, polls the flag and
// then checks EVERY word of the slot against the generator.  mode 0 = flag polling, mode 1 = the unsafe counted vmcnt
// (stores counted as "younger, may stay outstanding": expected to fail sometimes), mode 2 = vmcnt(0) (reference).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_flag.hip -o /tmp/probe_flag && /tmp/probe_flag
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned gen(unsigned i) { return i * 2654435761u ^ 0x9e3779b9u; }

__global__ void k_fill(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = gen((unsigned)i);
}
__global__ void k_table(unsigned* t) { t[threadIdx.x] = threadIdx.x; }

template <int MODE>
__global__ __launch_bounds__(512, 2) void k_probe(const unsigned* src, unsigned src_bytes, const unsigned* table, unsigned* sink, unsigned sink_bytes,
                                                  int iters, unsigned long long* result) {
  constexpr int SLOT = 4096, NSLOT = 2;
  constexpr int FLAG_BASE = 8 * NSLOT * SLOT;  // per wave, per slot: 256 B
  __shared__ __attribute__((aligned(16))) unsigned char smem[FLAG_BASE + 8 * NSLOT * 256];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 8 * NSLOT * 64; i += 512) reinterpret_cast<unsigned*>(smem + FLAG_BASE)[i] = 0xFFFFFFFFu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, 4096, 0x00020000);
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)sink, 0, sink_bytes, 0x00020000);
  unsigned char* ring = smem + wid * NSLOT * SLOT;
  unsigned char* flags = smem + FLAG_BASE + wid * NSLOT * 256;
  const unsigned nslots_src = src_bytes / SLOT;
  unsigned rng = (blockIdx.x * 8u + wid) * 747796405u + 2891336453u;
  auto next_slot = [&]() {
    rng = rng * 1664525u + 1013904223u;
    return (rng >> 4) % nslots_src;
  };
  unsigned long long bad = 0, polls = 0, timeouts = 0;
  unsigned pend_src[NSLOT];
  auto issue = [&](int s, unsigned seq) {
    const unsigned sl = next_slot();
    pend_src[s] = sl;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ring + s * SLOT + q * 1024), 16, (unsigned)lane * 16u, sl * SLOT + q * 1024, 0, 0);
    if (MODE == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (__attribute__((address_space(3))) void*)(flags + s * 256), 4, 0u, (seq & 1023u) * 4u, 0, 0);
  };
  auto stores = [&](unsigned seq) {  // 8 fire-and-forget 16-byte stores (the "epilogue")
    const unsigned base = (((blockIdx.x * 8u + wid) * 64u + (seq & 63u)) * 8u) * 1024u;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const u32x4 v = {seq, (unsigned)q, (unsigned)lane, rng};
      __builtin_amdgcn_raw_buffer_store_b128(v, rk, (unsigned)lane * 16u, (base + q * 1024u) % sink_bytes, 0);
    }
  };
  issue(0, 1);
  for (int it = 0; it < iters; ++it) {
    const int s = it & 1;
    const unsigned seq = (unsigned)it + 1u;
    stores(seq);
    issue(s ^ 1, seq + 1);
    // ---- wait for slot s
    if (MODE == 0) {
      const unsigned fl = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(flags + s * 256);
      unsigned spins = 0;
      while (true) {
        unsigned fv;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(fv) : "v"(fl) : "memory");
        const unsigned v = __builtin_amdgcn_readfirstlane(fv);
        ++polls;
        if (v == (seq & 1023u)) break;
        if (++spins > (1u << 22)) {
          ++timeouts;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    } else if (MODE == 1) {
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // 4 pieces of the next slot + the 8 stores issued since: UNSAFE
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // ---- check every word of the slot
    const unsigned sl = pend_src[s];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(ring + s * SLOT + q * 1024 + lane * 16);
      const unsigned w0 = (sl * SLOT + q * 1024 + lane * 16) / 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) bad += v[r] != gen(w0 + r);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (bad) atomicAdd(result + 0, bad);
  if (lane == 0) {
    atomicAdd(result + 1, polls);
    if (timeouts) atomicAdd(result + 2, timeouts);
  }
}

int main(int argc, char** argv) {
  const size_t src_bytes = (size_t)1 << 30;  // 1 GiB: beyond the 256 MiB Infinity Cache
  const unsigned sink_bytes = 1u << 30;
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned *src, *table, *sink;
  unsigned long long* res;
  hipMalloc(&src, src_bytes);
  hipMalloc(&table, 4096);
  hipMalloc(&sink, sink_bytes);
  hipMalloc(&res, 64);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, src, src_bytes / 4);
  hipLaunchKernelGGL(k_table, dim3(1), dim3(1024), 0, 0, table);
  hipDeviceSynchronize();
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(res, 0, 64);
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k_probe<0>, dim3(512), dim3(512), 0, 0, src, (unsigned)src_bytes, table, sink, sink_bytes, iters, res);
      if (mode == 1) hipLaunchKernelGGL(k_probe<1>, dim3(512), dim3(512), 0, 0, src, (unsigned)src_bytes, table, sink, sink_bytes, iters, res);
      if (mode == 2) hipLaunchKernelGGL(k_probe<2>, dim3(512), dim3(512), 0, 0, src, (unsigned)src_bytes, table, sink, sink_bytes, iters, res);
      hipEventRecord(e1);
      hipError_t err = hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[3];
      hipMemcpy(h, res, 24, hipMemcpyDeviceToHost);
      const double slots = 512.0 * 8 * iters;
      printf("flag probe mode %d (%s) rep %d: %s  mismatched words=%llu  polls/slot=%.2f  timeouts=%llu  %.2f ms  %.2f TB/s loads + %.2f TB/s stores\n", mode,
             mode == 0 ? "LDS flag polling" : mode == 1 ? "UNSAFE counted vmcnt over stores" : "vmcnt(0)", rep, err == hipSuccess ? "ran" : hipGetErrorString(err), h[0],
             h[1] / slots, h[2], ms, slots * 4096 / ms / 1e9, slots * 8192 / ms / 1e9);
    }
  }
  return 0;
}

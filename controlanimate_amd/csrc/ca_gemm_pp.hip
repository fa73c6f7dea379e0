// Ping-pong GEMM kernels: separate translation unit (compile time), see ca_gemm_pp.h.
#include "ca_gemm_core.h"
#include <type_traits>

namespace {
using namespace ca_gemm_detail;
#include "ca_gemm_pp2.h"
#include "ca_gemm_wres.h"
#include "ca_gemm_ps.h"
#include "ca_gemm_pq.h"
#ifdef CA_EXPERIMENTS  // round-2 experiments that never became defaults: csrc/experiments/, not in the product library
#include "experiments/ca_gemm_pp.h"
#include "experiments/ca_gemm_pp3.h"
#endif

int cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int DT, int MODE>
int launch_pp(const GemmKParams& p, int bn, unsigned tiles, hipStream_t st) {
#ifdef CA_EXPERIMENTS
  if (bn == 321) {  // persistent 128 x 320 kernel with the pipelined epilogue
    const int64_t ncols = p.geglu ? p.n / 2 : p.n;
    const unsigned c_bytes = (unsigned)((((int64_t)p.m - 1) * p.ldc + ncols) * 2);
    const unsigned res_bytes = p.res ? (unsigned)((((int64_t)p.m - 1) * p.ld_res + p.n) * 2) : 0u;
    const unsigned grid = tiles < (unsigned)cu_count() ? tiles : (unsigned)cu_count();
    if (p.dbg == 9) {  // timing experiment: shader-clock stamps of block 0 (see ca_gemm_pp3.h), printed to stderr
      static unsigned long long* dbuf = nullptr;
      if (!dbuf && hipMalloc(&dbuf, 4096 * 8) != hipSuccess) return CA_ERR_LAUNCH;
      (void)hipMemsetAsync(dbuf, 0, 4096 * 8, st);
      GemmKParams q = p;
      q.partial = reinterpret_cast<float*>(dbuf);
      hipLaunchKernelGGL((k_gemm_pp3<DT, MODE>), dim3(grid), dim3(512), 0, st, q, (int)tiles, c_bytes, res_bytes);
      (void)hipStreamSynchronize(st);
      static int printed = 0;
      if (printed++ < 1) {
        static unsigned long long host[4096];
        (void)hipMemcpy(host, dbuf, sizeof(host), hipMemcpyDeviceToHost);
        for (int g = 0; g < 2; ++g) {
          fprintf(stderr, "[pp3 stamps group %d] tag:delta_cycles ...\n", g);
          unsigned long long prev = host[g * 2048];
          for (int i = 0; i < 1000 && host[g * 2048 + 2 * i]; ++i) {
            fprintf(stderr, "%llu:%llu ", host[g * 2048 + 2 * i + 1], host[g * 2048 + 2 * i] - prev);
            prev = host[g * 2048 + 2 * i];
            if (host[g * 2048 + 2 * i + 1] == 12) fprintf(stderr, "| ");
            if (host[g * 2048 + 2 * i + 1] == 14) fprintf(stderr, "\n");
          }
          fprintf(stderr, "\n");
        }
      }
      return CA_OK;
    }
    hipLaunchKernelGGL((k_gemm_pp3<DT, MODE>), dim3(grid), dim3(512), 0, st, p, (int)tiles, c_bytes, res_bytes);
    return CA_OK;
  }
  if (bn == 256 || bn == 128) {
    if (bn == 256) hipLaunchKernelGGL((k_gemm_pp<DT, MODE, 256>), dim3(tiles), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((k_gemm_pp<DT, MODE, 128>), dim3(tiles), dim3(512), 0, st, p);
    return CA_OK;
  }
#endif
  if (bn == 323) {  // persistent streaming kernel, 256 x 320 tiles (ca_gemm_pq.h)
    const unsigned c_bytes = (unsigned)((((int64_t)p.m - 1) * p.ldc + p.n) * 2);
    const unsigned res_bytes = p.res ? (unsigned)((((int64_t)p.m - 1) * p.ld_res + p.n) * 2) : 0u;
    const unsigned grid = tiles < (unsigned)cu_count() ? tiles : (unsigned)cu_count();
#ifdef CA_STAMPS
    if (p.dbg == 9) {  // timing experiment: shader-clock stamps of block 0, printed to stderr (first launch only)
      static unsigned long long* dbuf = nullptr;
      if (!dbuf && hipMalloc(&dbuf, 4096 * 8) != hipSuccess) return CA_ERR_LAUNCH;
      (void)hipMemsetAsync(dbuf, 0, 4096 * 8, st);
      GemmKParams q = p;
      q.partial = reinterpret_cast<float*>(dbuf);
      hipLaunchKernelGGL((k_gemm_pq<DT, MODE>), dim3(grid), dim3(512), 0, st, q, (int)tiles, c_bytes, res_bytes);
      (void)hipStreamSynchronize(st);
      static int printed = 0;
      if (printed++ < 1) {
        static unsigned long long host[4096];
        (void)hipMemcpy(host, dbuf, sizeof(host), hipMemcpyDeviceToHost);
        for (int g = 0; g < 2; ++g) {
          fprintf(stderr, "[pq stamps group %d, %dx%dx%d] tag:delta_cycles ...\n", g, p.m, p.n, (p.c1 + p.c2) * p.taps);
          unsigned long long prev = host[g * 2048];
          for (int i = 0; i < 1000 && host[g * 2048 + 2 * i]; ++i) {
            fprintf(stderr, "%llu:%llu ", host[g * 2048 + 2 * i + 1], host[g * 2048 + 2 * i] - prev);
            prev = host[g * 2048 + 2 * i];
            if (host[g * 2048 + 2 * i + 1] == 8) fprintf(stderr, "| ");
            if (host[g * 2048 + 2 * i + 1] == 10) fprintf(stderr, "\n");
          }
          fprintf(stderr, "\n");
        }
      }
      return CA_OK;
    }
#endif
    if (MODE == 0 && (p.geglu || p.ln_colsum || p.ln_stats)) {
      const unsigned cb = p.geglu ? (unsigned)((((int64_t)p.m - 1) * p.ldc + p.n / 2) * 2) : c_bytes;
      hipLaunchKernelGGL((k_gemm_pq<DT, 0, 1>), dim3(grid), dim3(512), 0, st, p, (int)tiles, cb, 0u);
      return CA_OK;
    }
    if (MODE == 0 && p.row_sums) {
      hipLaunchKernelGGL((k_gemm_pq<DT, 0, 2>), dim3(grid), dim3(512), 0, st, p, (int)tiles, c_bytes, res_bytes);
      return CA_OK;
    }
    hipLaunchKernelGGL((k_gemm_pq<DT, MODE>), dim3(grid), dim3(512), 0, st, p, (int)tiles, c_bytes, res_bytes);
    return CA_OK;
  }
  if (bn == 322) {  // persistent streaming kernel, 128 x 320 tiles (ca_gemm_ps.h)
    const int64_t ncols = p.geglu ? p.n / 2 : p.n;
    const unsigned c_bytes = (unsigned)((((int64_t)p.m - 1) * p.ldc + ncols) * 2);
    const unsigned res_bytes = p.res ? (unsigned)((((int64_t)p.m - 1) * p.ld_res + p.n) * 2) : 0u;
    const unsigned grid = tiles < (unsigned)cu_count() ? tiles : (unsigned)cu_count();
#ifdef CA_STAMPS
    if (p.dbg == 9) {  // timing experiment: shader-clock stamps of block 0, printed to stderr (first launch only)
      static unsigned long long* dbuf = nullptr;
      if (!dbuf && hipMalloc(&dbuf, 4096 * 8) != hipSuccess) return CA_ERR_LAUNCH;
      (void)hipMemsetAsync(dbuf, 0, 4096 * 8, st);
      GemmKParams q = p;
      q.partial = reinterpret_cast<float*>(dbuf);
      hipLaunchKernelGGL((k_gemm_ps<DT, MODE>), dim3(grid), dim3(512), 0, st, q, (int)tiles, c_bytes, res_bytes);
      (void)hipStreamSynchronize(st);
      static int printed = 0;
      if (printed++ < 1) {
        static unsigned long long host[4096];
        (void)hipMemcpy(host, dbuf, sizeof(host), hipMemcpyDeviceToHost);
        for (int g = 0; g < 2; ++g) {
          fprintf(stderr, "[ps stamps group %d, %dx%dx%d] tag:delta_cycles ...\n", g, p.m, p.n, (p.c1 + p.c2) * p.taps);
          unsigned long long prev = host[g * 2048];
          for (int i = 0; i < 1000 && host[g * 2048 + 2 * i]; ++i) {
            fprintf(stderr, "%llu:%llu ", host[g * 2048 + 2 * i + 1], host[g * 2048 + 2 * i] - prev);
            prev = host[g * 2048 + 2 * i];
            if (host[g * 2048 + 2 * i + 1] == 8) fprintf(stderr, "| ");
            if (host[g * 2048 + 2 * i + 1] == 10) fprintf(stderr, "\n");
          }
          fprintf(stderr, "\n");
        }
      }
      return CA_OK;
    }
#endif
#ifdef CA_EXPERIMENTS
    static const int ps_flags = CA_KNOB("CA_PS_FLAGS", 1);
    if (!ps_flags) {
      hipLaunchKernelGGL((k_gemm_ps<DT, MODE, false>), dim3(grid), dim3(512), 0, st, p, (int)tiles, c_bytes, res_bytes);
      return CA_OK;
    }
#endif
    hipLaunchKernelGGL((k_gemm_ps<DT, MODE>), dim3(grid), dim3(512), 0, st, p, (int)tiles, c_bytes, res_bytes);
    return CA_OK;
  }
  if (bn == 160) {  // weight-resident streaming kernel (K = 320, dense only)
    if (MODE != 0) return CA_ERR_LAUNCH;
    const int panels = p.n / 160;
    const int chunks = (p.m + 255) / 256;
    int per = (cu_count() / 8) / panels;  // slab lanes per XCD
    if (per * 8 > chunks) per = chunks / 8;
    if (per < 1) per = 1;
    const unsigned rb_bytes = p.rowbias ? (unsigned)(((int64_t)((p.m - 1) / p.rows_per_group) * p.ld_rowbias + p.n) * 4) : 16u;
    const unsigned c_bytes = (unsigned)((((int64_t)p.m - 1) * p.ldc + (p.geglu ? p.n / 2 : p.n)) * 2);
    const unsigned res_bytes = p.res ? (unsigned)((((int64_t)p.m - 1) * p.ld_res + p.n) * 2) : 0u;
    hipLaunchKernelGGL((k_gemm_wres<DT>), dim3(8 * per * panels), dim3(512), 0, st, p, panels, 8 * per, chunks, rb_bytes, c_bytes, res_bytes);
    return CA_OK;
  }
  if (bn != 320) return CA_ERR_LAUNCH;
  hipLaunchKernelGGL((k_gemm_pp2<DT, MODE>), dim3(tiles), dim3(512), 0, st, p);
  return CA_OK;
}
}  // namespace

int ca_launch_gemm_pp(const ca_gemm_detail::GemmKParams& p0, int dtype, int mode, int bn, unsigned tiles, hipStream_t st) {
  static const int dbg = CA_KNOB("CA_PP_DBG", 0);  // (timing experiments: 1 = no epilogue, 2 = no main loop)
  ca_gemm_detail::GemmKParams p = p0;
  p.dbg = dbg;
  if (dtype == CA_BF16) return mode ? launch_pp<CA_BF16, 1>(p, bn, tiles, st) : launch_pp<CA_BF16, 0>(p, bn, tiles, st);
  return mode ? launch_pp<CA_F16, 1>(p, bn, tiles, st) : launch_pp<CA_F16, 0>(p, bn, tiles, st);
}

// Ping-pong GEMM kernels: separate translation unit (compile time), see ca_gemm_pp.h.
#include "ca_gemm_core.h"
#include <type_traits>

namespace {
using namespace ca_gemm_detail;
#include "ca_gemm_pp.h"
#include "ca_gemm_pp2.h"
#include "ca_gemm_pp3.h"

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
  if (bn == 321) {  // persistent 128 x 320 kernel with the pipelined epilogue
    const int64_t ncols = p.geglu ? p.n / 2 : p.n;
    const unsigned c_bytes = (unsigned)((((int64_t)p.m - 1) * p.ldc + ncols) * 2);
    const unsigned res_bytes = p.res ? (unsigned)((((int64_t)p.m - 1) * p.ld_res + p.n) * 2) : 0u;
    const unsigned grid = tiles < (unsigned)cu_count() ? tiles : (unsigned)cu_count();
    hipLaunchKernelGGL((k_gemm_pp3<DT, MODE>), dim3(grid), dim3(512), 0, st, p, (int)tiles, c_bytes, res_bytes);
    return CA_OK;
  }
  if (bn == 320) hipLaunchKernelGGL((k_gemm_pp2<DT, MODE>), dim3(tiles), dim3(512), 0, st, p);
  else if (bn == 256) hipLaunchKernelGGL((k_gemm_pp<DT, MODE, 256>), dim3(tiles), dim3(512), 0, st, p);
  else hipLaunchKernelGGL((k_gemm_pp<DT, MODE, 128>), dim3(tiles), dim3(512), 0, st, p);
  return CA_OK;
}
}  // namespace

int ca_launch_gemm_pp(const ca_gemm_detail::GemmKParams& p0, int dtype, int mode, int bn, unsigned tiles, hipStream_t st) {
  static const int dbg = getenv("CA_PP_DBG") ? atoi(getenv("CA_PP_DBG")) : 0;
  ca_gemm_detail::GemmKParams p = p0;
  p.dbg = dbg;
  if (dtype == CA_BF16) return mode ? launch_pp<CA_BF16, 1>(p, bn, tiles, st) : launch_pp<CA_BF16, 0>(p, bn, tiles, st);
  return mode ? launch_pp<CA_F16, 1>(p, bn, tiles, st) : launch_pp<CA_F16, 0>(p, bn, tiles, st);
}

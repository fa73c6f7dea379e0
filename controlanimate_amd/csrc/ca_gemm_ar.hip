// Activation-resident K = 320 GEMM (ca_gemm_ar.h): separate translation unit (compile time).
#include "ca_gemm_core.h"

namespace {
using namespace ca_gemm_detail;
#include "ca_gemm_ar.h"

int ar_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

template <int DT>
int launch_ar(const GemmKParams& p, hipStream_t st) {
  const int tiles_m = (p.m + 127) / 128;
  const unsigned c_bytes = (unsigned)((((int64_t)p.m - 1) * p.ldc + (p.geglu ? p.n / 2 : p.n)) * 2);
  const unsigned res_bytes = p.res ? (unsigned)((((int64_t)p.m - 1) * p.ld_res + p.n) * 2) : 0u;
  const int slots = 2 * ar_cu_count();  // two blocks (80 KB of LDS, four waves each) per CU
  const unsigned grid = (unsigned)(tiles_m < slots ? tiles_m : slots);
  float* ws = p.ln_inline ? p.partial : nullptr;
  if (p.geglu) hipLaunchKernelGGL((k_gemm_ar<DT, 2>), dim3(grid), dim3(256), 0, st, p, p.wf, tiles_m, c_bytes, 0u, ws);
  else if (p.ln_colsum) hipLaunchKernelGGL((k_gemm_ar<DT, 1>), dim3(grid), dim3(256), 0, st, p, p.wf, tiles_m, c_bytes, 0u, ws);
  else hipLaunchKernelGGL((k_gemm_ar<DT, 0>), dim3(grid), dim3(256), 0, st, p, p.wf, tiles_m, c_bytes, res_bytes, ws);
  return CA_OK;
}
}  // namespace

int ca_launch_gemm_ar(const ca_gemm_detail::GemmKParams& p0, int dtype, hipStream_t st) {
  static const int dbg = CA_KNOB("CA_PP_DBG", 0);  // (timing experiments: 1 = no stores)
  ca_gemm_detail::GemmKParams p = p0;
  p.dbg = dbg;
  return dtype == CA_BF16 ? launch_ar<CA_BF16>(p, st) : launch_ar<CA_F16>(p, st);
}

extern "C" int ca_pack_w_frag(const void* w, int32_t n, int32_t k, int32_t geglu, void* dst, void* stream) {
  CA_REQUIRE(w && dst, "ca_pack_w_frag: null operand");
  CA_REQUIRE(k == 320 && n > 0 && n % 64 == 0, "ca_pack_w_frag: n=%d (multiple of 64) k=%d (320)", n, k);
  CA_REQUIRE((((uintptr_t)w | (uintptr_t)dst) & 15) == 0, "ca_pack_w_frag: operands must be 16-byte aligned");
  const int64_t pieces = (int64_t)n * 40;
  hipLaunchKernelGGL(k_pack_w_frag, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u16*)w, (u16*)dst, n, geglu);
  CA_CHECK_LAUNCH("ca_pack_w_frag");
  return CA_OK;
}

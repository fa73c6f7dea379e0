// Issue cost of the packed fp32 -> 16-bit conversions on gfx950 (one wave, s_memtime around N independent instructions).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_cvt tools/probe_cvt.hip && /tmp/probe_cvt
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(const float* a, unsigned* o, long long* t) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = a[threadIdx.x + 64 * i];
  unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      unsigned r;
      if (MODE == 0) r = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){v[2 * i], v[2 * i + 1]}, h2));
      else if (MODE == 1) {
        _Float16 lo = (_Float16)v[2 * i], hi = (_Float16)v[2 * i + 1];
        r = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
      } else if (MODE == 2) r = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){v[2 * i], v[2 * i + 1]}, b2));
      else {
        typedef __fp16 hh2 __attribute__((ext_vector_type(2)));
        hh2 q = __builtin_amdgcn_cvt_pkrtz(v[2 * i], v[2 * i + 1]);
        r = __builtin_bit_cast(unsigned, q);
      }
      acc[i] ^= r;
      v[2 * i] += 1.0f;  // (keeps the conversion inside the loop)
    }
  }
  long long t1 = __builtin_readcyclecounter();
  unsigned x = 0;
  for (int i = 0; i < 8; ++i) x ^= acc[i];
  o[threadIdx.x] = x;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}

int main() {
  float* a;
  unsigned* o;
  long long* t;
  hipMalloc(&a, 64 * 16 * 4);
  hipMalloc(&o, 256);
  hipMalloc(&t, 8);
  hipMemset(a, 0, 64 * 16 * 4);
  const char* names[4] = {"v_cvt_pk_f16_f32 (vector conversion)", "cvt + cvt_sdwa + or (two scalar conversions)", "v_cvt_pk_bf16_f32", "v_cvt_pkrtz_f16_f32"};
  for (int rep = 0; rep < 2; ++rep)
    for (int m = 0; m < 4; ++m) {
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, a, o, t);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, a, o, t);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, a, o, t);
      if (m == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, a, o, t);
      long long h;
      hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
      printf("%-48s %6.1f cycles per pair (incl. one v_add + one v_xor)\n", names[m], (double)h / (64 * 8));
    }
  return 0;
}

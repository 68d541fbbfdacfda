// Does v_mfma_f32_32x32x16_f16 honour fp16 DENORMAL inputs on gfx950?  (The answer decides whether an fp16 hi + fp16 lo
// operand split -- 22 significand bits at the price of the bf16x3 scheme -- is usable: lo = a - fp16(a) is subnormal
// for |a| < 2^-3.)  Every A element = 2^-20 (subnormal in fp16), every B element = 1: D = 16 * 2^-20 if honoured, 0 if
// flushed.  Also: a normal-range control, and the same for bf16 with 2^-130.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f16_denorm tools/probes/mfma_f16_denorm.hip && ./mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void probe(float a_val, float b_val, float* out) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
  bf16x8 p, q;
  for (int i = 0; i < 8; ++i) { p[i] = (__bf16)a_val; q[i] = (__bf16)b_val; }
  f32x16 d = {};
  d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p, q, d, 0, 0, 0);
  if (threadIdx.x == 0) out[1] = d[0];
}
int main() {
  float* o;
  hipMalloc(&o, 8);
  const float cases[][2] = {{9.5367431640625e-07f /* 2^-20 */, 1.f}, {0.25f, 1.f}, {5.9604644775390625e-08f /* 2^-24: smallest subnormal */, 1.f},
                            {1.f, 9.5367431640625e-07f}};
  for (auto& cs : cases) {
    probe<<<1, 64>>>(cs[0], cs[1], o);
    float h[2];
    hipMemcpy(h, o, 8, hipMemcpyDeviceToHost);
    printf("a=%.6e b=%.6e : f16 mfma -> %.6e (expect %.6e if denormals are honoured)   bf16 mfma -> %.6e\n", cs[0], cs[1], h[0],
           16.0 * (double)(float)(_Float16)cs[0] * (double)(float)(_Float16)cs[1], h[1]);
  }
  return 0;
}

// libofx: measurement aid, not part of the operator path.
// ofx_probe_mfma_sustained: what rate does the matrix pipe SUSTAIN on this device with operands that look like the
// planes GraphConv's (fp16 hi / lo pairs of N(0, 1)-like activations and of weights scaled into [2^14, 2^15), read from
// LDS with ds_read_b128, 24 x v_mfma_f32_32x32x16_f16 per wave per step on four accumulators, 2 waves per SIMD, one
// s_barrier per step) and no global memory traffic at all?  On MI355X the answer is ~1.45 PFLOP/s at a shader clock of
// ~1.55 GHz -- not the 2.5 PFLOP/s of the data sheet, which the same loop only reaches with CONSTANT operands (2.4 PFLOP/s
// at 2.3 GHz): with real data the matrix pipe runs into the chip's power limit and the clock drops
// (tools/probes/mfma_rate.hip, profiles/r04/mfma_rate_probe.txt).  bench.py runs this on the box it benchmarks and
// reports the GraphConv kernel against both roofs.
#include "ofx_common.h"

typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* pldsp;

__device__ __forceinline__ float probe_gauss(unsigned h) {          // sum of four uniforms: near enough to normal
  float s = 0.f;
  for (int k = 0; k < 4; ++k) { h = h * 1664525u + 1013904223u; s += (float)(h >> 8) * (1.f / 16777216.f); }
  return (s - 2.f) * 1.7320508f;
}

__global__ void __launch_bounds__(512, 2) mfma_sustained_kernel(int steps, float* sink, unsigned long long* ticks) {
  extern __shared__ __attribute__((aligned(128))) char lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 96 * 1024 / 4; i += 512) {
    // 128-B lines [hi x 32 | lo x 32]; alternate 16 KB regions hold activations / scaled weights
    const int line = i >> 5, w_in = i & 31, is_lo = w_in >> 4, pairidx = w_in & 15;
    const bool weights = ((i * 4) >> 14) & 1;
    unsigned short hh[2];
    for (int e = 0; e < 2; ++e) {
      float v = probe_gauss((unsigned)(line * 32 + pairidx * 2 + e) * 2654435761u + blockIdx.x * 977u);
      if (weights) v *= 6000.f;
      const _Float16 hi = (_Float16)v;
      const _Float16 lo = (_Float16)(v - (float)hi);
      hh[e] = __builtin_bit_cast(unsigned short, is_lo ? lo : hi);
    }
    reinterpret_cast<unsigned*>(lds)[i] = (unsigned)hh[0] | ((unsigned)hh[1] << 16);
  }
  __syncthreads();
  pf16x8 fa[2][4], fb[2][4];
  const unsigned base = (unsigned)(uintptr_t)(pldsp)lds + (wid * 64 + lane) * 16;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[u][k]) : "v"(base), "n"(0));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[u][k]) : "v"(base), "n"(16384));
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  pf32x16 acc[2][2] = {};
  unsigned long long t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int s = 0; s < steps; ++s) {
    const unsigned ro = base + ((s * 1040) & 0x7fff);          // a different LDS window every step
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {                              // the next half step's fragments, under this half's MFMAs
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[half ^ 1][k]) : "v"(ro + half * 4096), "n"(0));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[half ^ 1][k]) : "v"(ro + half * 4096), "n"(16384));
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const pf16x8& a = fa[half][t == 0 ? 2 + i : i];
            const pf16x8& b = fb[half][t == 1 ? 2 + j : j];
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a), "v"(b));
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (half == 0) asm volatile("s_barrier" ::: "memory");
    }
  }
  unsigned long long t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  float r = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j)
      for (int k = 0; k < 16; ++k) r += acc[i][j][k];
  if (r == 123.456f) sink[0] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

// One launch of `steps` 24-MFMA steps per wave on `blocks` 512-thread blocks (0: one per compute unit).  The caller
// times it (events on `stream`): flops = blocks * 8 waves * steps * 24 * 32768; ticks[0] = s_memtime ticks of block 0
// (shader clocks: ticks / seconds = the clock the pipe ran at).  sink: >= 1 float, ticks: >= 1 uint64, both device.
extern "C" int ofx_probe_mfma_sustained(int steps, int blocks, float* sink, unsigned long long* ticks, void* stream) {
  if (steps < 1 || !sink || !ticks) return OFX_EINVAL;
  if (blocks <= 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      return OFX_ENODEV;
    blocks = cus;
  }
  static bool attr_set[OFX_MAX_DEVICES] = {};
  if (!ofx_raise_lds_limit(reinterpret_cast<const void*>(&mfma_sustained_kernel), 152 * 1024, attr_set)) return OFX_ELAUNCH;
  mfma_sustained_kernel<<<blocks, 512, 152 * 1024, ofx_stream(stream)>>>(steps, sink, ticks);
  OFX_LAUNCH_CHECK();
  return OFX_OK;
}

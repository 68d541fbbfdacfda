// What would the planes GraphConv's k-loop gain from 128 x 128 WAVE tiles ("build the 128 x 128
// wave-tile geometry")?  Two loops with the SAME instruction kinds and the same realistic operand data (fp16 hi / lo pairs
// of N(0, 1)-like activations and of weights scaled into [2^14, 2^15)), differing only in geometry:
//
//   A  today's gconv3 loop:  512 threads = 8 waves (2 per SIMD), wave tile 64 x 64 (MI 2, NI 2), block tile 256 x 128:
//      per k-step per wave 24 MFMAs (3 terms x 2 halves x 4 tiles), 16 ds_read_b128, 6 global_load_lds x 1 KB
//      (48 KB per CU per step), three LDS stages, one s_barrier per step;
//   B  the geometry asked for: 256 threads = 4 waves (1 per SIMD), wave tile 128 x 128 (MI 4, NI 4), block tile 256 x 256:
//      per k-step per wave 96 MFMAs on 256 accumulator registers (AGPRs), 32 ds_read_b128, 16 global_load_lds x 1 KB
//      (64 KB per CU per step = 2/3 of A's bytes per MFMA, half of its fragment reads per MFMA), two LDS stages;
//   C  4 waves with 128 x 64 wave tiles on today's block tile and stages;  D  8 waves (2 per SIMD) with 64 x 128 wave tiles on a
//      256 x 256 block tile, two stages: B's DMA ratio, 0.5 reads per MFMA, two waves per SIMD to cover each other's waits.
//
// The DMA source is an L2-resident buffer (the best case for both); LDS reads of the next half step are issued under the
// current half's MFMAs; the waits are hand-counted.  No epilogue, no gather misses, no tile switch: this is the ceiling of
// the k-loop alone.  Output: TFLOP/s issued, shader clock, clocks per k-step against the ideal (MFMAs x 32 clocks / SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/wave_tile tools/probes/wave_tile.hip && tools/probes/wave_tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const char __attribute__((address_space(1)))* gcp;
typedef __attribute__((address_space(3))) void* ldsp;

__device__ __forceinline__ float gauss(unsigned h) {
  float s = 0.f;
  for (int k = 0; k < 4; ++k) { h = h * 1664525u + 1013904223u; s += (float)(h >> 8) * (1.f / 16777216.f); }
  return (s - 2.f) * 1.7320508f;
}
__device__ __forceinline__ unsigned pair_word(unsigned idx, bool weights, bool is_lo, unsigned salt) {
  unsigned short hh[2];
  for (int e = 0; e < 2; ++e) {
    float v = gauss((idx * 2 + e) * 2654435761u + salt);
    if (weights) v *= 6000.f;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    hh[e] = __builtin_bit_cast(unsigned short, is_lo ? lo : hi);
  }
  return (unsigned)hh[0] | ((unsigned)hh[1] << 16);
}
// fill a buffer of 128-B lines [hi x 32 | lo x 32]; alternate 16 KB regions hold activations / scaled weights
__global__ void fill_kernel(unsigned* p, size_t words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned line = (unsigned)(i >> 5), w_in = (unsigned)(i & 31);
    p[i] = pair_word(line * 16 + (w_in & 15), ((i * 4) >> 14) & 1, (w_in >> 4) != 0, 12345u);
  }
}

template <bool AGPR>
__device__ __forceinline__ void mf(f32x16& c, const f16x8& a, const f16x8& b) {
  if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// WAVES per block, MI x NI tiles of 32 x 32 per wave, NBUF stages of STAGE bytes, GLDS DMA instructions per wave per step
template <int WAVES, int MI, int NI, int NBUF, int STAGE, int GLDS, bool AGPR, bool DMA>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) loop_kernel(int steps, const char* src, unsigned src_mask,
                                                                               float* out, unsigned long long* ticks) {
  extern __shared__ __attribute__((aligned(128))) char lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < NBUF * STAGE / 4; i += WAVES * 64) {
    const unsigned line = (unsigned)(i >> 5), w_in = (unsigned)(i & 31);
    reinterpret_cast<unsigned*>(lds)[i] = pair_word(line * 16 + (w_in & 15), ((i * 4) >> 14) & 1, (w_in >> 4) != 0, blockIdx.x * 977u);
  }
  __syncthreads();
  const unsigned lds0 = (unsigned)(uintptr_t)(ldsp)lds;
  // fragment read base: lane-linear 16-B pieces (conflict-free), a different window per wave
  const unsigned fbase = lds0 + ((wid * 64 + lane) * 16) % (STAGE / 2);
  f16x8 fa[2][2][MI], fb[2][2][NI];                      // [half buffer][hi | lo][tile]
  f32x16 acc[MI][NI] = {};
  // DMA: every lane fetches 16 B; a wave instruction = 1 KB landing lane-linear at its LDS destination
  gcp sp = (gcp)src + ((size_t)blockIdx.x * 40961u + (size_t)wid * 1024u + lane * 16u);
  auto dma = [&](int stage, int k, unsigned step) {
    const unsigned off = ((step * 24593u + k * 4099u + blockIdx.x * 7u) * 1024u) & src_mask;
    __builtin_amdgcn_global_load_lds(sp + off, (ldsp)(lds + stage * STAGE + (wid * GLDS + k) * 1024), 16, 0, 0);
  };
  auto read_half = [&](unsigned ro, int hb) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa[hb][u][i]) : "v"(ro + (u * MI + i) * 2048u));
#pragma unroll
      for (int j = 0; j < NI; ++j)
        asm volatile("ds_read_b128 %0, %1" : "=v"(fb[hb][u][j]) : "v"(ro + STAGE / 2 + (u * NI + j) * 2048u));
    }
  };
  if (DMA) {
#pragma unroll
    for (int s0 = 0; s0 < NBUF - 1; ++s0)
#pragma unroll
      for (int k = 0; k < GLDS; ++k) dma(s0, k, s0);
  }
  read_half(fbase, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  unsigned long long t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  int st = 0;                                               // stage of k-step s
  for (int s = 0; s < steps; ++s) {
    const unsigned ro = fbase + st * STAGE;
    const int st_dma = (st + NBUF - 1) % NBUF;              // the stage the step after next(s) will read
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      // next half step's fragments (second half: the NEXT stage's first half) under this half's MFMAs
      const unsigned rn = half == 0 ? ro + 1024u : fbase + ((st + 1) % NBUF) * STAGE;
      read_half(rn, half ^ 1);
      int issued = 0;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const f16x8& a = fa[half][t == 0 ? 1 : 0][i];
            const f16x8& b = fb[half][t == 1 ? 1 : 0][j];
            mf<AGPR>(acc[i][j], a, b);
            // the DMA requests of this step, spread between the MFMAs of its first half
            if (DMA && half == 0 && issued < GLDS && ((t * MI + i) * NI + j) % ((3 * MI * NI) / GLDS > 0 ? (3 * MI * NI) / GLDS : 1) == 0) {
              dma(st_dma, issued, (unsigned)s + NBUF - 1);
              ++issued;
            }
          }
      if (half == 0) {
        // all but this step's own requests have landed (the stage the next step reads), every wave has read its fragments
        if (DMA) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NBUF == 3 ? GLDS : 0) : "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    st = (st + 1) % NBUF;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  float r = 0.f;
  for (int i = 0; i < MI; ++i)
    for (int j = 0; j < NI; ++j)
      for (int k = 0; k < 16; ++k) r += acc[i][j][k];
  if (r == 123.456f) out[0] = r;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int WAVES, int MI, int NI, int NBUF, int STAGE, int GLDS, bool AGPR, bool DMA>
static void go(const char* name, int blocks, int steps, const char* src, unsigned mask) {
  float* o; unsigned long long* tk;
  hipMalloc(&o, 64); hipMalloc(&tk, 8 * blocks);
  auto kern = loop_kernel<WAVES, MI, NI, NBUF, STAGE, GLDS, AGPR, DMA>;
  const size_t lds = (size_t)NBUF * STAGE;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<blocks, WAVES * 64, lds>>>(steps / 8, src, mask, o, tk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<blocks, WAVES * 64, lds>>>(steps, src, mask, o, tk);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1]; hipMemcpy(h, tk, 8, hipMemcpyDeviceToHost);
  const double mfma_per_step = 2.0 * 3 * MI * NI;
  const double flop = (double)WAVES * blocks * steps * mfma_per_step * 32 * 32 * 16 * 2;
  const double ideal = mfma_per_step * 32.0 * (WAVES / 4.0);          // clocks per step per SIMD at full issue rate
  printf("%-58s %8.3f ms  %7.1f TFLOP/s issued  clocks/step %7.1f (ideal %5.0f: %.2f)  shader clock %.3f GHz  per MFMA: %.1f B DMA, %.2f reads\n",
         name, ms, flop / ms / 1e9, (double)h[0] / steps, ideal, ideal / ((double)h[0] / steps), (double)h[0] / (ms * 1e6),
         DMA ? GLDS * 1024.0 * WAVES / (mfma_per_step * WAVES) : 0.0, 4.0 * (MI + NI) / mfma_per_step);
  hipFree(o); hipFree(tk);
}

int main() {
  int dev = 0, cus = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const size_t src_bytes = 64u << 20;                       // 64 MB source; every CU cycles through an L2-sized window of it
  char* src;
  hipMalloc(&src, src_bytes + (1 << 20));
  fill_kernel<<<1024, 256>>>(reinterpret_cast<unsigned*>(src), (src_bytes + (1 << 20)) / 4);
  hipDeviceSynchronize();
  const unsigned mask = (2u << 20) - 1;                     // 2 MB window (+ the per-block / per-wave offsets): L2 hits
  printf("%d CUs; one block per CU\n", cus);
  for (int rep = 0; rep < 2; ++rep) {
    go<8, 2, 2, 3, 49152, 6, false, false>("A  8 waves, 64 x 64 wave tiles, no DMA", cus, 4000, src, mask);
    go<4, 4, 4, 2, 65536, 16, true, false>("B  4 waves, 128 x 128 wave tiles (AGPR acc), no DMA", cus, 2000, src, mask);
    go<8, 2, 2, 3, 49152, 6, false, true>("A  8 waves, 64 x 64 wave tiles, 48 KB DMA / step", cus, 4000, src, mask);
    go<4, 4, 4, 2, 65536, 16, true, true>("B  4 waves, 128 x 128 wave tiles (AGPR acc), 64 KB DMA / step", cus, 2000, src, mask);
    go<4, 4, 2, 3, 49152, 12, true, true>("C  4 waves, 128 x 64 wave tiles (AGPR acc), 48 KB DMA / step", cus, 4000, src, mask);
    go<8, 2, 4, 2, 65536, 8, true, true>("D  8 waves, 64 x 128 wave tiles (AGPR acc), 64 KB DMA / step, 2 stages", cus, 2000, src, mask);
    go<8, 2, 4, 2, 65536, 8, true, false>("D  8 waves, 64 x 128 wave tiles (AGPR acc), no DMA", cus, 2000, src, mask);
  }
  hipFree(src);
  return 0;
}

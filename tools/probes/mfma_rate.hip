// What MFMA rate does the planes GraphConv's instruction pattern reach with NOTHING else in the way?
// The kernel's k-step per wave = 24 x v_mfma_f32_32x32x16_f16 on 4 accumulators (3 terms x 2 halves), 2 waves per SIMD
// (512-thread block, one block per CU), accumulators in arch VGPRs.  This probe issues exactly that, with no memory
// traffic, and reports clocks per 24-MFMA step (ideal at 2 waves per SIMD: 2 x 24 x 32 = 1536) for:
//   acc in VGPRs / acc pinned in AGPRs;  1 / 2 waves per SIMD;  with / without ~30 integer VALU ops per step spliced
//   between the MFMAs (the DMA address arithmetic of the real loop);  with / without one s_barrier per step.
// Also reports s_memtime ticks against wall time (is s_memtime the shader clock?).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/probes/mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned long long memtime() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t));
  return t;
}

template <bool AGPR>
__device__ __forceinline__ void mf(f32x16& c, const f16x8& a, const f16x8& b) {
  if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

template <bool AGPR, bool VALU, bool BAR>
__global__ void __launch_bounds__(512, 2) rate(int steps, float* out, unsigned long long* ticks) {
  f16x8 ah[2], al[2], bh[2], bl[2];
  for (int i = 0; i < 8; ++i)
    for (int u = 0; u < 2; ++u) {
      ah[u][i] = (_Float16)(0.001f * (threadIdx.x + i + u)); al[u][i] = (_Float16)1e-4f;
      bh[u][i] = (_Float16)(0.002f * (i + 1)); bl[u][i] = (_Float16)2e-4f;
    }
  f32x16 acc[2][2] = {};
  unsigned long long v0 = threadIdx.x, v1 = blockIdx.x * 977u + 13u;
  const unsigned long long t0 = memtime();
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f16x8& a = t == 0 ? al[i] : ah[i];
            const f16x8& b = t == 1 ? bl[j] : bh[j];
            mf<AGPR>(acc[i][j], a, b);
            if constexpr (VALU) {
              if (t < 2) {        // ~16 of the 64-bit address ops per half step
                asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(v0) : "v"(v1));
                asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(v1));
              }
            }
          }
      if constexpr (BAR) { if (half == 0) asm volatile("s_barrier" ::: "memory"); }
    }
  }
  const unsigned long long t1 = memtime();
  float r = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) r += acc[i][j][k];
  if (r == 123.456f || v0 == 1) out[0] = r;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <bool AGPR, bool VALU, bool BAR>
static void go(const char* name, int threads, int blocks, int steps) {
  float* o; unsigned long long* tk;
  hipMalloc(&o, 64); hipMalloc(&tk, 8 * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rate<AGPR, VALU, BAR><<<blocks, threads>>>(steps / 10, o, tk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  rate<AGPR, VALU, BAR><<<blocks, threads>>>(steps, o, tk);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1]; hipMemcpy(h, tk, 8, hipMemcpyDeviceToHost);
  const double waves = threads / 64.0 * blocks;
  const double flop = waves * steps * 24.0 * 32 * 32 * 16 * 2;
  printf("%-44s %4d thr x %3d blk: %8.3f ms  %7.1f TFLOP/s  memtime ticks/step %7.1f  tick rate %.3f GHz\n", name, threads, blocks,
         ms, flop / ms / 1e9, (double)h[0] / steps, (double)h[0] / (ms * 1e6));
  hipFree(o); hipFree(tk);
}

// ---- second question: what does the CLOCK do?  The same 24-MFMA step with (a) random operands that change every step
// (real toggling), (b) + the 24 ds_read_b128 per step the real loop issues (operands come from LDS), (c) + a
// global_load_lds stream of 6 x 1 KB per wave per step from a large buffer (the L2 -> LDS gather traffic).  One barrier
// per step so that s_memtime ticks per step x steps / wall time = the shader clock.
typedef const char __attribute__((address_space(1)))* gcp;
typedef __attribute__((address_space(3))) void* ldsp;
// FILL 0: uniformly random bit patterns (worst-case toggling); 1: REALISTIC operand pairs -- even 128-B lines hold
// [hi x 32 | lo x 32] of N(0, 1)-like activations, the lines 16 KB further on [hi | lo] of U(-1, 1) weights scaled into
// [2^14, 2^15) as the packs are (fp16 pairs), or the same values as bf16 pairs (BF16).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float gauss(unsigned h) {          // sum of four uniforms: near enough to normal
  float s = 0.f;
  for (int k = 0; k < 4; ++k) { h = h * 1664525u + 1013904223u; s += (float)(h >> 8) * (1.f / 16777216.f); }
  return (s - 2.f) * 1.7320508f;
}
template <int LEVEL, int FILL = 0, bool BF16 = false>
__global__ void __launch_bounds__(512, 2) power(int steps, const char* src, size_t src_bytes, float* out, unsigned long long* ticks) {
  extern __shared__ __attribute__((aligned(128))) char lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  // fill 96 KB of LDS with pseudo-random halves
  for (int i = threadIdx.x; i < 96 * 1024 / 4; i += 512) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    // keep the fp16 exponents moderate: clear the top exponent bit of both halves
    unsigned word = h & 0xbfffbfffu;
    if constexpr (FILL == 1) {
      // word i holds halves 2i, 2i+1 of a 128-B line = 64 halves: first 32 = hi, last 32 = lo of the same 32 values
      const int line = i >> 5, w_in = i & 31, is_lo = w_in >> 4, pairidx = (w_in & 15);
      const bool weights = ((i * 4) >> 14) & 1;                       // alternate 16 KB regions: activations / weights
      unsigned short hh[2];
      for (int e = 0; e < 2; ++e) {
        const unsigned seed = (unsigned)(line * 32 + pairidx * 2 + e) * 2654435761u + blockIdx.x * 977u;
        float v = gauss(seed);
        if (weights) v = (v * 0.3f) * 20000.f;                       // |w| scaled towards [2^14, 2^15)
        if constexpr (BF16) {
          const unsigned u = __float_as_uint(v);
          const unsigned hb = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
          const float lo = v - __uint_as_float(hb << 16);
          const unsigned ul = __float_as_uint(lo);
          const unsigned lb = (ul + 0x7fffu + ((ul >> 16) & 1u)) >> 16;
          hh[e] = (unsigned short)(is_lo ? lb : hb);
        } else {
          const _Float16 hi = (_Float16)v;
          const _Float16 lo = (_Float16)(v - (float)hi);
          hh[e] = __builtin_bit_cast(unsigned short, is_lo ? lo : hi);
        }
      }
      word = (unsigned)hh[0] | ((unsigned)hh[1] << 16);
    }
    reinterpret_cast<unsigned*>(lds)[i] = word;
  }
  __syncthreads();
  f16x8 fa[2][4], fb[2][4];
  const unsigned base = (unsigned)(uintptr_t)(ldsp)lds + (wid * 64 + lane) * 16;
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[u][k]) : "v"(base), "n"(0));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[u][k]) : "v"(base), "n"(8192));
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  f32x16 acc[2][2] = {};
  const size_t stride = (size_t)gridDim.x * 8 * 6 * 1024;
  size_t goff = ((size_t)blockIdx.x * 8 + wid) * 6 * 1024 + lane * 16;
  const unsigned long long t0 = memtime();
  for (int s = 0; s < steps; ++s) {
    const unsigned ro = base + ((s * 1040) & 0x7fff);          // a different LDS window every step
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if constexpr (LEVEL >= 2) {
        // the real loop reads the NEXT half step's 12 fragments while this half's MFMAs run
#pragma unroll
        for (int k = 0; k < (LEVEL == 6 ? 2 : 4); ++k) {
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[half ^ 1][k]) : "v"(ro + half * 4096), "n"(0));
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[half ^ 1][k]) : "v"(ro + half * 4096), "n"(16384));
        }
        // (four extra reads: the real loop has 12 per half step, this has 8 -> add 4)
      }
      if constexpr (LEVEL >= 3 && LEVEL != 6) {
        if (half == 0) {
#pragma unroll
          for (int k = 0; k < (LEVEL == 5 ? 4 : 6); ++k) {
            // LEVEL 3: a stream through the whole buffer (HBM); LEVEL >= 4: every wave re-reads its own 12 KB (3 MB per XCD:
            // L2 hits, the regime of the real kernel, whose gathered lines are 7x re-used across the directions)
            const size_t o = LEVEL == 3 ? (goff + (size_t)k * 1024) % src_bytes
                                        : ((size_t)blockIdx.x * 8 + wid) * 12288 + (size_t)(s & 1) * 6144 + (size_t)k * 1024 + lane * 16;
            __builtin_amdgcn_global_load_lds((gcp)src + o, (ldsp)(lds + 96 * 1024 + wid * 6144 + k * 1024), 16, 0, 0);
          }
          goff += stride;
        }
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f16x8& a = fa[half][t == 0 ? 2 + i : i];
            const f16x8& b = fb[half][t == 1 ? 2 + j : j];
            if constexpr (BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a), "v"(b));
          }
      if constexpr (LEVEL >= 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (half == 0) {
        if constexpr (LEVEL >= 3 && LEVEL != 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
      }
    }
    if constexpr (LEVEL == 1) {       // no LDS reads: still change the operands every step (rotate the register sets)
      const f16x8 tmp = fa[0][0];
      fa[0][0] = fa[0][1]; fa[0][1] = fa[0][2]; fa[0][2] = fa[0][3]; fa[0][3] = fa[1][0];
      fa[1][0] = fa[1][1]; fa[1][1] = fa[1][2]; fa[1][2] = fa[1][3]; fa[1][3] = tmp;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = memtime();
  float r = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) r += acc[i][j][k];
  if (r == 123.456f) out[0] = r;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int LEVEL, int FILL = 0, bool BF16 = false>
static void gop(const char* name, int steps, size_t nb = (size_t)1 << 30) {
  float* o; unsigned long long* tk; char* src;
  hipMalloc(&o, 64); hipMalloc(&tk, 8 * 256); hipMalloc(&src, nb);
  hipMemset(src, 0x3c, nb);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&power<LEVEL, FILL, BF16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  power<LEVEL, FILL, BF16><<<256, 512, 152 * 1024>>>(steps / 10, src, nb, o, tk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  power<LEVEL, FILL, BF16><<<256, 512, 152 * 1024>>>(steps, src, nb, o, tk);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1]; hipMemcpy(h, tk, 8, hipMemcpyDeviceToHost);
  const double flop = 8.0 * 256 * steps * 24.0 * 32 * 32 * 16 * 2;
  printf("%-52s %8.3f ms  %7.1f TFLOP/s  ticks/step %7.1f  clock %.3f GHz  (gather stream %.2f TB/s)\n", name, ms, flop / ms / 1e9,
         (double)h[0] / steps, (double)h[0] / (ms * 1e6), LEVEL >= 3 ? 256.0 * 8 * 6 * 1024 * steps / ms / 1e9 : 0.0);
  hipFree(o); hipFree(tk); hipFree(src);
}

int main() {
  gop<1>("random operands rotating, MFMA only, barrier/step", 40000);
  gop<2>("+ 16 ds_read_b128 per step (operands from LDS)", 40000);
  gop<2, 1, false>("operands from LDS, REALISTIC fp16 hi/lo pairs", 40000);
  gop<2, 1, true>("operands from LDS, REALISTIC bf16 hi/lo pairs", 40000);
  gop<2, 0, true>("operands from LDS, random bits, bf16 MFMA", 40000);
  gop<3>("+ global_load_lds 6 KB / wave / step (L2/HBM -> LDS)", 10000);
  gop<4, 1, false>("realistic pairs + LDS reads + DMA 6 KB/wave/step, L2-HOT source (16 MB)", 20000, (size_t)32 << 20);
  gop<5, 1, false>("realistic pairs + LDS reads + DMA 4 KB/wave/step, L2-hot (the 256x256-tile ratio)", 20000, (size_t)32 << 20);
  gop<6, 1, false>("realistic pairs + HALF the LDS reads, no DMA (128x128 wave tiles)", 20000);

  const int S = 20000;
  go<false, false, false>("VGPR acc, MFMA only", 512, 256, S);
  go<true, false, false>("AGPR acc, MFMA only", 512, 256, S);
  go<false, false, false>("VGPR acc, MFMA only, 1 wave/SIMD", 256, 256, S);
  go<true, false, false>("AGPR acc, MFMA only, 1 wave/SIMD", 256, 256, S);
  go<false, true, false>("VGPR acc, + 64-bit VALU address ops", 512, 256, S);
  go<true, true, false>("AGPR acc, + 64-bit VALU address ops", 512, 256, S);
  go<false, true, true>("VGPR acc, + VALU + s_barrier per step", 512, 256, S);
  go<true, true, true>("AGPR acc, + VALU + s_barrier per step", 512, 256, S);
  go<false, false, false>("VGPR acc, MFMA only, ONE block (clock unsagged)", 512, 1, S);
  return 0;
}
